"""How often is each constraint slot of the step kernel in use under the benchmark's uniformly random policy?  (GPU box)
    python tools/slot_activity.py [task] [n] [steps]
Per control step and env the kernel's diagnostics word ORs the activity of every slot over the 20 substeps (lcr_config.diagnostics = 1);
printed: the fraction of envs with the slot active in a step, and the fraction of 64-env waves in which NO lane has it (the waves a
wave-uniform skip would help)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from gym_lowcostrobot_amd import VecSim  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "reach"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
sim = VecSim(task, n, diagnostics=1, step_kernel="single", preset="fast")   # (step_kernel pins a family of the sweep kernels: preset fast)
act = sim.alloc_actions()
names = {12: "finger 0 <-> cube", 13: "finger 1 <-> cube", 14: "finger 0 <-> floor", 15: "finger 1 <-> floor", 16: "link proxy"}
names.update({18 + j: f"joint limit {j}" for j in range(6)})
names.update({c * 4 + s: f"cube {c} floor corner slot {s}" for c in range(2 if task == "stack" else 1) for s in range(4)})
env_rate = {b: [] for b in names}
wave_none = {b: [] for b in names}
any_arm_none, any_lim_none = [], []
for t in range(steps):
    sim.fill_random_actions(act, 0, t)
    sim.step_device(act.ptr)
    if t < 10:
        continue
    m = sim.active_mask.numpy().astype(np.uint32)
    w = m.reshape(-1, 64)
    for b in names:
        on = (w >> np.uint32(b)) & np.uint32(1)
        env_rate[b].append(on.mean())
        wave_none[b].append((on.max(axis=1) == 0).mean())
    arm = (w >> np.uint32(14)) & np.uint32(7)
    any_arm_none.append((arm.max(axis=1) == 0).mean())
    lim = (w >> np.uint32(18)) & np.uint32(63)
    any_lim_none.append((lim.max(axis=1) == 0).mean())
print(f"{task}, {n} envs, steps 10..{steps - 1}: slot active in a control step (any of 20 substeps)")
for b in sorted(names):
    print(f"  bit {b:2d} {names[b]:28s}: {100 * np.mean(env_rate[b]):6.2f} % of envs   waves with no lane: {100 * np.mean(wave_none[b]):6.2f} %")
print(f"  waves with none of the three arm-only slots (finger<->floor, proxy): {100 * np.mean(any_arm_none):.2f} %; with no joint limit: {100 * np.mean(any_lim_none):.2f} %")
sim.close()
