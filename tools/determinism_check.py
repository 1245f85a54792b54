"""Run-to-run determinism of the dispatched step kernels at full occupancy (GPU box):  python tools/determinism_check.py [steps]
Two sims of 65 536 envs with the same seeds and actions must agree bit for bit after every step block (a race between the two cooperating
waves of a workgroup -- LDS hand-overs, the global scratch records of the two-waves-per-SIMD build -- shows up as run-to-run differences,
and only at the occupancy the race needs)."""
import sys

import numpy as np

sys.path.insert(0, ".")
from gym_lowcostrobot_amd import VecSim  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
for task, mode, n in (("reach", "joint", 65536), ("reach", "ee", 65536), ("push", "joint", 65536), ("pick_place", "ee", 32768), ("stack", "joint", 32768), ("push_loop", "joint", 32768)):
    sims = [VecSim(task, n, action_mode=mode, base_seed=3) for _ in range(2)]
    acts = [s.alloc_actions() for s in sims]
    bad = 0
    for t in range(steps):
        for s, a in zip(sims, acts):
            s.fill_random_actions(a, 9, t)
            s.step_device(a.ptr)
        if t % 20 == 19:
            st = [s.get_state() for s in sims]
            for k in ("qpos", "qvel", "warm", "elapsed", "rng"):
                bad += int((st[0][k] != st[1][k]).sum())
    print(f"{task:10s} {mode:5s} n={n}: {sims[0].step_kernel_family}: {steps} steps, differing state words between two runs: {bad}", flush=True)
    for s in sims:
        s.close()
