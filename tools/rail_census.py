"""Tail census of the cube state under the benchmark's random policy (GPU box):  python tools/rail_census.py [steps [tasks [key=value ...]]]
For every task: how fast / how deep / how far does the cube get in 65 536 envs, sampled every 7th step.  PushCubeLoop also counts the cubes
outside the rails.  (Round 3: with the rails as half-spaces, D7, 0.5 % of the PushCubeLoop states had the cube beyond a rail's outer face,
459 of 5.6e6 moved faster than 5 m/s, the fastest at 1 240 m/s -- ejected by a 'penetration' of decimetres; with the rails acting only
while the cube centre is inside their outer rectangle: 1 of 5.6e6 above 5 m/s, fastest 5.8 m/s.)"""
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from gym_lowcostrobot_amd import VecSim  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None          # e.g. stack,lift
kw = {}
for a in sys.argv[3:]:                                                 # e.g. pgs_iters=8 step_kernel=single
    k, v = a.split("=")
    kw[k] = int(v) if v.lstrip("-").isdigit() else v
n = 65536
for task, mode in (("push_loop", "joint"), ("reach", "joint"), ("push", "joint"), ("lift", "joint"), ("pick_place", "ee"), ("stack", "joint")):
    if only and task not in only:
        continue
    sim = VecSim(task, n, action_mode=mode, **kw)
    act = sim.alloc_actions()
    tot = fast5 = fast50 = outside = sunk = 0
    vmax = wmax = 0.0
    zmin = 1.0
    for t in range(steps):
        sim.fill_random_actions(act, 7, t)
        sim.step_device(act.ptr)
        if t % 7 == 6:
            st = sim.get_state()
            for c in range(2 if task == "stack" else 1):
                v = np.linalg.norm(st["qvel"][6 + 6 * c:9 + 6 * c], axis=0)
                w = np.linalg.norm(st["qvel"][9 + 6 * c:12 + 6 * c], axis=0)
                p = st["qpos"][6 + 7 * c:9 + 7 * c]
                assert np.isfinite(v).all() and np.isfinite(p).all()
                tot += n; fast5 += int((v > 5).sum()); fast50 += int((v > 50).sum())
                vmax = max(vmax, float(v.max())); wmax = max(wmax, float(w.max()))
                zmin = min(zmin, float(p[2].min())); sunk += int((p[2] < 0.010).sum())
                if task == "push_loop":
                    outside += int(((np.abs(p[0]) > 0.135) | (p[1] < 0.08) | (p[1] > 0.19)).sum())
    msg = (f"{task:10s} {mode:5s}: {tot} sampled cube states; |v| > 5 m/s in {fast5} ({100 * fast5 / tot:.4f} %), > 50 m/s in {fast50}; max |v| {vmax:.1f} m/s, "
           f"max |w| {wmax:.0f} rad/s; centre lower than 10 mm in {100 * sunk / tot:.3f} % (lowest {1e3 * zmin:.1f} mm)")
    if task == "push_loop":
        msg += f"; centre outside the rails' outer rectangle in {100 * outside / tot:.3f} %"
    print(msg, flush=True)
    sim.close()
