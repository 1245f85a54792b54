"""Kernel time per control step of every workload (GPU box).  python tools/quick_times.py [workload ...] [--n 65536] [--steps 200]
Used as the inner loop of kernel optimisation: HIP-event time of `steps` launches after a warm-up, median of 3 regions."""
import argparse
import sys

sys.path.insert(0, ".")
import numpy as np  # noqa: E402

from gym_lowcostrobot_amd import VecSim  # noqa: E402

W = {"reach": ("reach", "joint"), "push": ("push", "joint"), "lift": ("lift", "joint"), "pick_place_ee": ("pick_place", "ee"),
     "stack": ("stack", "joint"), "push_loop": ("push_loop", "joint"), "reach_ee": ("reach", "ee")}
ap = argparse.ArgumentParser()
ap.add_argument("names", nargs="*", default=list(W))
ap.add_argument("--n", type=int, default=65536)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--arm-collision", type=int, default=1)
ap.add_argument("--pgs-iters", type=int, default=None)
ap.add_argument("--preset", default=None)
ap.add_argument("--newton-iters", type=int, default=None)
ap.add_argument("--ls-iters", type=int, default=None)
ap.add_argument("--newton-tol", type=float, default=None)
ap.add_argument("--ls-tol", type=float, default=None)
ap.add_argument("--finger-cube-condim", type=int, default=None)
ap.add_argument("--cc-points", type=int, default=None)
a = ap.parse_args()
for name in a.names:
    task, mode = W[name]
    sim = VecSim(task, a.n, action_mode=mode, arm_collision=a.arm_collision, pgs_iters=a.pgs_iters, finger_cube_condim=a.finger_cube_condim, cc_points=a.cc_points, preset=a.preset, newton_iters=a.newton_iters, ls_iters=a.ls_iters, newton_tol=a.newton_tol, ls_tol=a.ls_tol)
    bufs = [sim.alloc_actions() for _ in range(32)]
    for i, b in enumerate(bufs):
        sim.fill_random_actions(b, 0, i)
    for i in range(60):
        sim.step_device(bufs[i % 32].ptr)
    ms = []
    for r in range(3):
        sim.timer_begin()
        for i in range(a.steps):
            sim.step_device(bufs[i % 32].ptr)
        ms.append(sim.timer_end() / a.steps)
    st = sim.get_state()
    ok = bool(np.isfinite(st["qpos"]).all())
    print(f"{name:14s} n={a.n} arm_collision={a.arm_collision} preset={a.preset} newton={a.newton_iters}/{a.ls_iters}/{a.newton_tol}/{a.ls_tol} pgs={a.pgs_iters} condim={a.finger_cube_condim} cc_points={a.cc_points}: {np.median(ms):.4f} ms/step  ({a.n / np.median(ms) * 1e3:.3e} env-steps/s)  finite={ok}", flush=True)
    sim.close()
