"""Why PushCubeLoop keeps the row-wise sweeps: physical tail statistics of the ORACLE under each solver variant, random policy.

    python tools/loop_solver_study.py [--n 2048] [--steps 200]

Prints, per variant, the lowest cube centre, the fastest cube and the fastest spin seen in n envs x steps control steps (auto-reset on), next to
MuJoCo's optimum (orc_params.solver = 1).  CPU only (oracle/); minutes.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402

VARIANTS = [("row-wise sweeps + radial projection, one sequence (the loop kernel)", dict(cone=0, jacobi=0)),
            ("row-wise sweeps, two concurrent groups", dict(cone=0, jacobi=1)),
            ("block projected gradient (three weights, exact projection), one sequence", dict(cone=3, jacobi=0)),
            ("block projected gradient, two concurrent groups (the other tasks' solver)", dict(cone=3, jacobi=1)),
            ("block projected gradient, two groups, 8 sweeps", dict(cone=3, jacobi=1, pgs_iters=8)),
            ("MuJoCo PGS (ray + exact friction QCQP), one sequence", dict(cone=1, jacobi=0)),
            ("MuJoCo's optimum (Newton on the primal)", dict(solver=1))]


def stats(task, n, steps, **kw):
    o = orc.Oracle(task, n, **kw)
    o.reset(seeds=np.arange(n, dtype=np.uint64))
    rng = np.random.default_rng(0)
    vmax = wmax = 0.0
    zmin = 1.0
    for _ in range(steps):
        o.step(rng.uniform(-1, 1, (n, o.action_dim)).astype(np.float32), threads=0)
        vmax = max(vmax, float(np.linalg.norm(o.qvel[:, 6:9], axis=1).max()))
        wmax = max(wmax, float(np.linalg.norm(o.qvel[:, 9:12], axis=1).max()))
        zmin = min(zmin, float(o.qpos[:, 8].min()))
    return zmin, vmax, wmax


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=2048)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--task", default="push_loop")
    a = ap.parse_args()
    print(f"{a.task}: {a.n} envs x {a.steps} control steps, random policy (cube half-size {7.5 if a.task == 'push_loop' else 15.0} mm)")
    for name, kw in VARIANTS:
        z, v, w = stats(a.task, a.n, a.steps, **kw)
        print(f"  {name:78s}: lowest centre {1e3 * z:7.1f} mm, fastest cube {v:6.2f} m/s, fastest spin {w:7.1f} rad/s", flush=True)
