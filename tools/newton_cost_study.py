"""What a 64-env wave pays for the Newton solve (CPU oracle): per substep the wave runs as many Newton iterations as its slowest env and every line search
as many evaluations as its slowest env -- from the oracle's per-(env, substep) trace.  Also: accuracy against the exact optimum for the same tolerances.
    python tools/newton_cost_study.py [--n 1024] [--steps 12]"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402

STATE = ("qpos", "qvel", "ee_lag", "target", "elapsed", "rng", "goal", "sim_time", "warm")
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--tasks", default="reach,lift,pick_place,stack,push_loop")
a = ap.parse_args()
L = orc.lib()
for task in a.tasks.split(","):
    mode = 1 if task == "pick_place" else 0
    kw = dict(auto_reset=0, max_episode_steps=0, action_mode=mode)
    n = a.n
    walk = orc.Oracle(task, n, preset="faithful", **kw)
    ex = orc.Oracle(task, n, preset="faithful", **{**kw, "solver": 1})
    walk.reset(np.arange(n, dtype=np.uint64) + 77)
    rng = np.random.default_rng(5)
    print(task)
    for tol, lst in ((1e-6, 1e-4), (1e-5, 1e-3), (1e-4, 1e-3), (1e-4, 1e-2), (1e-3, 1e-2)):
        c = orc.Oracle(task, n, preset="faithful", newton_tol=tol, ls_tol=lst, **kw)
        w2 = orc.Oracle(task, n, preset="faithful", **kw)
        for k in STATE:
            getattr(w2, k)[:] = getattr(walk, k)
        rng2 = np.random.default_rng(6)
        DQ, IT, LS, WIT, WLS = [], [], [], [], []
        for t in range(a.steps):
            act = rng2.uniform(-1, 1, (n, w2.action_dim)).astype(np.float32)
            if t >= 2:
                for o in (c, ex):
                    for k in STATE:
                        getattr(o, k)[:] = getattr(w2, k)
                ex.step(act, 0)
                trace = np.zeros((n, 20, 2), np.int32)
                L.orc_set_newton_trace(trace.ctypes.data_as(ctypes.c_void_p), 20)
                c.step(act, 0)
                L.orc_set_newton_trace(None, 0)
                DQ.append(np.abs(c.qpos[:, : ex.nq] - ex.qpos[:, : ex.nq]).max(1))
                it, ls = trace[:, :, 0], trace[:, :, 1]
                IT.append(it.mean()); LS.append(ls.mean())
                wv = it.reshape(n // 64, 64, 20)
                WIT.append(wv.max(1).mean())
                # a wave's line searches: per Newton iteration it runs max-over-lanes evaluations; bound: (max iterations of the wave) x (largest mean evaluations per iteration of a lane)
                per = np.where(it > 0, ls / np.maximum(it, 1), 0).reshape(n // 64, 64, 20)
                WLS.append((wv.max(1) * per.max(1)).mean())
            w2.step(act, 0)
        dq = np.concatenate(DQ)
        print(f"   newton_tol {tol:.0e} ls_tol {lst:.0e}: |dqpos| vs exact p50 {np.median(dq):.1e} p90 {np.percentile(dq, 90):.1e} p99 {np.percentile(dq, 99):.1e} max {dq.max():.1e}"
              f" | per env and substep: {np.mean(IT):.2f} iterations, {np.mean(LS):.2f} phi' evaluations | per 64-env wave: {np.mean(WIT):.2f} iterations, <= {np.mean(WLS):.1f} evaluations", flush=True)
