"""Distance of the product's contact solve from the EXACT optimum of MuJoCo's published convex constraint problem.

The yard-stick is solver-independent: the CPU oracle's Newton solve of the primal problem (orc_params.solver = 1, the algorithm MuJoCo runs by default --
follower.xml:3 names no solver), certified by the natural residual of the DUAL problem's KKT conditions (orc_io.kkt).  From identical states incl. the carried
constraint forces (the product's default mode), ONE control step (20 substeps) is made with
  default        4 warm-started sweeps of the block projected-gradient step (round 4)  = what both kernel families and the oracle's default run (deviation D1)
  default8 / *   8 sweeps / swept until converged                                       -> the same fixed point as `exact`
  rows+radial    rounds 1-3: row-by-row updates + radial cone projection, 4 sweeps      (deviations D1 + D2)
  rows+radial*   ... swept until converged                                               -> isolates D2: its fixed point is not the optimum
  qcqp           MuJoCo's PGS block update (ray step + exact friction QCQP), 4 sweeps
  exact          the optimum itself (primal Newton)
and |dqpos| of each against `exact` is reported per task, together with the KKT residual each variant leaves.
    python tools/kkt_distance.py [--n 512] [--steps 40] [--json profiles/r04_kkt_distance.json]
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402

STATE = ("qpos", "qvel", "ee_lag", "target", "elapsed", "rng", "goal", "sim_time", "warm")


def measure(task, mode, n, steps, seed=5):
    kw = dict(auto_reset=0, max_episode_steps=0, action_mode={"joint": 0, "ee": 1}[mode])
    walk = orc.Oracle(task, n, **kw)                                                  # generates the states (default solver, random policy)
    var = {"default": orc.Oracle(task, n, kkt=True, **kw),                                                          # block step, 4 sweeps: the product
           "gs": orc.Oracle(task, n, kkt=True, jacobi=0, **kw),                                                    # the same step, one Gauss-Seidel pass over all rows
           "default8": orc.Oracle(task, n, kkt=True, pgs_iters=8, **kw),
           "default*": orc.Oracle(task, n, kkt=True, pgs_iters=-1, pgs_tol=1e-11, pgs_cap=20000, **kw),            # ... swept to convergence
           "rows+radial": orc.Oracle(task, n, kkt=True, cone=0, **kw),                                              # rounds 1-3, 4 sweeps
           "rows+radial*": orc.Oracle(task, n, kkt=True, cone=0, pgs_iters=-1, pgs_tol=1e-11, pgs_cap=5000, **kw),  # ... swept to convergence: NOT the optimum
           "qcqp": orc.Oracle(task, n, kkt=True, cone=1, **kw),                                                    # MuJoCo's PGS block update, 4 sweeps
           "exact": orc.Oracle(task, n, kkt=True, solver=1, **kw)}
    walk.reset(np.arange(n, dtype=np.uint64) + 77)
    rng = np.random.default_rng(seed)
    dq = {k: [] for k in var if k != "exact"}
    kkt = {k: [] for k in var}
    touched = []
    for t in range(steps):
        act = rng.uniform(-1, 1, (n, walk.action_dim)).astype(np.float32)
        if t >= 3:
            for o in var.values():
                for k in STATE:
                    getattr(o, k)[:] = getattr(walk, k)
                o.step(act, 0)
            ex = var["exact"]
            for k in dq:
                dq[k].append(np.abs(var[k].qpos[:, : ex.nq] - ex.qpos[:, : ex.nq]).max(1))
            for k, o in var.items():
                kkt[k].append(o.kkt.copy())
            touched.append((ex.active_mask & 0x1F000) != 0)                            # an arm-coupled contact (finger / proxy) was active in the step
        walk.step(act, 0)
    out = {k: np.concatenate(v) for k, v in dq.items()}
    return out, {k: np.concatenate(v) for k, v in kkt.items()}, np.concatenate(touched)


def pct(x):
    return {"median": float(np.median(x)), "p90": float(np.percentile(x, 90)), "p99": float(np.percentile(x, 99)), "max": float(x.max())}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    res = {}
    for task, mode in (("reach", "joint"), ("push", "joint"), ("lift", "joint"), ("pick_place", "ee"), ("stack", "joint"), ("push_loop", "joint")):
        dq, kkt, touched = measure(task, mode, a.n, a.steps)
        res[f"{task}-{mode}"] = {"env_steps": int(dq["default"].size), "arm_contact_fraction": float(touched.mean()),
                                 "dqpos_vs_exact": {k: pct(v) for k, v in dq.items()},
                                 "dqpos_default_vs_exact_where_arm_touches": pct(dq["default"][touched]) if touched.any() else None,
                                 "kkt_residual": {k: pct(v) for k, v in kkt.items()}}
        r = res[f"{task}-{mode}"]
        f = lambda d: "median %.1e  p90 %.1e  p99 %.1e  max %.1e" % (d["median"], d["p90"], d["p99"], d["max"])
        print(f"{task:10s} {mode:5s} {r['env_steps']} env-steps, an arm contact in {100 * r['arm_contact_fraction']:.1f} %\n"
              + "".join(f"   |dqpos| {k:13s} vs exact: {f(pct(v))}   KKT residual {f(r['kkt_residual'][k])}\n" for k, v in dq.items())
              + f"   KKT residual of the exact solve: {f(r['kkt_residual']['exact'])}", flush=True)
    if a.json:
        with open(a.json, "w") as fjs:
            json.dump(res, fjs, indent=1)
