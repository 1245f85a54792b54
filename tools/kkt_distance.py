"""Distance of the product's contact solve from the EXACT optimum of MuJoCo's published convex constraint problem.

The yard-stick is solver-independent: the CPU oracle's Newton solve of the primal problem (orc_params.solver = 1, the algorithm MuJoCo runs by default --
follower.xml:3 names no solver), certified by the natural residual of the DUAL problem's KKT conditions (orc_io.kkt).  From identical states incl. the carried
constraint forces (the product's default mode), ONE control step (20 substeps) is made with
  default        round 5: preset faithful -- Newton's method on the primal with a fixed budget, six-row finger contacts, eight-point box-box (= the Newton kernels)
  default-f32    the same in float arithmetic (the oracle's fp32 build)
  fast / fast8   preset fast: 4 / 8 warm-started sweeps of the block projected-gradient step on its own (smaller) row set -- the rounds 1-4 default; its distance
                 contains the contact-model deviations D4 / D5 as well as the unconverged solve D1
  exact          the optimum itself (primal Newton to 1e-13)
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
    walk = orc.Oracle(task, n, **kw)                                                  # generates the states (the default = faithful preset, random policy)
    var = {"default": orc.Oracle(task, n, kkt=True, **kw),                                                          # round 5: preset faithful -- Newton with a fixed budget (newton_product) = the Newton kernels
           "default-f32": orc.Oracle(task, n, f32=True, **kw),                                                     # ... the same C source in float arithmetic: what fp32 costs
           "fast": orc.Oracle(task, n, kkt=True, preset="fast", **kw),                                             # preset fast: four block projected-gradient sweeps, fewer rows (rounds 1-4 default)
           "fast8": orc.Oracle(task, n, kkt=True, preset="fast", pgs_iters=8, **kw),
           "exact": orc.Oracle(task, n, kkt=True, solver=1, **kw)}                                                 # the optimum of the faithful preset's contact model (primal Newton to 1e-13)
    walk.reset(np.arange(n, dtype=np.uint64) + 77)
    rng = np.random.default_rng(seed)
    dq = {k: [] for k in var if k != "exact"}
    kkt = {k: [] for k in var}
    touched = []
    for t in range(steps):
        act = rng.uniform(-1, 1, (n, walk.action_dim)).astype(np.float32)
        if t >= 3:
            for name, o in var.items():
                for k in STATE:
                    if k == "warm" and name.endswith("f32"):
                        o.warm.view(np.float32)[:, :192] = walk.warm.view(np.float64)[:, :192]
                    else:
                        getattr(o, k)[:] = getattr(walk, k)
                o.step(act, 0)
            ex = var["exact"]
            for k in dq:
                dq[k].append(np.abs(var[k].qpos[:, : ex.nq] - ex.qpos[:, : ex.nq]).max(1))
            for k, o in var.items():
                kkt[k].append(o.kkt.copy() if o.kkt is not None else np.zeros(n))
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
