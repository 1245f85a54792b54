"""Which contact-solve iteration, swept until converged, reaches MuJoCo's optimum at which cost?  (round 5)

CPU oracle only.  From identical states incl. the carried forces, one control step with each candidate; reported per task:
|dqpos| against the exact optimum (primal Newton, orc_params.solver = 1) and the sweeps the candidate spent -- per env and
substep on average, and as a 64-env wave would pay them (the slowest lane of a wave decides: upper bound = sum over substeps of
the per-wave maximum is not available without per-substep records, so the per-wave maximum of the per-env sums is printed, a
lower bound of the wave-uniform cost).
    ORC_SWEEP_SUM=1 python tools/solver_modes_study.py [--n 512] [--steps 24] [--faithful 1]
"""
import argparse
import os
import sys

import numpy as np

os.environ.setdefault("ORC_SWEEP_SUM", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402

STATE = ("qpos", "qvel", "ee_lag", "target", "elapsed", "rng", "goal", "sim_time", "warm")


def measure(task, mode, n, steps, cands, base, seed=5, walk_kw={}):
    kw = dict(auto_reset=0, max_episode_steps=0, action_mode={"joint": 0, "ee": 1}[mode], **base)
    walk = orc.Oracle(task, n, **{**kw, **walk_kw})
    var = {k: orc.Oracle(task, n, **{**kw, **v}) for k, v in cands.items()}
    var["exact"] = orc.Oracle(task, n, solver=1, **kw)
    walk.reset(np.arange(n, dtype=np.uint64) + 77)
    rng = np.random.default_rng(seed)
    dq = {k: [] for k in cands}
    sw = {k: [] for k in cands}
    for t in range(steps):
        act = rng.uniform(-1, 1, (n, walk.action_dim)).astype(np.float32)
        if t >= 3:
            for o in var.values():
                for k in STATE:
                    getattr(o, k)[:] = getattr(walk, k)
                o.step(act, 0)
            ex = var["exact"]
            for k in cands:
                dq[k].append(np.abs(var[k].qpos[:, : ex.nq] - ex.qpos[:, : ex.nq]).max(1))
                sw[k].append(var[k].max_sweeps.copy())
        walk.step(act, 0)
    return {k: np.concatenate(v) for k, v in dq.items()}, {k: np.stack(v) for k, v in sw.items()}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=512)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--faithful", type=int, default=1)
    ap.add_argument("--adaptive", type=int, default=0)
    ap.add_argument("--exact_walk", type=int, default=0)
    ap.add_argument("--tasks", default="reach,push,lift,pick_place,stack,push_loop")
    a = ap.parse_args()
    cands = {"pg4": dict(cone=3, pgs_iters=4), "newton": dict(solver=2)}
    cands.update({f"newton{k}/{l}": dict(solver=2, newton_iters=k, ls_iters=l) for k, l in ((4, 8), (6, 8), (8, 4), (8, 6), (8, 8), (10, 8), (12, 8), (12, 12))})
    for tm in a.tasks.split(","):
        mode = "ee" if tm == "pick_place" else "joint"
        base = dict(condim6=2, cc_points=8) if a.faithful else {}
        dq, sw = measure(tm, mode, a.n, a.steps, cands, base, walk_kw=dict(solver=1) if a.exact_walk else {})
        print(f"{tm} ({mode}), {'six-row finger contacts everywhere, 8-point box-box' if a.faithful else 'round-4 contact model'}: {next(iter(dq.values())).size} env-steps")
        for k in cands:
            d, s = dq[k], sw[k].astype(float)
            nw = s.shape[1] // 64
            wave = s[:, : nw * 64].reshape(s.shape[0], nw, 64).max(2) if nw else s
            print(f"   {k:18s} |dqpos| p50 {np.median(d):.1e} p90 {np.percentile(d, 90):.1e} p99 {np.percentile(d, 99):.1e} max {d.max():.1e}"
                  f"   sweeps/substep: mean {s.mean() / 20:.1f}  p99 {np.percentile(s, 99) / 20:.1f}  per 64-env wave >= {wave.mean() / 20:.1f}", flush=True)
