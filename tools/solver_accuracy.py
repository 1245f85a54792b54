"""Accuracy of the default contact solve (4 warm-started sweeps, deviation D1) against the exact optimum of the same model (primal Newton,
orc_params.solver = 1), measured with the CPU oracle on contact-rich random rollouts.  All variants start every control step from the same
state; the difference after ONE control step (20 substeps) is reported for
  cold     the forces start from zero at every control step (LCR_COMPAT_COLD_SOLVE_EACH_STEP; the only mode before round 2), and
  carried  the forces of the last substep warm-start the next control step (default; MuJoCo's qacc_warmstart does the same).
python tools/solver_accuracy.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc

COLD = 2   # ORC_COMPAT_COLD_SOLVE_EACH_STEP


def measure(task, n=512, steps=50, seed=3):
    cold = orc.Oracle(task, n, preset="fast", pgs_iters=4, compat=COLD, auto_reset=0, max_episode_steps=0)
    carried = orc.Oracle(task, n, preset="fast", pgs_iters=4, compat=0, auto_reset=0, max_episode_steps=0)
    # reference: the EXACT optimum of every substep's convex problem (primal Newton, orc_params.solver = 1, certified by orc_io.kkt) -- rounds 1-3 compared
    # against 300 cold sweeps of the same iteration, whose fixed point was not the optimum (tools/kkt_distance.py)
    ref = orc.Oracle(task, n, preset="fast", solver=1, auto_reset=0, max_episode_steps=0)
    seeds = np.arange(n, dtype=np.uint64) + 1000
    for o in (cold, carried, ref):
        o.reset(seeds)
    rng = np.random.default_rng(seed)
    out = {"cold": [], "carried": []}
    for t in range(steps):
        act = rng.uniform(-1, 1, (n, ref.action_dim)).astype(np.float32)
        for o in (cold, carried):                                     # same start state (the carried forces stay with `carried`)
            o.qpos[:] = ref.qpos; o.qvel[:] = ref.qvel; o.ee_lag[:] = ref.ee_lag
        for o in (cold, carried, ref):
            o.step(act, 0)
        if t >= 2:
            for name, o in (("cold", cold), ("carried", carried)):
                out[name].append(np.abs(o.qpos[:, : o.nq] - ref.qpos[:, : o.nq]).max(1))
    return {k: np.concatenate(v) for k, v in out.items()}


if __name__ == "__main__":
    pct = lambda x: "median %.1e  95%% %.1e  99.9%% %.1e  max %.1e" % (np.median(x), np.percentile(x, 95), np.percentile(x, 99.9), x.max())
    for task in ("reach", "push", "lift", "stack", "push_loop"):
        r = measure(task)
        print(f"{task:9s} |dqpos| vs converged, cold   : {pct(r['cold'])}\n          |dqpos| vs converged, carried: {pct(r['carried'])}")
