"""Census of the parity outliers by class and task: the loop of tests/test_gpu_parity.py::test_step_rollout_vs_oracle -- free-running rollouts of
the HIP step against the fp64 oracle from the same seeds, every env outside the tolerance (|dqpos| 2e-5, |dqvel| 2e-3 per control step) has to be explained inside
tests/util.py::parity_step -- run per task at a larger size, cold and carried forces, and the explanations counted:
    flip        the discrete-decision signature (which vertices / faces / slots are in contact) differs between kernel and oracle
    fp32-twin   the oracle's own fp32 build leaves the tolerance against its fp64 build at that env
    family      (preset fast only) the other kernel family leaves the tolerance as well
    sensitive   the fp64 step map spreads two-ulp input noise beyond a quarter of the tolerance
    cap         a Newton solve ran into its iteration budget on either side
    unexplained must be zero
GPU box:    python tools/parity_census.py [n] [steps] > profiles/rNN_parity_census.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import util  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
print(f"parity census, preset faithful (the default), {n} envs x {steps} control steps per row, random actions in [-1.2, 1.2]; tolerance |dqpos| 2e-5, |dqvel| 2e-3")
print(f"{'task':12s} {'mode':5s} {'forces':6s} {'env-steps':>9s} {'outside':>8s} {'flip':>6s} {'fp32-twin':>9s} {'family':>6s} {'sensitive':>9s} {'cap':>5s} {'unexplained':>11s} {'max |dq|':>9s} {'max |dv|':>9s}")
tot = {}
for task, mode in (("reach", "joint"), ("push", "joint"), ("lift", "joint"), ("pick_place", "ee"), ("stack", "joint"), ("push_loop", "joint")):
    for carry in (False, True):
        util.CARRY_DEFAULT = carry
        for k in list(util.STATS):
            util.STATS[k] = 0 if not isinstance(util.STATS[k], dict) else {}
        util.STATS["max_dq"] = util.STATS["max_dv"] = 0.0
        sim, o = util.make_pair(task, n, action_mode=mode, auto_reset=False, max_episode_steps=0)
        seeds = np.arange(n, dtype=np.uint64) + 100
        o.reset(seeds=seeds); sim.reset(seeds=seeds)
        rng = np.random.default_rng(7)
        err = ""
        try:
            for t in range(steps):
                a = rng.uniform(-1.2, 1.2, (n, sim.action_dim)).astype(np.float32)
                util.parity_step(sim, o, a, 2e-5, 2e-3, where=(task, t))
        except AssertionError as e:   # (an unexplained env or one beyond the hard bounds: reported, the census goes on)
            err = "  ASSERT " + str(e)[:120]
        S = util.STATS
        fam, sens, cap = S.get("out_family", 0), S.get("out_sens", 0), S.get("out_cap", 0)
        twin = S["out_illcond"] - fam - sens - cap
        unexpl = S["out"] - S["out_flip"] - S["out_illcond"]
        print(f"{task:12s} {mode:5s} {'carry' if carry else 'cold':6s} {S['envs']:9d} {S['out']:8d} {S['out_flip']:6d} {twin:9d} {fam:6d} {sens:9d} {cap:5d} {unexpl:11d} {S['max_dq']:9.1e} {S['max_dv']:9.1e}{err}", flush=True)
        for k, v in (("envs", S["envs"]), ("out", S["out"]), ("flip", S["out_flip"]), ("twin", twin), ("sens", sens), ("cap", cap), ("unexpl", unexpl)):
            tot[k] = tot.get(k, 0) + v
        sim.close()
print(f"total: {tot['envs']} env-steps, {tot['out']} outside the tolerance ({100.0 * tot['out'] / max(tot['envs'], 1):.3f} %): flip {tot['flip']}, fp32-twin {tot['twin']}, sensitive {tot['sens']}, cap {tot['cap']}, unexplained {tot['unexpl']}")
