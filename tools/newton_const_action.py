"""The Newton kernels under a CONSTANT action (arms run into the floor / their joint limits and stay there): step time, kernel iteration counts
against the oracle's from the same states.    python tools/newton_const_action.py [task] [n] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gym_lowcostrobot_amd import VecSim  # noqa: E402
from tests import util  # noqa: E402

task = sys.argv[1] if len(sys.argv) > 1 else "reach"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 16
big = VecSim(task, 65536, auto_reset=True, base_seed=0)
a_big = np.random.default_rng(0).uniform(-1, 1, (65536, big.action_dim)).astype(np.float32)
big.reset()
sim, o = util.make_pair(task, n, auto_reset=False, max_episode_steps=0)
seeds = np.arange(n, dtype=np.uint64)
o.reset(seeds=seeds)
a = a_big[:n]
for t in range(steps):
    big.timer_begin()
    big.step(a_big)
    ms = big.timer_end()
    util.sync_oracle_to_f32(o, carry=True)
    util.push_state(sim, o, carry=True)
    sim.step(a)
    o.step(a, 0)
    st = sim.get_state()
    dq = np.abs(st["qpos"].T - o.qpos[:, : sim.nq]).max(1)
    k = sim.max_sweeps.numpy().astype(int)
    oo = o.max_sweeps.astype(int)
    print(f"step {t:2d}: 65536-env step {ms:7.3f} ms (incl. host action upload) | n={n}: kernel its mean {k.mean():.2f} max {k.max()} >=10: {(k >= 10).mean():.4f}   oracle mean {oo.mean():.2f} max {oo.max()} >=10: {(oo >= 10).mean():.4f}"
          f" | |dq| p99 {np.percentile(dq, 99):.1e} max {dq.max():.1e} | limits active {((o.active_mask >> 18) > 0).mean():.2f} finger-floor {(((o.active_mask >> 14) & 3) > 0).mean():.2f}", flush=True)
    if os.environ.get("DEV_DUMP") and t == steps - 1:
        bad = np.argsort(-(k - oo))[:10]
        for e in bad:
            print("   env", e, "kernel", k[e], "oracle", oo[e], "mask %x" % o.active_mask[e], "dq %.1e" % dq[e])
