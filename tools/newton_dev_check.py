"""Development check of the Newton kernels on the GPU: re-synchronised parity against the oracle's newton_product (carry mode), iteration
counts, step time.    python tools/newton_dev_check.py [tasks] [n] [steps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import orc  # noqa: E402
from tests import util  # noqa: E402


def check(task, mode, n, steps, preset="faithful"):
    extra = {k: (float(v) if "tol" in k else int(v)) for k, v in (kv.split("=") for kv in os.environ.get("DEV_KW", "").split(",") if kv)}   # e.g. DEV_KW=ls_iters=16,ls_tol=1e-4
    sim, o = util.make_pair(task, n, action_mode=mode, auto_reset=False, max_episode_steps=0, preset=preset, **extra)
    seeds = np.arange(n, dtype=np.uint64) + 11
    o.reset(seeds=seeds)
    rng = np.random.default_rng(3)
    DQ, IT, OIT = [], [], []
    for t in range(steps):
        util.sync_oracle_to_f32(o, carry=True)
        util.push_state(sim, o, carry=True)
        a = rng.uniform(-1, 1, (n, sim.action_dim)).astype(np.float32)
        sim.step(a)
        o.step(a, 0)
        st = sim.get_state()
        dq = np.abs(st["qpos"].T - o.qpos[:, : sim.nq]).max(1)
        DQ.append(dq)
        IT.append(sim.max_sweeps.numpy() if sim.max_sweeps is not None else np.zeros(n))
        OIT.append(o.max_sweeps.copy())
        if t == steps - 1 and os.environ.get("DEV_DUMP"):
            k = IT[-1].astype(int); oo = OIT[-1].astype(int)
            bad = np.where(k >= 8)[0][:12]
            am = sim.active_mask.numpy()
            for e in bad:
                print("   env", e, "kernel its", k[e], "oracle its", oo[e], "mask %x" % am[e], "oracle mask %x" % o.active_mask[e], "dq %.1e" % dq[e])
            print("   hist kernel", np.bincount(k, minlength=11), "oracle", np.bincount(oo, minlength=11))
    dq = np.concatenate(DQ)
    print(f"{task:10s} {mode:5s} {preset}: |dq| p50 {np.median(dq):.1e} p90 {np.percentile(dq, 90):.1e} p99 {np.percentile(dq, 99):.1e} max {np.nanmax(dq):.1e}"
          f"  within 2e-5: {100 * (dq <= 2e-5).mean():.2f} %  finite: {np.isfinite(dq).all()}   newton its (max per substep) kernel mean {np.mean(IT):.2f} max {np.max(IT)}"
          f" oracle mean {np.mean(OIT):.2f} max {np.max(OIT)}", flush=True)
    sim.close()


if __name__ == "__main__":
    tasks = (sys.argv[1] if len(sys.argv) > 1 else "reach,push,lift,pick_place").split(",")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    for t in tasks:
        check(t, "ee" if t == "pick_place" else "joint", n, steps)
