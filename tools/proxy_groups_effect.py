"""Deviation D3, measured with the oracle: what would one contact PER PROXY GROUP (each end of link_3 on its own, wrist / gripper body
together: orc_params.proxy_groups = 3) change against the product's one shared contact (the deepest candidate, proxy_groups = 1)?
    python tools/proxy_groups_effect.py [n_envs] [steps]
Random policy on ReachCube; reports (a) how deep the lowest proxy point gets below the floor in either model and (b) how far one control
step from the SAME state differs between the two, in the env-steps where a proxy touches at all."""
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import orc  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = np.random.default_rng(0)
sims = {g: orc.Oracle("reach", n, proxy_groups=g) for g in (1, 3)}
for o in sims.values():
    o.reset(seeds=np.arange(n))
depth = {1: [], 3: []}
two_ends = 0
touch_steps = 0
dq_touch = []
for t in range(steps):
    a = rng.uniform(-1, 1, (n, 5)).astype(np.float32)
    # (b) one step of the 3-group model from the 1-group model's state
    probe = sims[3]
    keep = {k: getattr(probe, k).copy() for k in ("qpos", "qvel", "ee_lag", "elapsed", "rng", "warm")}
    for k in ("qpos", "qvel", "ee_lag", "elapsed", "rng"):
        getattr(probe, k)[:] = getattr(sims[1], k)
    probe.warm[:] = 0
    w1 = sims[1].warm.copy()
    sims[1].warm[:] = 0
    q_before = sims[1].qpos.copy()
    probe.step(a, threads=0)
    q3 = probe.qpos.copy()
    m3 = probe.active_mask.copy()
    for k, v in keep.items():
        getattr(probe, k)[:] = v
    st = {k: getattr(sims[1], k).copy() for k in ("qpos", "qvel", "ee_lag", "elapsed", "rng")}
    sims[1].step(a, threads=0)
    q1 = sims[1].qpos.copy()
    touched = ((sims[1].active_mask >> 16) & 1).astype(bool) | (((m3 >> 16) & 3) != 0) | (((m3 >> 28) & 1) != 0)
    live = touched & (sims[1].did_reset == 0)
    dq_touch.extend(np.abs(q3[live, :6] - q1[live, :6]).max(axis=1))
    touch_steps += int(live.sum())
    two_ends += int(((((m3 >> 16) & 1) + ((m3 >> 17) & 1) + ((m3 >> 28) & 1)) >= 2)[live].sum())
    # restore the 1-group sim's own carried forces (its free-running trajectory continues), step the free-running 3-group sim
    sims[3].step(a, threads=0)
    if t % 5 == 4:
        for g, o in sims.items():
            for e in range(0, n, 4):
                c, r = orc.proxies(o.qpos[e, :6])
                depth[g].append(float((c[:, 2] - r).min()))
dq = np.array(dq_touch)
print(f"env-steps with a proxy contact: {touch_steps} of {n * steps} ({100.0 * touch_steps / (n * steps):.2f} %); two or three groups active at once in "
      f"{100.0 * two_ends / max(touch_steps, 1):.1f} % of them")
print(f"one control step from the same state, 3 groups vs 1 (arm |dq|, where a proxy touches): median {np.median(dq):.2e}, p90 {np.percentile(dq, 90):.2e}, "
      f"p99 {np.percentile(dq, 99):.2e}, max {dq.max():.2e}")
for g in (1, 3):
    d = np.array(depth[g])
    print(f"proxy_groups={g}: {len(d)} sampled states; lowest proxy point below the floor by > 1 mm in {100 * (d < -1e-3).mean():.2f} %, > 5 mm in "
          f"{100 * (d < -5e-3).mean():.3f} %, > 15 mm in {100 * (d < -15e-3).mean():.4f} %; deepest {1e3 * -d.min():.1f} mm")
