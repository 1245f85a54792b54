"""where a 65 536-env LowCostRobotVecEnv.step spends its time (GPU box)"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from gym_lowcostrobot_amd import LowCostRobotVecEnv
n = 65536
for preset in ("faithful", "fast"):
    v = LowCostRobotVecEnv("reach", n, seed=0, preset=preset)
    v.reset()
    a = np.random.default_rng(0).uniform(-1, 1, (n, 5)).astype(np.float32)
    for t in range(14):
        t0 = time.perf_counter(); v.sim.step(a); v.sim.L.lcr_sync(v.sim.handle) if hasattr(v.sim.L, "lcr_sync") else None
        t1 = time.perf_counter(); h = v.sim.fetch_host()
        t2 = time.perf_counter(); obs = v._obs_from(h)
        t3 = time.perf_counter()
        dones = h["terminated"] | h["truncated"]
        print(f"{preset} step {t}: sim.step {1e3*(t1-t0):.2f} ms, fetch_host {1e3*(t2-t1):.2f} ms, obs {1e3*(t3-t2):.2f} ms, finished {int(dones.sum())} did_reset {int(h['did_reset'].sum())}", flush=True)
    t0 = time.perf_counter()
    for _ in range(5): v.step(a)
    print(preset, "full step", (time.perf_counter() - t0) / 5 * 1e3, "ms")
    v.close()
