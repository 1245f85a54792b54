"""Latency of the single-env gymnasium facade (N = 1) per env.step() on the GPU box."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from gym_lowcostrobot_amd.envs import ReachCubeEnv, PickPlaceCubeEnv
for cls, kw in ((ReachCubeEnv, dict(observation_mode="state")), (PickPlaceCubeEnv, dict(observation_mode="state", action_mode="ee")), (ReachCubeEnv, dict(observation_mode="both"))):
    env = cls(**kw)
    env.reset(seed=0)
    a = env.action_space.sample()
    for _ in range(50): env.step(a)
    t0 = time.perf_counter()
    n = 500
    for _ in range(n):
        o, r, te, tr, info = env.step(a)
        if te or tr: env.reset()
    dt = time.perf_counter() - t0
    print(cls.__name__, kw, "%.1f us per step" % (1e6 * dt / n))
    env.close()
