#!/usr/bin/env python3
"""Random-action roll-out of one environment through the gymnasium-style facade (the usage pattern of the reference's
README / examples/gym_manipulation.py), running on the MI355X library.

    python examples/random_rollout.py --env PushCube-v0 --steps 100
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gym_lowcostrobot  # noqa: E402,F401  (registers the ids when gymnasium is installed)
from gym_lowcostrobot_amd import envs  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--env", default="ReachCube-v0", choices=sorted(envs.REGISTRY))
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--observation-mode", default="state")
    ap.add_argument("--action-mode", default="joint")
    args = ap.parse_args()
    env = getattr(envs, envs.REGISTRY[args.env])(observation_mode=args.observation_mode, action_mode=args.action_mode)
    obs, info = env.reset(seed=0)
    rng = np.random.default_rng(0)
    ret = 0.0
    for t in range(args.steps):
        action = rng.uniform(-1, 1, env.action_space.shape).astype(np.float32)
        obs, reward, terminated, truncated, info = env.step(action)
        ret += float(reward)
        if terminated or truncated or (t + 1) % envs.MAX_EPISODE_STEPS == 0:  # TimeLimit(50) comes from gymnasium.make
            obs, info = env.reset()
    print(f"{args.env}: {args.steps} steps, return {ret:.3f}, last obs keys {sorted(obs)}")
    env.close()


if __name__ == "__main__":
    main()
