"""`gym_lowcostrobot.envs:<Class>` entry points (reference: gym_lowcostrobot/envs/__init__.py:1-8)."""
from gym_lowcostrobot_amd.envs import (  # noqa: F401
    LiftCubeEnv,
    PickPlaceCubeEnv,
    PushCubeEnv,
    PushCubeLoopEnv,
    ReachCubeEnv,
    StackTwoCubesEnv,
)

__all__ = ["LiftCubeEnv", "PickPlaceCubeEnv", "PushCubeEnv", "ReachCubeEnv", "StackTwoCubesEnv", "PushCubeLoopEnv"]
