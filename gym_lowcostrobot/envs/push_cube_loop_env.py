"""Import path of the reference (`from gym_lowcostrobot.envs.push_cube_loop_env import PushCubeLoopEnv`, e.g. gym_lowcostrobot/envs/__init__.py:6); the class is the
MI355X-backed facade of gym_lowcostrobot_amd.envs."""
from gym_lowcostrobot_amd.envs import PushCubeLoopEnv  # noqa: F401

__all__ = ["PushCubeLoopEnv"]
