"""Import path of the reference (`from gym_lowcostrobot.envs.stack_two_cubes_env import StackTwoCubesEnv`, e.g. gym_lowcostrobot/envs/__init__.py:5); the class is the
MI355X-backed facade of gym_lowcostrobot_amd.envs."""
from gym_lowcostrobot_amd.envs import StackTwoCubesEnv  # noqa: F401

__all__ = ["StackTwoCubesEnv"]
