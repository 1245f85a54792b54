"""Import path of the reference (`from gym_lowcostrobot.envs.lift_cube_env import LiftCubeEnv`, e.g. gym_lowcostrobot/envs/__init__.py:1); the class is the
MI355X-backed facade of gym_lowcostrobot_amd.envs."""
from gym_lowcostrobot_amd.envs import LiftCubeEnv  # noqa: F401

__all__ = ["LiftCubeEnv"]
