"""Import path of the reference (`from gym_lowcostrobot.envs.pick_place_cube_env import PickPlaceCubeEnv`, e.g. gym_lowcostrobot/envs/__init__.py:2); the class is the
MI355X-backed facade of gym_lowcostrobot_amd.envs."""
from gym_lowcostrobot_amd.envs import PickPlaceCubeEnv  # noqa: F401

__all__ = ["PickPlaceCubeEnv"]
