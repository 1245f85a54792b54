"""Import path of the reference (`from gym_lowcostrobot.envs.push_cube_env import PushCubeEnv`, e.g. examples/gym_manipulation_img_multi.py:3); the class is the
MI355X-backed facade of gym_lowcostrobot_amd.envs."""
from gym_lowcostrobot_amd.envs import PushCubeEnv  # noqa: F401

__all__ = ["PushCubeEnv"]
