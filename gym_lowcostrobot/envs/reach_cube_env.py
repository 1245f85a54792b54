"""Import path of the reference (`from gym_lowcostrobot.envs.reach_cube_env import ReachCubeEnv`, e.g. examples/hdf5_record.py:5); the class is the
MI355X-backed facade of gym_lowcostrobot_amd.envs."""
from gym_lowcostrobot_amd.envs import ReachCubeEnv  # noqa: F401

__all__ = ["ReachCubeEnv"]
