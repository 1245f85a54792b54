"""`gym_lowcostrobot.envs:<Class>` entry points (reference: gym_lowcostrobot/envs/__init__.py:1-8).  Re-exports only: the classes
live in gym_lowcostrobot_amd.envs (ctypes facade over the HIP C ABI)."""
from .lift_cube_env import LiftCubeEnv
from .pick_place_cube_env import PickPlaceCubeEnv
from .push_cube_env import PushCubeEnv
from .push_cube_loop_env import PushCubeLoopEnv
from .reach_cube_env import ReachCubeEnv
from .stack_two_cubes_env import StackTwoCubesEnv

__all__ = ["LiftCubeEnv", "PickPlaceCubeEnv", "PushCubeEnv", "ReachCubeEnv", "StackTwoCubesEnv", "PushCubeLoopEnv"]
