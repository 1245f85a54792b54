"""Drop-in alias so code written against the reference keeps working unchanged:

    import gym_lowcostrobot                      # registers LiftCube-v0 ... StackTwoCubes-v0
    python -m rl_zoo3.train --gym-packages gym_lowcostrobot ...

Everything is implemented in gym_lowcostrobot_amd (HIP kernels behind a C ABI); this package only re-exports
the env classes under the reference's import path and performs the gymnasium registration.
"""
from gym_lowcostrobot_amd import __version__  # noqa: F401
from gym_lowcostrobot_amd.envs import register_envs

REGISTERED = register_envs(package="gym_lowcostrobot.envs")
