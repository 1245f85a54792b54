"""gym_lowcostrobot_amd -- MI355X-native batched simulator for the low-cost 6-DoF arm + cube tasks.

Drop-in for the hot path of perezjln/gym-lowcostrobot (batched reset/step of the five cube tasks);
see DESIGN.md / INTEGRATION.md.  Requires liblcr_hip.so (python -m gym_lowcostrobot_amd.build) and a
gfx950 GPU: there is no CPU fallback.
"""
__version__ = "0.1.0"

from .vecsim import VecSim  # noqa: F401
from . import envs  # noqa: F401,E402
from .envs import register_envs  # noqa: F401,E402
from .vecenv import LowCostRobotVecEnv, LowCostRobotVectorEnv  # noqa: F401,E402

register_envs()
