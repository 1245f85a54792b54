"""gym_lowcostrobot.envs.wrappers (the reference directory has no __init__: namespace package there, regular package here)."""
