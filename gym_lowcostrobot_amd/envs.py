"""Single-environment gymnasium.Env facades over the batched HIP simulator (N = 1 view).

Same class names, constructor kwargs and defaults, spaces, return types and exception types as the
reference env classes (gym_lowcostrobot/envs/*_env.py); physics runs in the fused HIP step kernel through
the C ABI.  Differences a caller can observe are listed in INTEGRATION.md (no `model` / `data` MuJoCo
objects -> get_state()/set_state(); image observations and render() come from an approximate ray-caster).
"""
import os

import numpy as np

from . import spaces as sp
from .vecsim import VecSim

class _LowCostRobotEnv(sp.EnvBase):
    # reach_cube_env.py:75
    metadata = {"render_modes": ["human", "rgb_array"], "render_fps": 25}
    _task = None

    def _setup(self, observation_mode, action_mode, reward_type, block_gripper, render_mode, n_substeps, **task_kw):
        if action_mode not in ("joint", "ee"):
            raise ValueError("Invalid action mode, must be 'ee' or 'joint'")  # reach_cube_env.py:269-270 (raised at first step there)
        assert render_mode is None or render_mode in self.metadata["render_modes"]  # reach_cube_env.py:120
        if observation_mode not in ("image", "state", "both"):
            raise ValueError(f"invalid observation_mode {observation_mode!r}")
        self.observation_mode = observation_mode
        self.action_mode = action_mode
        self.reward_type = reward_type
        self.block_gripper = block_gripper
        self.render_mode = render_mode
        self.num_dof = 6
        self.control_decimation = n_substeps
        action_shape = {"joint": 5, "ee": 3}[action_mode] + (0 if block_gripper else 1)  # reach_cube_env.py:95-96
        self.action_space = sp.Box(low=-1.0, high=1.0, shape=(action_shape,), dtype=np.float32)
        subs = {  # reach_cube_env.py:105-115
            "arm_qpos": sp.Box(low=-np.pi, high=np.pi, shape=(6,), dtype=np.float32),
            "arm_qvel": sp.Box(low=-10.0, high=10.0, shape=(6,), dtype=np.float32),
        }
        if self._task in ("push", "pick_place"):
            subs["target_pos"] = sp.Box(low=-10.0, high=10.0, shape=(3,), dtype=np.float32)  # push_cube_env.py:111
        if observation_mode in ("image", "both"):
            subs["image_front"] = sp.Box(0, 255, shape=(240, 320, 3), dtype=np.uint8)
            subs["image_top"] = sp.Box(0, 255, shape=(240, 320, 3), dtype=np.uint8)
        if observation_mode in ("state", "both"):
            if self._task == "stack":  # stack_two_cubes_env.py:115-116
                subs["cube_red_pos"] = sp.Box(low=-10.0, high=10.0, shape=(3,), dtype=np.float32)
                subs["cube_blue_pos"] = sp.Box(low=-10.0, high=10.0, shape=(3,), dtype=np.float32)
            else:
                subs["cube_pos"] = sp.Box(low=-10.0, high=10.0, shape=(3,), dtype=np.float32)
        self.observation_space = sp.Dict(subs)
        # TimeLimit(50) is added by gymnasium.make from the registration (gym_lowcostrobot/__init__.py:12);
        # the bare class never truncates (reach_cube_env.py:331) and never auto-resets
        self._sim = VecSim(
            self._task, 1, observation_mode=observation_mode, action_mode=action_mode, reward_type=reward_type,
            block_gripper=block_gripper, n_substeps=n_substeps, max_episode_steps=0, auto_reset=False,
            base_seed=int.from_bytes(os.urandom(7), "little"), device=int(os.environ.get("LCR_DEVICE", "0")), **task_kw,
        )

    # ---- gymnasium.Env protocol ----
    def get_observation(self):
        obs = self._sim.observations()
        keys = self.observation_space.spaces  # gymnasium.spaces.Dict has no key-membership __contains__; .spaces is the mapping
        # (keys in the reference's insertion order: arm_qpos, arm_qvel, [target_pos], [images], cube position(s) -- push_cube_env.py:293-306)
        return {k: (obs[k][0] if obs[k].ndim >= 2 else obs[k]) for k in keys if k in obs}

    def reset(self, seed=None, options=None):
        try:
            super().reset(seed=seed, options=options)  # seeds gymnasium's own np_random when present
        except TypeError:
            pass
        self._sim.reset(seeds=None if seed is None else [int(seed)])
        return self.get_observation(), {}

    def step(self, action):
        if np.array(action).shape != self.action_space.shape:
            raise ValueError("Action dimension mismatch")  # reach_cube_env.py:231-232
        self._sim.step(np.asarray(action, np.float32)[None, :])
        out = self._sim.outputs()
        observation = self.get_observation()
        terminated = np.bool_(out["terminated"][0])
        r = out["reward"][0]
        if self._task == "lift":  # lift_cube_env.py:336-345: dense float64, info = {}
            div = bool(out["did_reset"][0])
            return observation, np.float64(r), False, div, ({"diverged": True} if div else {})
        reward = np.float32(r) if self.reward_type == "sparse" else np.float64(r)  # reach_cube_env.py:345-348
        info = {"is_success": np.bool_(out["is_success"][0])}
        # the bare class never truncates (reach_cube_env.py:331).  The one exception is the kernel's divergence guard (the
        # analogue of MuJoCo's mj_checkPos/mj_checkVel auto-reset): it re-initialises the env in place, so the episode
        # boundary is reported instead of being swallowed
        diverged = bool(out["did_reset"][0])
        if diverged:
            info["diverged"] = True
        return observation, reward, terminated, diverged, info

    def render(self):
        if self.render_mode == "rgb_array":  # 640x640 frame of camera_vizu (reach_cube_env.py:350-355), ray-cast approximation
            return self._sim.render(0, "camera_vizu", 640, 640)
        return None

    def close(self):
        if getattr(self, "_sim", None) is not None:
            self._sim.close()
            self._sim = None

    # ---- replacements for poking env.data.qpos / env.data.qvel ----
    def get_state(self):
        st = self._sim.get_state()
        return {k: (v[..., 0] if v.ndim > 1 else v[0]) for k, v in st.items()}

    def set_state(self, qpos=None, qvel=None):
        self._sim.set_state(qpos=None if qpos is None else np.asarray(qpos, np.float64)[:, None],
                            qvel=None if qvel is None else np.asarray(qvel, np.float64)[:, None])


class ReachCubeEnv(_LowCostRobotEnv):
    """ReachCube-v0 (reference: envs/reach_cube_env.py:77-87)."""
    _task = "reach"

    def __init__(self, observation_mode="image", action_mode="joint", reward_type="sparse", block_gripper=True,
                 distance_threshold=0.05, cube_xy_range=0.3, n_substeps=20, render_mode=None):
        self.distance_threshold = distance_threshold
        self.cube_xy_range = cube_xy_range
        self._setup(observation_mode, action_mode, reward_type, block_gripper, render_mode, n_substeps,
                    distance_threshold=distance_threshold, cube_xy_range=cube_xy_range)


class LiftCubeEnv(_LowCostRobotEnv):
    """LiftCube-v0 (reference: envs/lift_cube_env.py:77-88)."""
    _task = "lift"

    def __init__(self, observation_mode="image", action_mode="joint", reward_type="sparse", block_gripper=False,
                 distance_threshold=0.05, height_threshold=0.1, cube_xy_range=0.3, n_substeps=20, render_mode=None):
        self.distance_threshold = distance_threshold
        self.height_threshold = height_threshold
        self.cube_xy_range = cube_xy_range
        self._setup(observation_mode, action_mode, reward_type, block_gripper, render_mode, n_substeps,
                    distance_threshold=distance_threshold, cube_xy_range=cube_xy_range, height_threshold=height_threshold)


class PushCubeEnv(_LowCostRobotEnv):
    """PushCube-v0 (reference: envs/push_cube_env.py:79-90)."""
    _task = "push"

    def __init__(self, observation_mode="image", action_mode="joint", reward_type="sparse", block_gripper=True,
                 distance_threshold=0.05, cube_xy_range=0.3, target_xy_range=0.3, n_substeps=20, render_mode=None):
        self.distance_threshold = distance_threshold
        self.cube_xy_range = cube_xy_range
        self.target_xy_range = target_xy_range
        self._setup(observation_mode, action_mode, reward_type, block_gripper, render_mode, n_substeps,
                    distance_threshold=distance_threshold, cube_xy_range=cube_xy_range, target_xy_range=target_xy_range)


class PickPlaceCubeEnv(_LowCostRobotEnv):
    """PickPlaceCube-v0 (reference: envs/pick_place_cube_env.py:79-91)."""
    _task = "pick_place"

    def __init__(self, observation_mode="image", action_mode="joint", reward_type="sparse", block_gripper=False,
                 distance_threshold=0.05, cube_xy_range=0.3, target_xy_range=0.3, goal_z_range=0.1, n_substeps=20,
                 render_mode=None):
        self.distance_threshold = distance_threshold
        self.cube_xy_range = cube_xy_range
        self.target_xy_range = target_xy_range
        self.goal_z_range = goal_z_range
        self._setup(observation_mode, action_mode, reward_type, block_gripper, render_mode, n_substeps,
                    distance_threshold=distance_threshold, cube_xy_range=cube_xy_range, target_xy_range=target_xy_range,
                    goal_z_range=goal_z_range)


class PushCubeLoopEnv(_LowCostRobotEnv):
    """PushCubeLoop-v0 (reference: envs/push_cube_loop_env.py:76-139): the cube is shuttled between two goal regions inside
    four rails; the goal side switches on success and persists across resets; never terminates."""
    _task = "push_loop"

    def __init__(self, observation_mode="image", action_mode="joint", block_gripper=True, n_substeps=20, render_mode=None):
        self._setup(observation_mode, action_mode, "sparse", block_gripper, render_mode, n_substeps)

    @property
    def current_goal(self):
        return int(self._sim.current_goal.numpy()[0])

    def reset(self, seed=None, options=None):
        obs, _ = super().reset(seed=seed, options=options)
        return obs, {"timestamp": 0.0}  # push_cube_loop_env.py:317

    def step(self, action):
        if np.array(action).shape != self.action_space.shape:
            raise ValueError("Action dimension mismatch")
        self._sim.step(np.asarray(action, np.float32)[None, :])
        out = self._sim.outputs()
        info = {"timestamp": float(self._sim.timestamp.numpy()[0]), "success": int(out["is_success"][0])}  # :328
        div = bool(out["did_reset"][0])
        if div:
            info["diverged"] = True
        # reward types as the reference's arithmetic leaves them (push_cube_loop_env.py:340-357): the int 5 on success, the int -2 when
        # `min(max(x, -2), -1)` clips, numpy.float64 otherwise
        r = float(out["reward"][0])
        reward = 5 if info["success"] else (-2 if r <= -2.0 else np.float64(r))
        return self.get_observation(), reward, False, div, info


class StackTwoCubesEnv(_LowCostRobotEnv):
    """StackTwoCubes-v0 (reference: envs/stack_two_cubes_env.py:78-88)."""
    _task = "stack"

    def __init__(self, observation_mode="image", action_mode="joint", reward_type="sparse", block_gripper=False,
                 distance_threshold=0.05, cube_xy_range=0.3, n_substeps=20, render_mode=None):
        self.distance_threshold = distance_threshold
        self.cube_xy_range = cube_xy_range
        self._setup(observation_mode, action_mode, reward_type, block_gripper, render_mode, n_substeps,
                    distance_threshold=distance_threshold, cube_xy_range=cube_xy_range)


__all__ = ["LiftCubeEnv", "PickPlaceCubeEnv", "PushCubeEnv", "ReachCubeEnv", "StackTwoCubesEnv", "PushCubeLoopEnv"]

# registry ids of the reference (gym_lowcostrobot/__init__.py:9-43)
REGISTRY = {
    "LiftCube-v0": "LiftCubeEnv",
    "PickPlaceCube-v0": "PickPlaceCubeEnv",
    "PushCube-v0": "PushCubeEnv",
    "ReachCube-v0": "ReachCubeEnv",
    "StackTwoCubes-v0": "StackTwoCubesEnv",
    "PushCubeLoop-v0": "PushCubeLoopEnv",
}
MAX_EPISODE_STEPS = 50


def register_envs(package="gym_lowcostrobot_amd.envs"):
    """Register the six ids with gymnasium (no-op when gymnasium is not installed)."""
    if not sp.HAVE_GYMNASIUM:
        return []
    from gymnasium.envs.registration import register, registry

    done = []
    for env_id, cls in REGISTRY.items():
        if env_id not in registry:
            register(id=env_id, entry_point=f"{package}:{cls}", max_episode_steps=MAX_EPISODE_STEPS)
        done.append(env_id)
    return done
