"""VecSim: N independent low-cost-robot environments stepped in lockstep on one MI355X.

Thin host-side mirror of the reference env classes' reset()/step() for a batch (reference:
gym_lowcostrobot/envs/reach_cube_env.py:297-333 and the four sibling files).  All arithmetic happens in
the HIP kernels behind the C ABI (include/lcr.h); this file only owns handles and moves bytes.
"""
import ctypes
import os

import numpy as np

from . import _capi
from ._capi import ACTION_MODES, OBS_MODES, REWARD_TYPES, TASKS, LcrConfig, LcrHostView, LcrObsView, LcrOutView, check


def _vp(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


class DeviceArray:
    """A typed view of device memory exposing __cuda_array_interface__ (zero-copy into torch on ROCm)."""

    def __init__(self, sim, ptr, shape, dtype, readonly=True):
        self._sim = sim  # keeps the owner alive
        self.ptr = int(ptr)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape)) * self.dtype.itemsize
        self.__cuda_array_interface__ = {
            "shape": self.shape,
            "typestr": self.dtype.str,
            "data": (self.ptr, False),  # torch rejects read-only exports; observation views are read-only by convention
            "version": 3,
            "strides": None,
        }

    def numpy(self):
        out = np.empty(self.shape, self.dtype)
        check(self._sim.L.lcr_memcpy_d2h(self._sim.handle, _vp(out), ctypes.c_void_p(self.ptr), self.nbytes))
        return out

    def torch(self):
        import torch

        return torch.as_tensor(self, device=f"cuda:{self._sim.device}")


class VecSim:
    def __init__(
        self,
        task,
        n_envs=1,
        *,
        device=0,
        env_id_offset=0,
        observation_mode="state",
        action_mode="joint",
        reward_type="sparse",
        block_gripper=None,
        distance_threshold=0.05,
        cube_xy_range=0.3,
        target_xy_range=0.3,
        goal_z_range=0.1,
        height_threshold=0.1,
        n_substeps=20,
        max_episode_steps=50,
        impratio=100.0,
        pgs_iters=None,
        compat=0,
        auto_reset=True,
        base_seed=0,
        arm_collision=True,
        pgs_tol=1e-6,
        diagnostics=False,
        finger_cube_condim=None,
        step_kernel="auto",
        cc_points=None,
        global_envs=None,
        profile=None,
        preset=None,
        solver=None,
        newton_iters=None,
        ls_iters=None,
        newton_tol=None,
        ls_tol=None,
        finger_floor_condim=None,
    ):
        self.L = _capi.load()
        if action_mode not in ACTION_MODES:
            raise ValueError("Invalid action mode, must be 'ee' or 'joint'")  # reach_cube_env.py:269-270
        if observation_mode not in OBS_MODES:
            raise ValueError(f"invalid observation_mode {observation_mode!r}")
        if reward_type not in REWARD_TYPES:
            raise ValueError(f"invalid reward_type {reward_type!r}")
        self.task_name = task if isinstance(task, str) else {v: k for k, v in TASKS.items()}[task]
        cfg = LcrConfig()
        # preset: "faithful" (the reference's contact model solved by Newton's method) | "fast" (rounds 1-4: four sweeps, fewer rows); None = the library's default
        if preset is None:
            preset = os.environ.get("LCR_PRESET") or None   # (what the single-env facade classes, whose constructors are the reference's, can be switched with)
        if preset is None:
            check(self.L.lcr_config_default(ctypes.byref(cfg), TASKS[self.task_name]))
        else:
            if preset not in _capi.PRESETS:
                raise ValueError(f"invalid preset {preset!r} (faithful | fast)")
            check(self.L.lcr_config_preset(ctypes.byref(cfg), TASKS[self.task_name], _capi.PRESETS[preset]))
        cfg.n_envs = int(n_envs)
        cfg.device = int(device)
        cfg.env_id_offset = int(env_id_offset)
        cfg.action_mode = ACTION_MODES[action_mode]
        cfg.obs_mode = OBS_MODES[observation_mode]
        cfg.reward_type = REWARD_TYPES[reward_type]
        cfg.block_gripper = -1 if block_gripper is None else int(bool(block_gripper))
        cfg.distance_threshold = distance_threshold
        cfg.cube_xy_range = cube_xy_range
        cfg.target_xy_range = target_xy_range
        cfg.goal_z_range = goal_z_range
        cfg.height_threshold = height_threshold
        cfg.impratio = impratio
        cfg.n_substeps = int(n_substeps)
        cfg.max_episode_steps = int(max_episode_steps)
        if pgs_iters is not None:
            cfg.pgs_iters = int(pgs_iters)
        if solver is not None:
            if solver not in _capi.SOLVERS:
                raise ValueError(f"invalid solver {solver!r} (pgs | newton)")
            cfg.solver = _capi.SOLVERS[solver]
        for name, val in (("newton_iters", newton_iters), ("ls_iters", ls_iters), ("finger_floor_condim", finger_floor_condim)):
            if val is not None:
                setattr(cfg, name, int(val))
        for name, val in (("newton_tol", newton_tol), ("ls_tol", ls_tol)):
            if val is not None:
                setattr(cfg, name, float(val))
        cfg.compat = int(compat)
        cfg.auto_reset = int(bool(auto_reset))
        cfg.base_seed = int(base_seed)
        cfg.arm_collision = int(bool(arm_collision))
        cfg.pgs_tol = float(pgs_tol)
        # diagnostics is a boolean here (the decision signature + data.ctrl of every step); the profiling aids that reuse those arrays as
        # per-wave cycle counters are a separate, named argument: profile="wave_cycles" (one-wave kernels) | "phase_cycles" (two-wave kernels)
        if diagnostics not in (False, True, 0, 1):
            raise ValueError(f"diagnostics must be a bool, got {diagnostics!r} (cycle read-back: profile='wave_cycles' | 'phase_cycles')")
        if profile not in _capi.PROFILE_MODES:
            raise ValueError(f"invalid profile {profile!r} (None | 'wave_cycles' | 'phase_cycles')")
        cfg.diagnostics = _capi.PROFILE_MODES[profile] if profile is not None else int(bool(diagnostics))
        if finger_cube_condim is not None:   # default: lcr_config_default's choice for the task (6 for PushCubeLoop, else 4)
            cfg.finger_cube_condim = int(finger_cube_condim)
        if step_kernel not in _capi.STEP_KERNELS:
            raise ValueError(f"invalid step_kernel {step_kernel!r} (auto | single | coop)")
        cfg.step_kernel = _capi.STEP_KERNELS[step_kernel]   # kernel family: "auto" = by task and JOB size (global_envs), never by shard size
        if global_envs is not None:   # envs of the whole job this sim is a shard of: every shard declares the same number -> same family, same bits
            cfg.global_envs = int(global_envs)
        if cc_points is not None:   # StackTwoCubes: 4 (default) or 8 cube<->cube manifold points
            cfg.cc_points = int(cc_points)
        self.cfg = cfg
        self.n = int(n_envs)
        self.device = int(device)
        self.action_dim = check(self.L.lcr_action_dim(ctypes.byref(cfg)))
        self.nq = self.L.lcr_nq(cfg.task)
        self.nv = self.L.lcr_nv(cfg.task)
        h = ctypes.c_void_p()
        check(self.L.lcr_create(ctypes.byref(cfg), ctypes.byref(h)))
        self.handle = h
        fam = self.L.lcr_step_kernel_family(self.handle)
        self.step_kernel_family = {0: "one wave per 64 envs (lcr_step_kernel)", 1: "two cooperating waves per 64 envs (lcr_step2_kernel, one wave per SIMD)",
                                   2: "two cooperating waves per 64 envs (lcr_step2_kernel, two waves per SIMD)"}[fam]
        self.step_kernel_name = "lcr_step_kernel" if fam == 0 else "lcr_step2_kernel"
        ov, out = LcrObsView(), LcrOutView()
        check(self.L.lcr_get_obs(self.handle, ctypes.byref(ov)))
        check(self.L.lcr_get_outputs(self.handle, ctypes.byref(out)))
        N = self.n
        self.has_aux = bool(ov.has_aux)
        self.aux_name = {"push": "target_pos", "pick_place": "target_pos", "stack": "cube_blue_pos"}.get(self.task_name)
        self.cube_name = "cube_red_pos" if self.task_name == "stack" else "cube_pos"
        self.arm_qpos = DeviceArray(self, ov.arm_qpos, (6, N), np.float32)
        self.arm_qvel = DeviceArray(self, ov.arm_qvel, (6, N), np.float32)
        self.cube_pos = DeviceArray(self, ov.cube_pos, (3, N), np.float32)
        self.aux_pos = DeviceArray(self, ov.aux_pos, (3, N), np.float32) if ov.has_aux else None
        img = (N, _capi.IMG_H, _capi.IMG_W, 3)
        self.image_front = DeviceArray(self, ov.image_front, img, np.uint8) if ov.image_front else None
        self.image_top = DeviceArray(self, ov.image_top, img, np.uint8) if ov.image_top else None
        self.reward = DeviceArray(self, out.reward, (N,), np.float32)
        self.terminated = DeviceArray(self, out.terminated, (N,), np.uint8)
        self.truncated = DeviceArray(self, out.truncated, (N,), np.uint8)
        self.is_success = DeviceArray(self, out.is_success, (N,), np.uint8)
        self.did_reset = DeviceArray(self, out.did_reset, (N,), np.uint8)
        self.terminal_obs = DeviceArray(self, out.terminal_obs, (18, N), np.float32)
        self.terminal_quat = DeviceArray(self, out.terminal_quat, (8, N), np.float32)
        self.timestamp = DeviceArray(self, out.timestamp, (N,), np.float64)
        self.current_goal = DeviceArray(self, out.current_goal, (N,), np.int32)
        self.active_mask = DeviceArray(self, out.active_mask, (N,), np.uint32) if out.active_mask else None
        self.active_count = DeviceArray(self, out.active_count, (N,), np.uint32) if out.active_count else None
        self.max_sweeps = DeviceArray(self, out.max_sweeps, (N,), np.uint32) if out.max_sweeps else None
        self.choice = DeviceArray(self, out.choice, (N,), np.uint32) if out.choice else None
        self.ctrl = DeviceArray(self, out.ctrl, (6, N), np.float32) if out.ctrl else None

    # ---- lifecycle ----
    def close(self):
        if getattr(self, "handle", None):
            if getattr(self, "_calib_dst", None) is not None:
                self.L.lcr_free(self.handle, self._calib_dst)
                self._calib_dst = None
            self.L.lcr_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        check(self.L.lcr_set_stream(self.handle, ctypes.c_void_p(int(stream_ptr) if stream_ptr else 0)))

    def sync(self):
        check(self.L.lcr_sync(self.handle))

    # ---- reset / step ----
    def reset(self, seeds=None, mask=None):
        s = None if seeds is None else np.ascontiguousarray(seeds, np.uint64)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        if s is not None and s.shape != (self.n,):
            raise ValueError("seeds must have shape (n_envs,)")
        if m is not None and m.shape != (self.n,):
            raise ValueError("mask must have shape (n_envs,)")
        check(self.L.lcr_reset(self.handle, _vp(m), _vp(s)))

    def step_device(self, action_ptr):
        """action_ptr: device pointer to float32 [k][N] (e.g. tensor.data_ptr())."""
        check(self.L.lcr_step(self.handle, ctypes.c_void_p(int(action_ptr))))

    def step(self, action):
        """action: host array, shape (N, k) (env-major, as a VecEnv passes it) -> transposed to [k][N]."""
        a = np.asarray(action, np.float32)
        if a.shape != (self.n, self.action_dim):
            raise ValueError("Action dimension mismatch")  # reach_cube_env.py:231-232
        at = np.ascontiguousarray(a.T)
        check(self.L.lcr_step_host(self.handle, _vp(at)))

    # ---- device helpers ----
    def alloc_actions(self):
        p = ctypes.c_void_p()
        check(self.L.lcr_malloc(self.handle, 4 * self.action_dim * self.n, ctypes.byref(p)))
        return DeviceArray(self, p.value, (self.action_dim, self.n), np.float32, readonly=False)

    def free(self, arr):
        check(self.L.lcr_free(self.handle, ctypes.c_void_p(arr.ptr)))

    def fill_random_actions(self, arr, seed, step):
        check(self.L.lcr_fill_random_actions(self.handle, ctypes.c_void_p(arr.ptr), int(seed), int(step)))

    def render(self, env=0, camera="camera_vizu", width=640, height=640):
        """ray-cast one env: camera_front / camera_top / camera_vizu -> (height, width, 3) uint8"""
        cam = {"camera_front": 0, "camera_top": 1, "camera_vizu": 2}[camera]
        out = np.empty((height, width, 3), np.uint8)
        check(self.L.lcr_render(self.handle, int(env), cam, int(width), int(height), _vp(out)))
        return out

    def render_state(self, qpos, target=None, camera="camera_front", width=320, height=240):
        """ray-cast an arbitrary pose (qpos of length nq as env.data.qpos; target_pos or None) without touching the sim state"""
        cam = {"camera_front": 0, "camera_top": 1, "camera_vizu": 2}[camera]
        q = np.ascontiguousarray(qpos, np.float64)
        if q.shape != (self.nq,):
            raise ValueError(f"qpos must have shape ({self.nq},)")
        t = None if target is None else np.ascontiguousarray(target, np.float32)
        out = np.empty((height, width, 3), np.uint8)
        check(self.L.lcr_render_state(self.handle, cam, int(width), int(height), _vp(q), _vp(t), _vp(out)))
        return out

    def render_terminal(self, env_ids):
        """last frames (camera_front, camera_top) of the episodes the last step ended in the listed envs, ray-cast as ONE batch from their
        terminal poses (the envs themselves have already been reset): two (len(env_ids), 240, 320, 3) uint8 arrays"""
        ids = np.ascontiguousarray(env_ids, np.int32)
        if ids.size and self.image_front is not None and ids.min() >= 0 and ids.max() < self.n and not self.did_reset.numpy()[ids].all():   # (lcr.h: the terminal poses belong to envs the LAST step reset; anything else is a stale frame)
            raise ValueError("render_terminal: every listed env must have finished an episode in the last step (did_reset)")
        front = np.empty((ids.size, _capi.IMG_H, _capi.IMG_W, 3), np.uint8)
        top = np.empty_like(front)
        check(self.L.lcr_render_terminal(self.handle, _vp(ids), int(ids.size), _vp(front), _vp(top)))
        return front, top

    def read_rows(self, arr, rows):
        """host copies of selected leading-axis rows of a device array (e.g. the frames of a few recorded envs): one small
        device-to-host copy per row instead of the whole array"""
        rows = list(rows)
        row_shape = arr.shape[1:]
        nb = int(np.prod(row_shape)) * arr.dtype.itemsize
        out = np.empty((len(rows),) + row_shape, arr.dtype)
        for i, r in enumerate(rows):
            check(self.L.lcr_memcpy_d2h(self.handle, _vp(out[i]), ctypes.c_void_p(arr.ptr + int(r) * nb), nb))
        return out

    def calibrate_copy(self, n_floats):
        """launch the known-byte-count copy kernel once (profiling calibration); returns bytes read == bytes written"""
        if getattr(self, "_calib_dst", None) is None or self._calib_n < n_floats:
            if getattr(self, "_calib_dst", None) is not None:
                check(self.L.lcr_free(self.handle, self._calib_dst))
                self._calib_dst = None
            p = ctypes.c_void_p()
            check(self.L.lcr_malloc(self.handle, 4 * n_floats, ctypes.byref(p)))
            self._calib_dst, self._calib_n = p, n_floats
        check(self.L.lcr_calibrate_copy(self.handle, self._calib_dst, n_floats))
        return 4 * n_floats

    def timer_begin(self):
        check(self.L.lcr_timer_begin(self.handle))

    def timer_end(self):
        ms = ctypes.c_float()
        check(self.L.lcr_timer_end(self.handle, ctypes.byref(ms)))
        return ms.value

    # ---- host-side views ----
    def observations(self):
        """dict of (N, .) float32 numpy arrays with the reference's keys (get_observation reach:281-295)."""
        obs = {"arm_qpos": self.arm_qpos.numpy().T.copy(), "arm_qvel": self.arm_qvel.numpy().T.copy()}
        # (insertion order of the reference's get_observation: target_pos, images, cube position(s) -- push_cube_env.py:293-306)
        if self.task_name in ("push", "pick_place"):
            obs["target_pos"] = self.aux_pos.numpy().T.copy()  # always present (push_cube_env.py:297)
        if self.image_front is not None:
            obs["image_front"] = self.image_front.numpy()
            obs["image_top"] = self.image_top.numpy()
        if OBS_MODES["state"] == self.cfg.obs_mode or OBS_MODES["both"] == self.cfg.obs_mode:
            obs[self.cube_name] = self.cube_pos.numpy().T.copy()
            if self.task_name == "stack":
                obs["cube_blue_pos"] = self.aux_pos.numpy().T.copy()
        return obs

    def fetch_host(self):
        """State observations, rewards and flags of all envs after ONE device-to-host copy into pinned memory (lcr_fetch_host).
        Returns a dict of numpy VIEWS (no further copy) valid until the next call: observations as (N, .) float32 (strided),
        reward (N,) float32, terminated / truncated / is_success / did_reset (N,) bool, terminal_obs (N, 18) or None."""
        hv = LcrHostView()
        check(self.L.lcr_fetch_host(self.handle, ctypes.byref(hv)))
        N = self.n

        def view(ptr, rows, dt):
            if not ptr:
                return None
            cnt = rows * N
            buf = (ctypes.c_char * (cnt * np.dtype(dt).itemsize)).from_address(ptr)
            a = np.frombuffer(buf, dt, cnt)
            return a.reshape(rows, N).T if rows > 1 else a

        out = {
            "arm_qpos": view(hv.arm_qpos, 6, np.float32), "arm_qvel": view(hv.arm_qvel, 6, np.float32),
            "cube_pos": view(hv.cube_pos, 3, np.float32), "aux_pos": view(hv.aux_pos, 3, np.float32),
            "reward": view(hv.reward, 1, np.float32),
            "terminated": view(hv.terminated, 1, np.bool_), "truncated": view(hv.truncated, 1, np.bool_),
            "is_success": view(hv.is_success, 1, np.bool_), "did_reset": view(hv.did_reset, 1, np.bool_),
            "terminal_obs": view(hv.terminal_obs, 18, np.float32) if hv.any_reset else None,
        }
        return out

    def outputs(self):
        return {
            "reward": self.reward.numpy(),
            "terminated": self.terminated.numpy().astype(bool),
            "truncated": self.truncated.numpy().astype(bool),
            "is_success": self.is_success.numpy().astype(bool),
            "did_reset": self.did_reset.numpy().astype(bool),
        }

    def get_state(self):
        N = self.n
        st = {
            "qpos": np.zeros((self.nq, N)), "qvel": np.zeros((self.nv, N)), "ee_lag": np.zeros((3, N)),
            "target": np.zeros((3, N), np.float32), "elapsed": np.zeros(N, np.int32), "rng": np.zeros((4, N), np.uint64),
            "current_goal": np.zeros(N, np.int32), "sim_time": np.zeros(N),
            "warm": np.zeros((_capi.NWARM, N), np.float32),   # carried constraint forces (mjData.qacc_warmstart of the reference's sim)
        }
        check(self.L.lcr_get_state(self.handle, _vp(st["qpos"]), _vp(st["qvel"]), _vp(st["ee_lag"]), _vp(st["target"]),
                                   _vp(st["elapsed"]), _vp(st["rng"]), _vp(st["current_goal"]), _vp(st["sim_time"]), _vp(st["warm"])))
        return st

    def set_state(self, qpos=None, qvel=None, ee_lag=None, target=None, elapsed=None, rng=None, current_goal=None, sim_time=None, warm=None):
        """Overwrite (parts of) the simulator state.  `set_state(**get_state())` is an exact checkpoint restore.  Setting qpos or qvel
        WITHOUT `warm` drops the carried constraint forces: the next step then starts from a cold solve."""
        N = self.n

        def prep(a, shape, dt):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dt)
            if a.shape != shape:
                raise ValueError(f"bad state shape {a.shape}, want {shape}")
            return a

        qpos, qvel = prep(qpos, (self.nq, N), np.float64), prep(qvel, (self.nv, N), np.float64)
        ee_lag, target = prep(ee_lag, (3, N), np.float64), prep(target, (3, N), np.float32)
        elapsed, rng = prep(elapsed, (N,), np.int32), prep(rng, (4, N), np.uint64)
        current_goal, sim_time = prep(current_goal, (N,), np.int32), prep(sim_time, (N,), np.float64)
        warm = prep(warm, (_capi.NWARM, N), np.float32)
        check(self.L.lcr_set_state(self.handle, _vp(qpos), _vp(qvel), _vp(ee_lag), _vp(target), _vp(elapsed), _vp(rng),
                                   _vp(current_goal), _vp(sim_time), _vp(warm)))
