"""ctypes binding of the C ABI declared in include/lcr.h (liblcr_hip.so).

There is deliberately NO fallback: if the HIP library is missing or no MI355X is visible, importing the
library / creating a simulator raises -- the product path never routes through a CPU implementation.
"""
import ctypes
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LCR_LIB_PATH") or os.path.join(_HERE, "liblcr_hip.so")  # override: A/B builds of the same ABI

ABI_VERSION = 5
NWARM = 124   # LCR_NWARM: floats per env of carried constraint forces (layout: include/lcr.h)
TASKS = {"reach": 0, "lift": 1, "push": 2, "pick_place": 3, "stack": 4, "push_loop": 5}
ACTION_MODES = {"joint": 0, "ee": 1}
OBS_MODES = {"image": 0, "state": 1, "both": 2}
REWARD_TYPES = {"sparse": 0, "dense": 1}
STEP_KERNELS = {"auto": 0, "single": 1, "coop": 2}   # lcr_config.step_kernel
SOLVERS = {"pgs": 0, "newton": 1}                     # lcr_config.solver
PRESETS = {"faithful": 0, "fast": 1}                  # lcr_config_preset
PROFILE_MODES = {None: None, "wave_cycles": 2, "phase_cycles": 3}   # lcr_config.diagnostics values 2, 3 (per-wave cycle read-back; see include/lcr.h)
COMPAT_ZERO_QVEL_ON_RESET = 1
COMPAT_COLD_SOLVE_EACH_STEP = 2   # contact solver starts every control step from zero forces (default: forces carried across steps)
IMG_H, IMG_W = 240, 320

LCR_OK, LCR_ERR_INVALID, LCR_ERR_NO_DEVICE, LCR_ERR_HIP, LCR_ERR_OOM, LCR_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5

# every symbol include/lcr.h declares (tests check the .so exports exactly these)
SYMBOLS = [
    "lcr_abi_version", "lcr_last_error", "lcr_config_default", "lcr_config_preset", "lcr_action_dim", "lcr_nq", "lcr_nv",
    "lcr_create", "lcr_destroy", "lcr_set_stream", "lcr_sync", "lcr_reset", "lcr_step", "lcr_step_host",
    "lcr_get_obs", "lcr_get_outputs", "lcr_fetch_host", "lcr_get_state", "lcr_set_state", "lcr_malloc", "lcr_free",
    "lcr_memcpy_h2d", "lcr_memcpy_d2h", "lcr_timer_begin", "lcr_timer_end", "lcr_fill_random_actions",
    "lcr_calibrate_copy", "lcr_render", "lcr_render_state", "lcr_render_terminal", "lcr_step_kernel_family",
]


class LcrConfig(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("task", ctypes.c_int32),
        ("n_envs", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("env_id_offset", ctypes.c_int64),
        ("action_mode", ctypes.c_int32),
        ("obs_mode", ctypes.c_int32),
        ("reward_type", ctypes.c_int32),
        ("block_gripper", ctypes.c_int32),
        ("distance_threshold", ctypes.c_double),
        ("cube_xy_range", ctypes.c_double),
        ("target_xy_range", ctypes.c_double),
        ("goal_z_range", ctypes.c_double),
        ("height_threshold", ctypes.c_double),
        ("impratio", ctypes.c_double),
        ("n_substeps", ctypes.c_int32),
        ("max_episode_steps", ctypes.c_int32),
        ("pgs_iters", ctypes.c_int32),
        ("compat", ctypes.c_uint32),
        ("auto_reset", ctypes.c_int32),
        ("arm_collision", ctypes.c_int32),
        ("base_seed", ctypes.c_uint64),
        ("pgs_tol", ctypes.c_double),
        ("diagnostics", ctypes.c_int32),
        ("finger_cube_condim", ctypes.c_int32),
        ("step_kernel", ctypes.c_int32),
        ("cc_points", ctypes.c_int32),
        ("global_envs", ctypes.c_int64),   # ABI v4: envs of the whole job (0 = n_envs); the step_kernel = 0 dispatch looks at it, never at the shard size
        ("solver", ctypes.c_int32),        # ABI v5: SOLVERS
        ("newton_iters", ctypes.c_int32),
        ("ls_iters", ctypes.c_int32),
        ("finger_floor_condim", ctypes.c_int32),
        ("newton_tol", ctypes.c_double),
        ("ls_tol", ctypes.c_double),
    ]


class LcrObsView(ctypes.Structure):
    _fields_ = [
        ("n_envs", ctypes.c_int32),
        ("has_aux", ctypes.c_int32),
        ("arm_qpos", ctypes.c_void_p),
        ("arm_qvel", ctypes.c_void_p),
        ("cube_pos", ctypes.c_void_p),
        ("aux_pos", ctypes.c_void_p),
        ("image_front", ctypes.c_void_p),
        ("image_top", ctypes.c_void_p),
    ]


class LcrOutView(ctypes.Structure):
    _fields_ = [
        ("n_envs", ctypes.c_int32),
        ("_pad", ctypes.c_int32),
        ("reward", ctypes.c_void_p),
        ("terminated", ctypes.c_void_p),
        ("truncated", ctypes.c_void_p),
        ("is_success", ctypes.c_void_p),
        ("did_reset", ctypes.c_void_p),
        ("terminal_obs", ctypes.c_void_p),
        ("terminal_quat", ctypes.c_void_p),
        ("timestamp", ctypes.c_void_p),
        ("current_goal", ctypes.c_void_p),
        ("active_mask", ctypes.c_void_p),
        ("active_count", ctypes.c_void_p),
        ("max_sweeps", ctypes.c_void_p),
        ("choice", ctypes.c_void_p),
        ("ctrl", ctypes.c_void_p),
    ]


class LcrHostView(ctypes.Structure):
    _fields_ = [
        ("n_envs", ctypes.c_int32),
        ("any_reset", ctypes.c_int32),
        ("arm_qpos", ctypes.c_void_p),
        ("arm_qvel", ctypes.c_void_p),
        ("cube_pos", ctypes.c_void_p),
        ("aux_pos", ctypes.c_void_p),
        ("reward", ctypes.c_void_p),
        ("terminated", ctypes.c_void_p),
        ("truncated", ctypes.c_void_p),
        ("is_success", ctypes.c_void_p),
        ("did_reset", ctypes.c_void_p),
        ("terminal_obs", ctypes.c_void_p),
    ]


class LcrError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"lcr error {code}: {msg}")
        self.code = code
        self.msg = msg


_lib = None


def load():
    """Load liblcr_hip.so; raises OSError with a build hint if it is absent (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            f"{LIB_PATH} not found: build it with `python -m gym_lowcostrobot_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    # PyTorch-ROCm wheels bundle their own HIP/HSA runtime (torch/lib/libamdhip64.so, same SONAME as the system one).  Two HIP
    # runtimes in one process do not coexist, so whichever of torch and this library comes first must settle on ONE copy.
    # If torch is installed but not imported yet, its bundled runtime is pre-loaded here BY PATH (no `import torch`): the
    # dynamic linker then resolves liblcr_hip.so's DT_NEEDED libamdhip64.so.7 to that already-loaded object, and a later
    # `import torch` finds its own runtime in place -- the import order no longer matters.  LCR_NO_TORCH_PRELOAD=1 skips this
    # (processes that never use torch run on the system runtime of /opt/rocm).
    if "torch" not in sys.modules and os.environ.get("LCR_NO_TORCH_PRELOAD") != "1":
        try:
            spec = importlib.util.find_spec("torch")
            rt = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so") if spec and spec.submodule_search_locations else None
            if rt and os.path.exists(rt):
                ctypes.CDLL(rt, mode=ctypes.RTLD_GLOBAL)
        except Exception:
            pass
    L = ctypes.CDLL(LIB_PATH)
    _check_single_hip_runtime()
    vp, i32, u64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint64
    L.lcr_abi_version.restype = ctypes.c_int
    L.lcr_last_error.restype = ctypes.c_char_p
    L.lcr_config_default.argtypes = [ctypes.POINTER(LcrConfig), ctypes.c_int]
    L.lcr_config_preset.argtypes = [ctypes.POINTER(LcrConfig), ctypes.c_int, ctypes.c_int]
    L.lcr_action_dim.argtypes = [ctypes.POINTER(LcrConfig)]
    L.lcr_nq.argtypes = [ctypes.c_int]
    L.lcr_nv.argtypes = [ctypes.c_int]
    L.lcr_create.argtypes = [ctypes.POINTER(LcrConfig), ctypes.POINTER(vp)]
    L.lcr_destroy.argtypes = [vp]
    L.lcr_destroy.restype = None
    L.lcr_step_kernel_family.argtypes = [vp]
    L.lcr_set_stream.argtypes = [vp, vp]
    L.lcr_sync.argtypes = [vp]
    L.lcr_reset.argtypes = [vp, vp, vp]
    L.lcr_step.argtypes = [vp, vp]
    L.lcr_step_host.argtypes = [vp, vp]
    L.lcr_get_obs.argtypes = [vp, ctypes.POINTER(LcrObsView)]
    L.lcr_get_outputs.argtypes = [vp, ctypes.POINTER(LcrOutView)]
    L.lcr_fetch_host.argtypes = [vp, ctypes.POINTER(LcrHostView)]
    L.lcr_get_state.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.lcr_set_state.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.lcr_malloc.argtypes = [vp, ctypes.c_size_t, ctypes.POINTER(vp)]
    L.lcr_free.argtypes = [vp, vp]
    L.lcr_memcpy_h2d.argtypes = [vp, vp, vp, ctypes.c_size_t]
    L.lcr_memcpy_d2h.argtypes = [vp, vp, vp, ctypes.c_size_t]
    L.lcr_timer_begin.argtypes = [vp]
    L.lcr_timer_end.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.lcr_fill_random_actions.argtypes = [vp, vp, u64, u64]
    L.lcr_calibrate_copy.argtypes = [vp, vp, ctypes.c_size_t]
    L.lcr_render.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp]
    L.lcr_render_state.argtypes = [vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, vp, vp]
    L.lcr_render_terminal.argtypes = [vp, vp, ctypes.c_int, vp, vp]
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("lcr_last_error", "lcr_destroy"):
            fn.restype = ctypes.c_int
    if L.lcr_abi_version() != ABI_VERSION:
        raise OSError(f"liblcr_hip.so ABI {L.lcr_abi_version()} != binding ABI {ABI_VERSION}: rebuild")
    _lib = L
    return L


def _check_single_hip_runtime():
    """Two HIP runtimes in one process do not coexist (the second one sees no GPU).  The by-path preload above only unifies them when
    torch's bundled libamdhip64 has the SONAME liblcr_hip.so was linked against (libamdhip64.so.7 for ROCm 7.x wheels); with a torch
    wheel built for another ROCm major both would be mapped: say so loudly instead of failing later with 'no HIP device'."""
    try:
        with open("/proc/self/maps") as f:
            libs = {line.split()[-1] for line in f if "libamdhip64" in line}
    except OSError:
        return
    real = {os.path.realpath(p) for p in libs}
    if len(real) > 1:
        import warnings

        warnings.warn("two HIP runtimes are mapped in this process (" + ", ".join(sorted(real)) + "): liblcr_hip.so was built against the "
                      "ROCm 7 runtime (libamdhip64.so.7); use a PyTorch-ROCm wheel of the same ROCm major, or set LCR_NO_TORCH_PRELOAD=1 and "
                      "import torch in a different process", RuntimeWarning, stacklevel=3)


def check(rc):
    if rc < 0:
        L = load()
        msg = L.lcr_last_error().decode("utf-8", "replace")
        if rc == LCR_ERR_INVALID:
            raise ValueError(msg)
        raise LcrError(rc, msg)
    return rc
