"""ctypes binding of the CPU oracle (oracle/lcr_oracle.c).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never by the
product package (gym_lowcostrobot_amd), which must fail loudly without its HIP library.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TASKS = {"reach": 0, "lift": 1, "push": 2, "pick_place": 3, "stack": 4, "push_loop": 5}
NQ_MAX, NV_MAX = 20, 18


# orc_params.solver.  The oracle knows one solver more than the product's ABI (include/lcr.h: lcr_solver {LCR_SOLVER_PGS = 0, LCR_SOLVER_NEWTON = 1}): the yard-stick
# "Newton to machine precision" of tools/kkt_distance.py sits at 1 here, the product's Newton (same budget, tolerances and line search as the kernels) at 2.
ORC_SOLVER_PGS, ORC_SOLVER_EXACT, ORC_SOLVER_NEWTON = 0, 1, 2
LCR_TO_ORC_SOLVER = {0: ORC_SOLVER_PGS, 1: ORC_SOLVER_NEWTON}   # lcr_solver -> orc_params.solver


class OrcParams(ctypes.Structure):
    _fields_ = [
        ("task", ctypes.c_int32),
        ("action_mode", ctypes.c_int32),
        ("reward_type", ctypes.c_int32),
        ("block_gripper", ctypes.c_int32),
        ("distance_threshold", ctypes.c_double),
        ("cube_xy_range", ctypes.c_double),
        ("target_xy_range", ctypes.c_double),
        ("goal_z_range", ctypes.c_double),
        ("height_threshold", ctypes.c_double),
        ("n_substeps", ctypes.c_int32),
        ("max_episode_steps", ctypes.c_int32),
        ("impratio", ctypes.c_double),
        ("pgs_iters", ctypes.c_int32),
        ("compat", ctypes.c_uint32),
        ("auto_reset", ctypes.c_int32),
        ("warm_start", ctypes.c_int32),
        ("arm_collision", ctypes.c_int32),
        ("pgs_tol", ctypes.c_double),
        ("condim6", ctypes.c_int32),
        ("proxy_groups", ctypes.c_int32),
        ("cc_points", ctypes.c_int32),
        ("cone", ctypes.c_int32),       # 3 (default, five tasks): block projected-gradient step; 0 (default of push_loop): rows + radial projection (D2); 1: MuJoCo's PGS block update (QCQP)
        ("pgs_cap", ctypes.c_int32),    # most sweeps of the converged mode (0 = 50)
        ("solver", ctypes.c_int32),     # 0: PGS (= the kernels); 1: primal Newton to machine precision (exact optimum of MuJoCo's convex problem)
        ("jacobi", ctypes.c_int32),     # 1 (default; push_loop: 0): two row groups (arm-only | cube rows) sweep concurrently (= the kernels' two waves); 0: one Gauss-Seidel pass
        ("newton_iters", ctypes.c_int32),  # solver = 2 (preset "faithful"): most Newton iterations per substep
        ("ls_iters", ctypes.c_int32),      # ... most evaluations of phi' per line search
        ("newton_tol", ctypes.c_double),
        ("ls_tol", ctypes.c_double),
        ("finger_geom", ctypes.c_int32),
    ]


class OrcIO(ctypes.Structure):
    _fields_ = [
        ("qpos", ctypes.c_void_p),
        ("qvel", ctypes.c_void_p),
        ("ee_lag", ctypes.c_void_p),
        ("target", ctypes.c_void_p),
        ("elapsed", ctypes.c_void_p),
        ("rng", ctypes.c_void_p),
        ("obs", ctypes.c_void_p),
        ("term_obs", ctypes.c_void_p),
        ("reward", ctypes.c_void_p),
        ("reward64", ctypes.c_void_p),
        ("terminated", ctypes.c_void_p),
        ("truncated", ctypes.c_void_p),
        ("is_success", ctypes.c_void_p),
        ("did_reset", ctypes.c_void_p),
        ("goal", ctypes.c_void_p),
        ("sim_time", ctypes.c_void_p),
        ("active_mask", ctypes.c_void_p),
        ("active_count", ctypes.c_void_p),
        ("max_sweeps", ctypes.c_void_p),
        ("choice", ctypes.c_void_p),
        ("warm", ctypes.c_void_p),
        ("kkt", ctypes.c_void_p),
    ]


def build(force=False):
    """Compile the oracle with gcc (oracle/Makefile)."""
    so = os.path.join(_HERE, "liblcr_oracle.so")
    src = os.path.join(_HERE, "lcr_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


_libs = {}


def lib(f32=False):
    """f32: False -> fp64 oracle, True -> fp32-arithmetic twin, "count" -> flop-census build (instrumented scalar)"""
    key = "count" if f32 == "count" else bool(f32)
    if key not in _libs:
        build()
        name = {False: "liblcr_oracle.so", True: "liblcr_oracle_f32.so", "count": "liblcr_oracle_count.so"}[key]
        L = ctypes.CDLL(os.path.join(_HERE, name))
        L.orc_rng_double.restype = ctypes.c_double
        L.orc_nq.restype = ctypes.c_int
        L.orc_nv.restype = ctypes.c_int
        L.orc_action_dim.restype = ctypes.c_int
        L.orc_max_threads.restype = ctypes.c_int
        L.orc_ik.restype = ctypes.c_int
        L.orc_warm_bytes.restype = ctypes.c_int
        _libs[key] = L
    return _libs[key]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class Oracle:
    """n independent envs stepped on the CPU, AoS arrays (env-major)."""

    def __init__(self, task, n, f32=False, kkt=False, preset=None, **kw):
        self.L = lib(f32)
        self.task = TASKS[task] if isinstance(task, str) else int(task)
        self.n = n
        self.params = OrcParams()
        if preset is None:
            preset = os.environ.get("LCR_PRESET") or None   # (the switch the product's VecSim reads too: one variable flips both sides of a parity test)
        if preset is None:
            self.L.orc_default_params(ctypes.byref(self.params), self.task)
        else:   # "faithful": six-row finger contacts everywhere, eight-point box-box, Newton on the primal; "fast": rounds 1-4
            self.L.orc_preset_params(ctypes.byref(self.params), self.task, {"faithful": 0, "fast": 1}[preset])
        for k, v in kw.items():
            if not hasattr(self.params, k):
                raise AttributeError(k)
            setattr(self.params, k, v)
        if self.params.solver == 2 and any(k in kw for k in ("pgs_iters", "cone", "jacobi", "pgs_tol", "pgs_cap")):
            raise ValueError("sweep parameters (pgs_iters, cone, jacobi, pgs_tol, pgs_cap) belong to preset='fast' (solver = 0); the default preset runs Newton's method")
        self.nq = self.L.orc_nq(self.task)
        self.nv = self.L.orc_nv(self.task)
        self.action_dim = self.L.orc_action_dim(ctypes.byref(self.params))
        self.qpos = np.zeros((n, NQ_MAX))
        self.qpos[:, 9] = 1.0
        self.qpos[:, 16] = 1.0
        self.qvel = np.zeros((n, NV_MAX))
        self.ee_lag = np.zeros((n, 3))
        self.target = np.zeros((n, 3), np.float32)
        self.elapsed = np.zeros(n, np.int32)
        self.rng = np.zeros((n, 4), np.uint64)
        self.obs = np.zeros((n, 18), np.float32)
        self.term_obs = np.zeros((n, 18), np.float32)
        self.reward = np.zeros(n, np.float32)
        self.reward64 = np.zeros(n)
        self.terminated = np.zeros(n, np.uint8)
        self.truncated = np.zeros(n, np.uint8)
        self.is_success = np.zeros(n, np.uint8)
        self.did_reset = np.zeros(n, np.uint8)
        self.goal = np.zeros(n, np.int32)
        self.sim_time = np.zeros(n)
        self.active_mask = np.zeros(n, np.uint32)
        self.active_count = np.zeros(n, np.uint32)
        self.max_sweeps = np.zeros(n, np.uint32)
        self.choice = np.zeros(n, np.uint32)
        # constraint forces carried between control steps (opaque records; zero them when a state is set from outside)
        self.warm = np.zeros((n, (self.L.orc_warm_bytes() + 7) // 8 * 8), np.uint8)
        # solver-independent optimality certificate of the contact solve (orc_io.kkt): computed only on request (O(rows^2) per substep)
        self.kkt = np.zeros(n) if kkt else None
        self.io = OrcIO(
            _p(self.qpos), _p(self.qvel), _p(self.ee_lag), _p(self.target), _p(self.elapsed), _p(self.rng),
            _p(self.obs), _p(self.term_obs), _p(self.reward), _p(self.reward64), _p(self.terminated),
            _p(self.truncated), _p(self.is_success), _p(self.did_reset), _p(self.goal), _p(self.sim_time),
            _p(self.active_mask), _p(self.active_count), _p(self.max_sweeps), _p(self.choice), _p(self.warm),
            _p(self.kkt) if kkt else None,
        )

    def reset(self, seeds=None, mask=None):
        s = None if seeds is None else np.ascontiguousarray(seeds, np.uint64)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.L.orc_reset(ctypes.byref(self.params), ctypes.byref(self.io), self.n,
                         None if m is None else _p(m), None if s is None else _p(s))

    def step(self, action, threads=1):
        a = np.ascontiguousarray(action, np.float32)
        assert a.shape == (self.n, self.action_dim), (a.shape, (self.n, self.action_dim))
        self.L.orc_step(ctypes.byref(self.params), ctypes.byref(self.io), self.n, _p(a), int(threads))

    def diag(self):
        r, c, res = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        self.L.orc_last_diag(ctypes.byref(r), ctypes.byref(c), ctypes.byref(res))
        return r.value, c.value, res.value


# ---- model queries ----
def fk(q):
    q = np.ascontiguousarray(q, np.float64)
    lp, site, sph = np.zeros((6, 3)), np.zeros(3), np.zeros((2, 3))
    lib().orc_fk(_p(q), _p(lp), _p(site), _p(sph))
    return lp, site, sph


def link_frames(q):
    """world rotation matrices [6][3][3] (columns = the link's axes) and origins [6][3] of link_1 .. link_6"""
    q = np.ascontiguousarray(q, np.float64)
    R, p = np.zeros((6, 3, 3)), np.zeros((6, 3))
    lib().orc_link_frames(_p(q), _p(R), _p(p))
    return R, p


def proxies(q):
    """world centres and radii of the arm-link proxy spheres (D3)"""
    q = np.ascontiguousarray(q, np.float64)
    c, r = np.zeros((8, 3)), np.zeros(8)
    n = lib().orc_proxies(_p(q), _p(c), _p(r))
    return c[:n], r[:n]


def mass_matrix(q, armature=True):
    q = np.ascontiguousarray(q, np.float64)
    M = np.zeros((6, 6))
    lib().orc_mass_matrix(_p(q), int(armature), _p(M))
    return M


def bias(q, qd):
    q = np.ascontiguousarray(q, np.float64)
    qd = np.ascontiguousarray(qd, np.float64)
    b = np.zeros(6)
    lib().orc_bias(_p(q), _p(qd), _p(b))
    return b


def site_jac(q):
    q = np.ascontiguousarray(q, np.float64)
    J = np.zeros((3, 6))
    lib().orc_site_jac(_p(q), _p(J))
    return J


def invweight0():
    t, r, d = np.zeros(6), np.zeros(6), np.zeros(6)
    lib().orc_invweight0(_p(t), _p(r), _p(d))
    return t, r, d


def ik(q, target):
    q = np.ascontiguousarray(q, np.float64)
    t = np.ascontiguousarray(target, np.float64)
    qc, qs, sl = np.zeros(6), np.zeros(6), np.zeros(3)
    it = lib().orc_ik(_p(q), _p(t), _p(qc), _p(qs), _p(sl))
    return it, qc, qs, sl


def model_table():
    """the oracle's own L0 constants as a dict (see orc_model_table in lcr_oracle.c)"""
    buf = np.zeros(256)
    n = lib().orc_model_table(_p(buf))
    it = iter(buf[:n])
    take = lambda k: np.array([next(it) for _ in range(k)])
    links = []
    for _ in range(6):
        links.append({"pos": take(3), "axis": take(3), "ipos": take(3), "iquat": take(4), "mass": take(1)[0], "diaginertia": take(3), "range": take(2)})
    t = {"links": links, "site": take(3)}
    for k in ("armature", "damping", "kp", "kv", "frcrange", "timestep", "cube_half"):
        t[k] = take(1)[0]
    t["tasks"] = [dict(zip(("cube_mass", "cube_inertia", "mu_tan", "mu_tors"), take(4))) for _ in range(6)]
    t["walls"] = dict(zip(("x", "y0", "y1", "top", "thick"), take(5)))
    t["finger"] = dict(zip(("solimp_d0", "solimp_dmax", "solimp_width", "mu_tan", "mu_tors", "mu_roll"), take(6)))
    t["spheres"] = [{"link": int(take(1)[0]), "pos": take(3), "rad": take(1)[0]} for _ in range(2)]
    rest = list(it)
    t["proxies"] = [{"link": int(rest[6 * i]), "pos": np.array(rest[6 * i + 1:6 * i + 4]), "rad": rest[6 * i + 4], "cube": int(rest[6 * i + 5])}
                    for i in range(len(rest) // 6)]
    return t


def rng_seed(seed):
    r = np.zeros(4, np.uint64)
    lib().orc_rng_seed(ctypes.c_uint64(int(seed)), _p(r))
    return r


def rng_double(r):
    return lib().orc_rng_double(_p(r))


def loop_reward(cube_xyz_f32, goal):
    c = np.ascontiguousarray(cube_xyz_f32, np.float32)
    g = ctypes.c_int32(int(goal))
    ov, r, s = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint8()
    lib().orc_loop_reward(_p(c), ctypes.byref(g), ctypes.byref(ov), ctypes.byref(r), ctypes.byref(s))
    return ov.value, r.value, int(s.value), g.value
