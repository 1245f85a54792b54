// lcr_capi.hip -- the C ABI of include/lcr.h: device-memory ownership, launches, host<->device state I/O.
// No simulation arithmetic lives here (that is lcr_kernels.hip) and there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/lcr.h"
#include "lcr_device.h"
#include "lcr_model_gen.h"

static_assert(LCR_NWARM == LCR_DEV_NWARM, "include/lcr.h and lcr_device.h disagree on the carried-force block");

namespace {
thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                           \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess) return fail(LCR_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e)); \
    } while (0)
}  // namespace

namespace {
void cam_finish(LcrCam &c, const double p[3], double X[3], double Y[3], int height) {
    auto nrm = [](double *v) { double n = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= n; v[1] /= n; v[2] /= n; };
    nrm(X);
    double d = X[0] * Y[0] + X[1] * Y[1] + X[2] * Y[2];
    for (int i = 0; i < 3; i++) Y[i] -= d * X[i];
    nrm(Y);
    double Z[3] = {X[1] * Y[2] - X[2] * Y[1], X[2] * Y[0] - X[0] * Y[2], X[0] * Y[1] - X[1] * Y[0]};
    c.px = (float)p[0]; c.py = (float)p[1]; c.pz = (float)p[2];
    c.xx = (float)X[0]; c.xy = (float)X[1]; c.xz = (float)X[2];
    c.yx = (float)Y[0]; c.yy = (float)Y[1]; c.yz = (float)Y[2];
    c.zx = (float)Z[0]; c.zy = (float)Z[1]; c.zz = (float)Z[2];
    c.s = (float)(2.0 * std::tan(0.5 * 45.0 * M_PI / 180.0) / height);  // MuJoCo default camera fovy = 45 deg
}
// cameras of the scene files (reach_cube.xml:29-31 and siblings)
void make_cameras(int task, LcrCam &front, LcrCam &top, LcrCam &vizu) {
    {   // camera_front pos="0.049 0.5 0.225" xyaxes="-0.998 0.056 -0.000 -0.019 -0.335 0.942"
        double p[3] = {0.049, 0.5, 0.225}, X[3] = {-0.998, 0.056, -0.000}, Y[3] = {-0.019, -0.335, 0.942};
        cam_finish(front, p, X, Y, LCR_IMG_H);
    }
    {   // camera_top pos="0 0.1 0.6" euler="0 0 0"
        double p[3] = {0, 0.1, 0.6}, X[3] = {1, 0, 0}, Y[3] = {0, 1, 0};
        cam_finish(top, p, X, Y, LCR_IMG_H);
    }
    {   // camera_vizu pos="-0.2 0.6 0.3" (reach) / "-0.1 0.6 0.3" (others) quat="-0.15 -0.1 0.6 1"
        double p[3] = {task == LCR_TASK_REACH ? -0.2 : -0.1, 0.6, 0.3};
        double q[4] = {-0.15, -0.1, 0.6, 1.0}, n = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
        double w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
        double X[3] = {1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)};
        double Y[3] = {2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)};
        cam_finish(vizu, p, X, Y, 640);
    }
}
}  // namespace

struct lcr_sim {
    lcr_config cfg;
    LcrDev dev;
    int nq, nv, k;
    int ee_mode;
    hipStream_t stream;
    void *arena;        // one allocation holding every SoA array
    size_t arena_bytes;
    float *action_stage;     // [k][N] staging for lcr_step_host
    unsigned char *mask_dev; // [N]
    unsigned long long *seeds_dev; // [N]
    hipEvent_t ev0, ev1;
    bool has_images;
    LcrCam cam_front, cam_top, cam_vizu;
    unsigned char *render_dev;  // scratch frame for lcr_render
    size_t render_bytes;
    void *term_stage;           // staging of lcr_render_terminal: env ids, gathered poses, two frame blocks
    size_t term_stage_bytes;
    // lcr_fetch_host: pinned host mirror of the arena range [qpos .. did_reset] (+ terminal observations)
    size_t fetch_bytes, tobs_off, tobs_bytes;
    char *host_mirror;
    // image observations: the frames of step k are ray-cast on a second stream from a snapshot of the poses while the step kernel of step k + 1 runs (lcr_step).  The
    // step kernel is VALU-bound and ends with a tail of few slow waves, the frame kernel is HBM-write-bound: together they take little more than the longer one.
    // Every other entry point first makes the caller's stream wait for the pending frames (join_render), so nothing but lcr_step sees the second stream.
    hipStream_t rstream;         // null: frames on the caller's stream, after the step kernel (LCR_RENDER_OVERLAP=0)
    hipEvent_t ev_snap[2], ev_rdone[2];
    float *snap_qpos[2], *snap_target[2];
    bool snap_used[2];
    int rpar, rlast;
    bool rpending;
};

static int join_render(lcr_sim *s) {
    if (s->rpending) {
        hipError_t e = hipStreamWaitEvent(s->stream, s->ev_rdone[s->rlast], 0);
        if (e != hipSuccess) return (int)e;
        s->rpending = false;
    }
    return 0;
}

extern "C" {

int lcr_abi_version(void) { return LCR_ABI_VERSION; }
const char *lcr_last_error(void) { return g_err; }

int lcr_nq(int task) { return task == LCR_TASK_STACK ? 20 : 13; }
int lcr_nv(int task) { return task == LCR_TASK_STACK ? 18 : 12; }

// the reference constructor defaults + the rounds 1-4 solver settings (LCR_PRESET_FAST); lcr_config_preset builds on it
static int config_base(lcr_config *cfg, int task) {
    if (!cfg) return fail(LCR_ERR_INVALID, "cfg is NULL");
    if (task < LCR_TASK_REACH || task > LCR_TASK_PUSH_LOOP) return fail(LCR_ERR_INVALID, "unknown task %d", task);
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof(lcr_config);
    cfg->task = task;
    cfg->n_envs = 1;
    cfg->device = 0;
    cfg->env_id_offset = 0;
    cfg->action_mode = LCR_ACTION_JOINT;   // reach_cube_env.py:80
    cfg->obs_mode = LCR_OBS_IMAGE;         // reach_cube_env.py:79 (the reference default is "image")
    cfg->reward_type = LCR_REWARD_SPARSE;  // reach_cube_env.py:81
    cfg->block_gripper = -1;
    cfg->distance_threshold = 0.05;
    cfg->cube_xy_range = 0.3;
    cfg->target_xy_range = 0.3;
    cfg->goal_z_range = 0.1;
    cfg->height_threshold = 0.1;
    cfg->impratio = 100.0;
    cfg->n_substeps = 20;
    cfg->max_episode_steps = 50;
    cfg->pgs_iters = 4;
    cfg->compat = 0;
    cfg->auto_reset = 1;
    cfg->arm_collision = 1;
    cfg->base_seed = 0;
    cfg->pgs_tol = 1e-6;
    // rolling rows of the finger<->cube contacts: on where their effect on a touched cube within one control step exceeds the fp32
    // parity tolerance in the median (tools/condim6_effect.py: PushCubeLoop 1.4e-2 -- rolling coefficient 1.5 m, push_cube_loop.xml:31;
    // StackTwoCubes 4e-4 -- default coefficient 1e-4 m but cube inertia 1.1e-5), off where it does not (<= 4e-5: deviation D4)
    cfg->finger_cube_condim = (task == LCR_TASK_PUSH_LOOP || task == LCR_TASK_STACK) ? 6 : 4;
    cfg->diagnostics = 0;
    cfg->step_kernel = 0;
    cfg->cc_points = 0;
    cfg->global_envs = 0;   // this handle is the whole job
    cfg->solver = LCR_SOLVER_PGS;
    cfg->newton_iters = 30;
    cfg->ls_iters = 8;
    cfg->finger_floor_condim = 0;
    cfg->newton_tol = 1e-6;
    cfg->ls_tol = 1e-2;   // (MuJoCo's own default: ls_tolerance 0.01; measured on the oracle: a sixth fewer evaluations of phi', same distance to the exact optimum)
    return LCR_OK;
}

int lcr_config_default(lcr_config *cfg, int task) { return lcr_config_preset(cfg, task, LCR_PRESET_FAITHFUL); }

int lcr_config_preset(lcr_config *cfg, int task, int preset) {
    if (preset != LCR_PRESET_FAITHFUL && preset != LCR_PRESET_FAST) return fail(LCR_ERR_INVALID, "unknown preset %d", preset);
    const int rc = config_base(cfg, task);
    if (rc != LCR_OK) return rc;
    if (preset == LCR_PRESET_FAITHFUL) {
        cfg->solver = LCR_SOLVER_NEWTON;
        cfg->finger_cube_condim = 6;
        cfg->finger_floor_condim = 6;
        cfg->cc_points = task == LCR_TASK_STACK ? 8 : 0;
    } else {
        cfg->solver = LCR_SOLVER_PGS;
        cfg->finger_floor_condim = 4;
    }
    return LCR_OK;
}

// the two-wave kernels implement neither the converged solver mode nor the per-wave cycle read-back of diagnostics = 2
static bool two_wave_possible(const lcr_config *cfg) { return cfg->pgs_iters >= 0 && cfg->diagnostics != 2; }

static int resolved_block_gripper(const lcr_config *cfg) {
    if (cfg->block_gripper >= 0) return cfg->block_gripper ? 1 : 0;
    return (cfg->task == LCR_TASK_REACH || cfg->task == LCR_TASK_PUSH || cfg->task == LCR_TASK_PUSH_LOOP) ? 1 : 0;  // reach:82 push:84 loop:80 / lift:82
}

int lcr_action_dim(const lcr_config *cfg) {
    if (!cfg) return fail(LCR_ERR_INVALID, "cfg is NULL");
    if (cfg->action_mode != LCR_ACTION_JOINT && cfg->action_mode != LCR_ACTION_EE)
        return fail(LCR_ERR_INVALID, "Invalid action mode, must be 'ee' or 'joint'");
    return (cfg->action_mode == LCR_ACTION_EE ? 3 : 5) + (resolved_block_gripper(cfg) ? 0 : 1);
}

int lcr_create(const lcr_config *cfg, lcr_sim **out) {
    if (!cfg || !out) return fail(LCR_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(lcr_config))
        return fail(LCR_ERR_INVALID, "lcr_config size mismatch (got %u, want %zu): ABI version skew", cfg->struct_size, sizeof(lcr_config));
    if (cfg->task < LCR_TASK_REACH || cfg->task > LCR_TASK_PUSH_LOOP) return fail(LCR_ERR_INVALID, "unknown task %d", cfg->task);
    if (cfg->n_envs <= 0) return fail(LCR_ERR_INVALID, "n_envs must be positive");
    // the kernels index [component][env] arrays with 32-bit products (20 components at most): keep 20 * n_envs < 2^31
    if (cfg->n_envs > (1 << 26)) return fail(LCR_ERR_INVALID, "n_envs %d exceeds 67108864 per handle; shard the batch over several handles", cfg->n_envs);
    if (cfg->n_substeps <= 0) return fail(LCR_ERR_INVALID, "n_substeps must be positive");
    if (cfg->pgs_iters < 0 && !(cfg->pgs_tol > 0)) return fail(LCR_ERR_INVALID, "pgs_tol must be positive in converged mode (pgs_iters < 0)");
    if (cfg->finger_cube_condim != 0 && cfg->finger_cube_condim != 4 && cfg->finger_cube_condim != 6) return fail(LCR_ERR_INVALID, "finger_cube_condim must be 4 or 6");
    if (cfg->cc_points != 0 && cfg->cc_points != 4 && cfg->cc_points != 8) return fail(LCR_ERR_INVALID, "cc_points must be 4 or 8");
    if (cfg->cc_points == 8 && cfg->task != LCR_TASK_STACK) return fail(LCR_ERR_INVALID, "cc_points = 8 is the cube<->cube manifold of StackTwoCubes; this task has one cube");
    if (cfg->cc_points == 8 && cfg->pgs_iters < 0) return fail(LCR_ERR_UNSUPPORTED, "cc_points = 8 is implemented by the two-wave kernels, the converged solver mode (pgs_iters < 0) by the one-wave kernels");
    if (cfg->step_kernel < 0 || cfg->step_kernel > 2) return fail(LCR_ERR_INVALID, "step_kernel must be 0 (by task and job size), 1 (one wave per 64 envs) or 2 (two cooperating waves)");
    if (cfg->diagnostics < 0 || cfg->diagnostics > 3) return fail(LCR_ERR_INVALID, "diagnostics must be 0, 1 (decision signature), 2 or 3 (per-wave cycle read-back, profiling)");
    // combinations no kernel implements are refused, not silently degraded
    if (cfg->cc_points == 8 && cfg->solver == LCR_SOLVER_PGS && cfg->diagnostics == 2) return fail(LCR_ERR_UNSUPPORTED, "cc_points = 8 runs on the two-wave kernels, diagnostics = 2 (per-wave cycles) on the one-wave kernels");
    if (cfg->cc_points == 8 && cfg->solver == LCR_SOLVER_PGS && cfg->step_kernel == 1) return fail(LCR_ERR_UNSUPPORTED, "cc_points = 8 is implemented by the two-wave kernels only (step_kernel = 1 pins the one-wave family)");
    if (cfg->step_kernel == 2 && cfg->pgs_iters < 0) return fail(LCR_ERR_UNSUPPORTED, "the converged solver mode (pgs_iters < 0) is implemented by the one-wave kernels only (step_kernel = 2 pins the two-wave family)");
    if (cfg->step_kernel == 2 && cfg->task == LCR_TASK_PUSH_LOOP) return fail(LCR_ERR_UNSUPPORTED, "PushCubeLoop has the one-wave step kernel only (lcr_kernels_loop.hip); step_kernel = 2 pins the two-wave family");
    if (cfg->step_kernel == 2 && cfg->diagnostics == 2) return fail(LCR_ERR_UNSUPPORTED, "diagnostics = 2 (per-wave cycles) reads back the one-wave kernels only (step_kernel = 2 pins the two-wave family)");
    if (cfg->solver != LCR_SOLVER_PGS && cfg->solver != LCR_SOLVER_NEWTON) return fail(LCR_ERR_INVALID, "solver must be LCR_SOLVER_PGS (0) or LCR_SOLVER_NEWTON (1)");
    if (cfg->finger_floor_condim != 0 && cfg->finger_floor_condim != 4 && cfg->finger_floor_condim != 6) return fail(LCR_ERR_INVALID, "finger_floor_condim must be 4 or 6");
    if (cfg->solver == LCR_SOLVER_NEWTON) {
        if (cfg->newton_iters <= 0 || cfg->ls_iters <= 0 || !(cfg->newton_tol > 0) || !(cfg->ls_tol > 0)) return fail(LCR_ERR_INVALID, "newton_iters, ls_iters, newton_tol and ls_tol must be positive");
        if (cfg->finger_cube_condim == 4 || cfg->finger_floor_condim == 4) return fail(LCR_ERR_UNSUPPORTED, "the Newton kernels carry six-row finger contacts (finger_cube_condim = finger_floor_condim = 6)");
        if (cfg->step_kernel == 2) return fail(LCR_ERR_UNSUPPORTED, "the Newton kernels are one-wave kernels (step_kernel = 2 pins the two-wave family)");
        if (cfg->pgs_iters < 0) return fail(LCR_ERR_INVALID, "pgs_iters < 0 (converged sweeps) belongs to LCR_SOLVER_PGS");
        if (cfg->diagnostics == 3) return fail(LCR_ERR_UNSUPPORTED, "the per-phase cycle read-back (diagnostics = 3) belongs to the two-wave sweep kernels");
    } else if (cfg->finger_floor_condim == 6) return fail(LCR_ERR_UNSUPPORTED, "six-row finger<->floor contacts are implemented by the Newton kernels (solver = LCR_SOLVER_NEWTON)");
    if (cfg->global_envs < 0) return fail(LCR_ERR_INVALID, "global_envs must be >= 0 (0 = n_envs)");
    if (cfg->global_envs > 0 && (cfg->env_id_offset < 0 || cfg->env_id_offset + (int64_t)cfg->n_envs > cfg->global_envs))
        return fail(LCR_ERR_INVALID, "shard [env_id_offset, env_id_offset + n_envs) = [%lld, %lld) does not lie inside the job of global_envs = %lld",
                    (long long)cfg->env_id_offset, (long long)(cfg->env_id_offset + cfg->n_envs), (long long)cfg->global_envs);
    // A wave (64 consecutive env ids) takes its shortcuts -- slots no lane touches, the coupled lanes it solves cooperatively, the exits of the solver loops -- for all of
    // its lanes at once, so the low-order bits of an env's result depend on which envs share its wave.  "Identical bits for every sharding of a job" (lcr.h) therefore
    // requires that shards are cut at wave boundaries: a shard of a larger job starts at a multiple of 64 and, unless it is the job's last, holds a multiple of 64 envs.
    {
        const bool last = cfg->global_envs <= 0 || cfg->env_id_offset + (int64_t)cfg->n_envs == cfg->global_envs;   // (no job declared: the handle is the job's only or last shard)
        if (cfg->env_id_offset % 64 != 0 || (!last && cfg->n_envs % 64 != 0))
            return fail(LCR_ERR_INVALID, "shard [%lld, %lld) of a job of %lld envs is not cut at wave boundaries: env_id_offset and (except for the last shard) n_envs must be multiples of 64 (lcr.h: global_envs)",
                        (long long)cfg->env_id_offset, (long long)(cfg->env_id_offset + cfg->n_envs), (long long)cfg->global_envs);
    }
    if (cfg->obs_mode < LCR_OBS_IMAGE || cfg->obs_mode > LCR_OBS_BOTH) return fail(LCR_ERR_INVALID, "invalid observation_mode");
    if (cfg->reward_type != LCR_REWARD_SPARSE && cfg->reward_type != LCR_REWARD_DENSE) return fail(LCR_ERR_INVALID, "invalid reward_type");
    int k = lcr_action_dim(cfg);
    if (k < 0) return k;
    const bool gripper_task = !(cfg->task == LCR_TASK_REACH || cfg->task == LCR_TASK_PUSH || cfg->task == LCR_TASK_PUSH_LOOP);
    if (cfg->action_mode == LCR_ACTION_EE && gripper_task && resolved_block_gripper(cfg))
        return fail(LCR_ERR_INVALID, "ee mode with block_gripper on a gripper task indexes action[3] out of range in the reference (lift_cube_env.py:242)");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(LCR_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback", e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(LCR_ERR_INVALID, "device %d out of range (have %d)", cfg->device, ndev);
    HIPCHK(hipSetDevice(cfg->device));
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(LCR_ERR_NO_DEVICE, "device %d is %s; the kernels are built for gfx950 (MI355X) only", cfg->device, prop.gcnArchName);

    lcr_sim *s = new (std::nothrow) lcr_sim();
    if (!s) return fail(LCR_ERR_OOM, "host allocation failed");
    memset(s, 0, sizeof *s);
    s->cfg = *cfg;
    s->cfg.block_gripper = resolved_block_gripper(cfg);
    s->nq = lcr_nq(cfg->task);
    s->nv = lcr_nv(cfg->task);
    s->k = k;
    s->ee_mode = cfg->action_mode == LCR_ACTION_EE;
    s->stream = nullptr;
    s->has_images = cfg->obs_mode != LCR_OBS_STATE;
    const size_t N = (size_t)cfg->n_envs;

    // ---- one arena for all SoA arrays (256-B aligned slices) ----
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    size_t off = 0;
    size_t o_qpos = off; off += al(sizeof(float) * s->nq * N);
    size_t o_qvel = off; off += al(sizeof(float) * s->nv * N);
    size_t o_ee = off; off += al(sizeof(float) * 3 * N);
    size_t o_tgt = off; off += al(sizeof(float) * 3 * N);
    // step outputs follow the observable state directly: [qpos .. did_reset] is ONE contiguous range, fetched by lcr_fetch_host
    // with a single device-to-host copy (the terminal observations behind it only when some env was reset)
    size_t o_rew = off; off += al(sizeof(float) * N);
    size_t o_term = off; off += al(N);
    size_t o_trunc = off; off += al(N);
    size_t o_succ = off; off += al(N);
    size_t o_dres = off; off += al(N);
    size_t o_fetch_end = off;
    size_t o_tobs = off; off += al(sizeof(float) * LCR_OBS_DIM * N);
    size_t o_tquat = off; off += al(sizeof(float) * 8 * N);
    size_t o_tobs_end = off;
    size_t o_el = off; off += al(sizeof(int) * N);
    size_t o_rng = off; off += al(sizeof(unsigned long long) * 4 * N);
    size_t o_goal = off; off += al(sizeof(int) * N);
    size_t o_time = off; off += al(sizeof(double) * N);
    size_t o_diag = off; if (cfg->diagnostics) off += 4 * al(sizeof(unsigned) * N) + al(sizeof(float) * 6 * N);
    // scratch: Stack on the one-wave kernels keeps the g rows of the arm-link proxy slot (+ the rolling rows of the finger slots) here, 24 / 48 floats per env;
    // the two-wave kernels compiled for two waves per SIMD hand Wm = (M + hD)^-1 L from the cube wave to the arm wave through it in substeps with a finger on
    // a cube, 36 floats per lane of every (64-lane) workgroup
    size_t o_scr = off; off += al(sizeof(float) * 48 * (((N + 63) / 64) * 64));
    const bool carry_warm = !(cfg->compat & LCR_COMPAT_COLD_SOLVE_EACH_STEP);
    size_t o_warm = off; if (carry_warm) off += al(sizeof(float) * LCR_NWARM * N);   // constraint forces carried between control steps
    size_t o_act = off; off += al(sizeof(float) * 6 * N);
    size_t o_mask = off; off += al(N);
    size_t o_seeds = off; off += al(sizeof(unsigned long long) * N);
    size_t o_img0 = off, o_img1 = off;
    const size_t img_bytes = (size_t)LCR_IMG_H * LCR_IMG_W * 3;
    size_t o_bg = off;
    if (s->has_images) { o_img0 = off; off += al(img_bytes * N); o_img1 = off; off += al(img_bytes * N); o_bg = off; off += al(2 * img_bytes); }
    s->arena_bytes = off;
    s->fetch_bytes = o_fetch_end;
    s->tobs_off = o_tobs;
    s->tobs_bytes = o_tobs_end - o_tobs;
    s->host_mirror = nullptr;
    e = hipMalloc(&s->arena, off);
    if (e != hipSuccess) { delete s; return fail(LCR_ERR_OOM, "hipMalloc(%zu bytes) failed: %s", off, hipGetErrorString(e)); }
    e = hipMemset(s->arena, 0, off);
    if (e != hipSuccess) { (void)hipFree(s->arena); delete s; return fail(LCR_ERR_HIP, "hipMemset failed: %s", hipGetErrorString(e)); }
    char *base = (char *)s->arena;

    LcrDev &D = s->dev;
    D.n = cfg->n_envs;
    D.k = k;
    D.task = cfg->task;
    D.n_substeps = cfg->n_substeps;
    D.max_steps = cfg->max_episode_steps;
    D.pgs_iters = cfg->pgs_iters;
    D.auto_reset = cfg->auto_reset ? 1 : 0;
    D.gripper_active = gripper_task ? 1 : 0;
    D.reward_type = cfg->reward_type;
    D.has_target = (cfg->task == LCR_TASK_PUSH || cfg->task == LCR_TASK_PICK_PLACE) ? 1 : 0;
    D.compat = cfg->compat;
    D.env_off = cfg->env_id_offset;
    D.dist_thr = (float)cfg->distance_threshold;
    D.height_thr = (float)cfg->height_threshold;
    D.inv_impratio = (float)(1.0 / (cfg->impratio > 1e-15 ? cfg->impratio : 1e-15));
    // scene constants: reach/lift/push cube 0.1 kg, I=1.6667e-4 (reach_cube.xml:25); pick_place 10 kg (pick_place_cube.xml:27);
    // stack 0.1 kg, I=1.125e-5 (stack_two_cubes.xml:27,33)
    // push_cube_loop.xml:29-31: 0.05 kg, I=1.125e-5, friction 1.5 / 1.5 (torsional)
    const bool loop = cfg->task == LCR_TASK_PUSH_LOOP;
    // (numbers from the scene files via tests/golden/model_golden.json -> lcr_model_gen.h)
    double cm = lcrm::SCENE_CUBE_MASS[cfg->task];
    double ci = lcrm::SCENE_CUBE_INERTIA[cfg->task];
    {
        const double mu = lcrm::SCENE_CUBE_MU[cfg->task], mut = lcrm::SCENE_CUBE_MU_TORS[cfg->task];   // cube geom friction (tangential, torsional)
        const double muf = mu > 1.5 ? mu : 1.5, muft = mut > 0.005 ? mut : 0.005;   // finger<->cube pair (both priority 1): max of both geoms (finger: follower.xml:15)
        D.rt_cube = (float)(mu * mu / (mut * mut));
        D.inv_mu_c2 = (float)(1.0 / (mu * mu));
        D.inv_mu_ct2 = (float)(1.0 / (mut * mut));
        D.rt_fc = (float)(muf * muf / (muft * muft));
        D.inv_mu_fct2 = (float)(1.0 / (muft * muft));
        const double mur = lcrm::SCENE_CUBE_MU_ROLL[cfg->task], mufr = mur > 0.0001 ? mur : 0.0001;   // rolling: max(cube, finger default 1e-4)
        D.rr_fc = (float)(muf * muf / (mufr * mufr));
        D.inv_mu_fcr2 = (float)(1.0 / (mufr * mufr));
        D.mu_c2 = (float)(mu * mu); D.mu_ct2 = (float)(mut * mut);
        D.mu_fc2 = (float)(muf * muf); D.mu_fct2 = (float)(muft * muft); D.mu_fcr2 = (float)(mufr * mufr);
        const bool roll_default = loop || cfg->task == LCR_TASK_STACK;
        D.roll = (cfg->finger_cube_condim == 6 || (cfg->finger_cube_condim == 0 && roll_default)) ? 1 : 0;   // 0 = the task's default
        // Stack shards of at most three waves per CU (MI355X: up to 49 152 envs; BASELINE config 5's per-GPU size is 32 768) run the kernel
        // variant that keeps every g row in LDS (46 / 52 KiB per wave, 3 x 52 <= 160 KiB); LCR_STACK_LDS=small|big overrides the choice
        // (tests exercise both variants at small sizes)
        D.big_lds = (cfg->task == LCR_TASK_STACK && (N + 63) / 64 <= 3 * (size_t)prop.multiProcessorCount) ? 1 : 0;
        if (const char *ov = getenv("LCR_STACK_LDS")) { if (cfg->task == LCR_TASK_STACK) D.big_lds = strcmp(ov, "big") == 0 ? 1 : (strcmp(ov, "small") == 0 ? 0 : D.big_lds); }
        D.walls = loop ? 1 : 0;
        // step-kernel family: a function of the task, the config and the size of the JOB (lcr_config.global_envs; 0 = this handle is the job) -- never of
        // the shard size, so that every sharding of a job runs the same arithmetic (bit-identical results for G = 1/2/4/8).  Measured on an
        // MI355X with the job as ONE shard (random policy, round 4): the two-cooperating-waves kernels win at every size for the
        // four one-cube tasks without rails (65 536 envs: ReachCube 0.257 against 0.285 ms, Push / Lift / PickPlace 0.308-0.312 against 0.333-0.336;
        // 32 768: 0.205-0.229 against 0.33) -- those tasks ALWAYS run them.  StackTwoCubes needs more than 256 registers per lane in its waves: up to
        // 32 768 envs (2 x 512 waves: one per SIMD) the two-wave kernels win (0.43 against 0.70 ms), above that one round of one-wave workgroups beats two
        // rounds of two-wave ones (65 536: 0.78 against 0.82).  PushCubeLoop has one kernel (lcr_kernels_loop.hip, one wave per 64 envs).
        // lcr_config.step_kernel pins a family (a Stack job cut into shards of <= 32 768 envs pins 2); LCR_STEP_KERNEL=single|coop1|coop2 overrides
        // (tests and profiling exercise every build).
        // WHICH BUILD of the two-wave family a shard runs does follow its size (one wave per SIMD while 2 x ceil(N / 64) waves fit the chip's SIMDs, else the
        // build compiled for two waves per SIMD): same source, same bits.
        {
            const size_t waves2 = 2 * ((N + 63) / 64), simds = 4 * (size_t)prop.multiProcessorCount;
            const int64_t job = cfg->global_envs > 0 ? cfg->global_envs : (int64_t)N;
            D.cc8 = (cfg->task == LCR_TASK_STACK && cfg->cc_points == 8) ? 1 : 0;
            // StackTwoCubes: two-wave workgroups while the JOB's 2 x job / 64 waves fit one per SIMD -- 32 envs per SIMD of the part the job runs on (MI355X: 1 024 SIMDs
            // -> 32 768 envs).  Every shard of a job runs on the same part and declares the same global_envs: the same family on all of them
            const int64_t stack_two_wave_max = 32 * (int64_t)simds;
            bool two_wave = job <= stack_two_wave_max || cfg->task != LCR_TASK_STACK;
            if (cfg->step_kernel == 1) two_wave = false;
            else if (cfg->step_kernel == 2) two_wave = true;
            if (D.cc8) two_wave = true;                               // the eight-point manifold lives in the two-wave kernels only
            if (cfg->pgs_iters < 0 || cfg->diagnostics == 2) two_wave = false;   // converged solver mode, per-wave cycle read-back: one-wave kernels only
            D.coop = two_wave ? (waves2 <= simds ? 1 : 2) : 0;
            if (const char *ov = getenv("LCR_STEP_KERNEL")) {
                if (strcmp(ov, "single") == 0 && !D.cc8) D.coop = 0;
                else if (strcmp(ov, "coop1") == 0 && two_wave_possible(cfg)) D.coop = 1;
                else if (strcmp(ov, "coop2") == 0 && two_wave_possible(cfg)) D.coop = 2;
            }
            if (D.cc8) D.coop = 1;   // (74-80 KiB of LDS per workgroup: its only build is the one-wave-per-SIMD one)
            if (loop) D.coop = 0;    // PushCubeLoop: one kernel
            D.newton = cfg->solver == LCR_SOLVER_NEWTON ? 1 : 0;
            D.newton_iters = cfg->newton_iters; D.ls_iters = cfg->ls_iters;
            D.newton_tol = (float)cfg->newton_tol; D.ls_tol = (float)cfg->ls_tol;
            // coupled envs a wave of the Newton kernels solves cooperatively (lcr_newton_coop.h: four per pass, one per 16-lane row; StackTwoCubes' three-body patients one at a time)
            // before it falls back to the coupled SIMT solves: 8 with one cube (two passes; flat beyond), 16 with rails and for Stack (PushCubeLoop 5.52 / 4.91 / 4.90 ms at 4 / 8 / 16: the
            // fall-back's 12-dim SIMT iteration spills; Stack's 18-dim one 3 KB per lane: profiles/r06_coop_sweep.txt, r06_coop_sweep2.txt); measurement override: LCR_COOP_MAX (0: never)
            D.coop_max = (cfg->task == LCR_TASK_STACK || loop) ? 16 : 8;
            if (const char *cm_ov = getenv("LCR_COOP_MAX")) D.coop_max = atoi(cm_ov) < 0 ? 0 : (atoi(cm_ov) > 64 ? 64 : atoi(cm_ov));
            if (D.newton) { D.coop = 0; D.roll = 1; D.big_lds = 1; }   // the Newton kernels: one wave per 64 envs, six-row finger slots, every g row in LDS (cc8: slots 4-7 of their eight cube<->cube records)
        }
    }
    D.arm_collision = cfg->arm_collision ? 1 : 0;
    D.diag = cfg->diagnostics;   // 2: max_sweeps carries the wave's cycle count instead (profiling aid, one-wave kernels); 3: per-wave phase cycles of the two-wave kernels
    D.pgs_tol = (float)cfg->pgs_tol;
    D.cube_mass = (float)cm;
    D.cube_minv = (float)(1.0 / cm);
    D.cube_iinv = (float)(1.0 / ci);
    {   // reach_cube_env.py:132-139, push_cube_env.py:141-148, pick_place_cube_env.py:143-150 -- same fp64 expressions
        double lo[3] = {-cfg->cube_xy_range / 2, -cfg->cube_xy_range / 2, 0}, hi[3] = {cfg->cube_xy_range / 2, cfg->cube_xy_range / 2, 0};
        lo[1] += 0.165; hi[1] += 0.10;
        double tl[3] = {-cfg->target_xy_range / 2, -cfg->target_xy_range / 2, 0};
        double th[3] = {cfg->target_xy_range / 2, cfg->target_xy_range / 2, cfg->task == LCR_TASK_PICK_PLACE ? cfg->goal_z_range : 0.0};
        tl[1] += 0.165; th[1] += 0.10;
        if (loop) {  // push_cube_loop_env.py:130-135 with push_cube_loop.xml:38: goal_region_high = size/2, [:2] -= 0.008, low = high*(-1,-1,1)
            double gh[3] = {0.035 / 2, 0.045 / 2, 0.007 / 2};
            gh[0] -= 0.008; gh[1] -= 0.008;
            for (int i = 0; i < 3; i++) { hi[i] = gh[i]; lo[i] = gh[i] * (i < 2 ? -1.0 : 1.0); }
        }
        for (int i = 0; i < 3; i++) { D.cube_lo[i] = lo[i]; D.cube_rng[i] = hi[i] - lo[i]; D.tgt_lo[i] = tl[i]; D.tgt_rng[i] = th[i] - tl[i]; }
    }
    D.qpos = (float *)(base + o_qpos);
    D.qvel = (float *)(base + o_qvel);
    D.ee_lag = (float *)(base + o_ee);
    D.target = (float *)(base + o_tgt);
    D.elapsed = (int *)(base + o_el);
    D.rng = (unsigned long long *)(base + o_rng);
    D.goal = (int *)(base + o_goal);
    D.sim_time = (double *)(base + o_time);
    D.reward = (float *)(base + o_rew);
    D.terminated = (unsigned char *)(base + o_term);
    D.truncated = (unsigned char *)(base + o_trunc);
    D.is_success = (unsigned char *)(base + o_succ);
    D.did_reset = (unsigned char *)(base + o_dres);
    D.term_obs = (float *)(base + o_tobs);
    D.term_quat = (float *)(base + o_tquat);
    D.active_mask = cfg->diagnostics ? (unsigned *)(base + o_diag) : nullptr;
    D.active_count = cfg->diagnostics ? (unsigned *)(base + o_diag + al(sizeof(unsigned) * N)) : nullptr;
    D.max_sweeps = cfg->diagnostics ? (unsigned *)(base + o_diag + 2 * al(sizeof(unsigned) * N)) : nullptr;
    D.choice = cfg->diagnostics ? (unsigned *)(base + o_diag + 3 * al(sizeof(unsigned) * N)) : nullptr;
    D.ctrl_out = cfg->diagnostics ? (float *)(base + o_diag + 4 * al(sizeof(unsigned) * N)) : nullptr;
    D.scratch = (float *)(base + o_scr);
    D.warm = carry_warm ? (float *)(base + o_warm) : nullptr;
    D.img_front = s->has_images ? (unsigned char *)(base + o_img0) : nullptr;
    D.img_top = s->has_images ? (unsigned char *)(base + o_img1) : nullptr;
    D.img_bg = s->has_images ? (unsigned char *)(base + o_bg) : nullptr;
    make_cameras(cfg->task, s->cam_front, s->cam_top, s->cam_vizu);
    s->render_dev = nullptr;
    s->render_bytes = 0;
    s->term_stage = nullptr;
    s->term_stage_bytes = 0;
    s->action_stage = (float *)(base + o_act);
    s->mask_dev = (unsigned char *)(base + o_mask);
    s->seeds_dev = (unsigned long long *)(base + o_seeds);
    e = hipEventCreate(&s->ev0);
    if (e == hipSuccess) e = hipEventCreate(&s->ev1);
    if (e != hipSuccess) { (void)hipFree(s->arena); delete s; return fail(LCR_ERR_HIP, "hipEventCreate failed: %s", hipGetErrorString(e)); }

    // initial state: reset of seed (base_seed + global env id) for every env
    int rc = lcr_launch_reset(D, nullptr, nullptr, 1, cfg->base_seed, s->stream);
    if (rc) { (void)hipFree(s->arena); delete s; return fail(LCR_ERR_HIP, "reset kernel launch failed: %s", hipGetErrorString((hipError_t)rc)); }
    if (s->has_images) {
        lcr_launch_render_bg(D, s->cam_front, s->cam_top, s->stream);
        lcr_launch_render_obs(D, s->cam_front, s->cam_top, s->stream);
    }
    e = hipStreamSynchronize(s->stream);
    if (e != hipSuccess) { (void)hipFree(s->arena); delete s; return fail(LCR_ERR_HIP, "initial reset failed: %s", hipGetErrorString(e)); }
    // the second stream of the frames (see lcr_sim); LCR_RENDER_OVERLAP=0: frames on the caller's stream, after the step kernel (A/B, profiling of one kernel at a time)
    const char *ro = getenv("LCR_RENDER_OVERLAP");
    if (s->has_images && !(ro && atoi(ro) == 0)) {
        e = hipStreamCreateWithFlags(&s->rstream, hipStreamNonBlocking);
        for (int p = 0; p < 2 && e == hipSuccess; p++) {
            e = hipEventCreateWithFlags(&s->ev_snap[p], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&s->ev_rdone[p], hipEventDisableTiming);
            if (e == hipSuccess) e = hipMalloc((void **)&s->snap_qpos[p], sizeof(float) * (size_t)s->nq * N);
            if (e == hipSuccess) e = hipMalloc((void **)&s->snap_target[p], sizeof(float) * 3 * N);
        }
        if (e != hipSuccess) { lcr_destroy(s); return fail(LCR_ERR_HIP, "setting up the frame stream failed: %s", hipGetErrorString(e)); }
    }
    *out = s;
    return LCR_OK;
}

void lcr_destroy(lcr_sim *s) {
    if (!s) return;
    (void)hipSetDevice(s->cfg.device);
    (void)hipStreamSynchronize(s->stream);
    if (s->rstream) {
        (void)hipStreamSynchronize(s->rstream);
        for (int p = 0; p < 2; p++) {
            (void)hipEventDestroy(s->ev_snap[p]); (void)hipEventDestroy(s->ev_rdone[p]);
            (void)hipFree(s->snap_qpos[p]); (void)hipFree(s->snap_target[p]);
        }
        (void)hipStreamDestroy(s->rstream);
    }
    (void)hipEventDestroy(s->ev0);
    (void)hipEventDestroy(s->ev1);
    if (s->render_dev) (void)hipFree(s->render_dev);
    if (s->term_stage) (void)hipFree(s->term_stage);
    if (s->host_mirror) (void)hipHostFree(s->host_mirror);
    (void)hipFree(s->arena);
    delete s;
}

#define SIMCHK_NOJOIN(s)                                      \
    if (!(s)) return fail(LCR_ERR_INVALID, "sim is NULL"); \
    HIPCHK(hipSetDevice((s)->cfg.device))
#define SIMCHK(s)       \
    SIMCHK_NOJOIN(s);   \
    if (int jr_ = join_render(s)) return fail(LCR_ERR_HIP, "waiting for the frame kernel failed: %s", hipGetErrorString((hipError_t)jr_))

int lcr_step_kernel_family(lcr_sim *s) {
    if (!s) return fail(LCR_ERR_INVALID, "sim is NULL");
    return s->dev.coop;
}

int lcr_set_stream(lcr_sim *s, void *hip_stream) {
    SIMCHK(s);
    if (s->rstream) HIPCHK(hipStreamSynchronize(s->rstream));   // (nothing of the old stream's frames is in flight when the new stream takes over)
    s->stream = (hipStream_t)hip_stream;
    return LCR_OK;
}

int lcr_sync(lcr_sim *s) {
    SIMCHK(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    return LCR_OK;
}

int lcr_reset(lcr_sim *s, const uint8_t *mask_host, const uint64_t *seeds_host) {
    SIMCHK(s);
    const size_t N = (size_t)s->dev.n;
    if (mask_host) HIPCHK(hipMemcpyAsync(s->mask_dev, mask_host, N, hipMemcpyHostToDevice, s->stream));
    if (seeds_host) HIPCHK(hipMemcpyAsync(s->seeds_dev, seeds_host, N * sizeof(uint64_t), hipMemcpyHostToDevice, s->stream));
    int rc = lcr_launch_reset(s->dev, mask_host ? s->mask_dev : nullptr, seeds_host ? s->seeds_dev : nullptr, 0, 0, s->stream);
    if (rc) return fail(LCR_ERR_HIP, "reset kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    if (s->has_images) {
        rc = lcr_launch_render_obs(s->dev, s->cam_front, s->cam_top, s->stream);
        if (rc) return fail(LCR_ERR_HIP, "render launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
    // the staging copies above read caller memory: do not return before they are consumed
    if (mask_host || seeds_host) HIPCHK(hipStreamSynchronize(s->stream));
    return LCR_OK;
}

int lcr_step(lcr_sim *s, const float *action_dev) {
    SIMCHK_NOJOIN(s);
    if (!action_dev) return fail(LCR_ERR_INVALID, "action is NULL");
    int rc = lcr_launch_step(s->dev, action_dev, s->ee_mode, s->stream);
    if (rc) return fail(LCR_ERR_HIP, "step kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    if (s->has_images && s->rstream) {
        // the frame kernel reads qpos and target only: snapshot them (2.7 MB for 32 768 StackTwoCubes envs against 15 GB of frames), then ray-cast on the second stream while
        // this stream goes on with the next step.  Two snapshots in turn; the one about to be overwritten was read by the frames of two steps ago.
        const int p = s->rpar;
        const size_t N = (size_t)s->dev.n;
        if (s->snap_used[p]) HIPCHK(hipStreamWaitEvent(s->stream, s->ev_rdone[p], 0));
        HIPCHK(hipMemcpyAsync(s->snap_qpos[p], s->dev.qpos, sizeof(float) * (size_t)s->nq * N, hipMemcpyDeviceToDevice, s->stream));
        HIPCHK(hipMemcpyAsync(s->snap_target[p], s->dev.target, sizeof(float) * 3 * N, hipMemcpyDeviceToDevice, s->stream));
        HIPCHK(hipEventRecord(s->ev_snap[p], s->stream));
        HIPCHK(hipStreamWaitEvent(s->rstream, s->ev_snap[p], 0));
        LcrDev R = s->dev;
        R.qpos = s->snap_qpos[p]; R.target = s->snap_target[p];
        rc = lcr_launch_render_obs(R, s->cam_front, s->cam_top, s->rstream);
        if (rc) return fail(LCR_ERR_HIP, "render launch failed: %s", hipGetErrorString((hipError_t)rc));
        HIPCHK(hipEventRecord(s->ev_rdone[p], s->rstream));
        s->snap_used[p] = true; s->rpending = true; s->rlast = p; s->rpar = p ^ 1;
    } else if (s->has_images) {
        rc = lcr_launch_render_obs(s->dev, s->cam_front, s->cam_top, s->stream);
        if (rc) return fail(LCR_ERR_HIP, "render launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
    return LCR_OK;
}

int lcr_step_host(lcr_sim *s, const float *action_host) {
    SIMCHK(s);
    if (!action_host) return fail(LCR_ERR_INVALID, "action is NULL");
    HIPCHK(hipMemcpyAsync(s->action_stage, action_host, sizeof(float) * (size_t)s->k * s->dev.n, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return lcr_step(s, s->action_stage);
}

int lcr_get_obs(lcr_sim *s, lcr_obs_view *out) {
    if (!s || !out) return fail(LCR_ERR_INVALID, "NULL argument");
    const size_t N = (size_t)s->dev.n;
    out->n_envs = s->dev.n;
    out->arm_qpos = s->dev.qpos;
    out->arm_qvel = s->dev.qvel;
    out->cube_pos = s->dev.qpos + 6 * N;
    if (s->dev.has_target) { out->has_aux = 1; out->aux_pos = s->dev.target; }
    else if (s->cfg.task == LCR_TASK_STACK) { out->has_aux = 1; out->aux_pos = s->dev.qpos + 13 * N; }
    else { out->has_aux = 0; out->aux_pos = nullptr; }
    out->image_front = s->dev.img_front;
    out->image_top = s->dev.img_top;
    return LCR_OK;
}

int lcr_get_outputs(lcr_sim *s, lcr_out_view *out) {
    if (!s || !out) return fail(LCR_ERR_INVALID, "NULL argument");
    out->n_envs = s->dev.n;
    out->_pad = 0;
    out->reward = s->dev.reward;
    out->terminated = s->dev.terminated;
    out->truncated = s->dev.truncated;
    out->is_success = s->dev.is_success;
    out->did_reset = s->dev.did_reset;
    out->terminal_obs = s->dev.term_obs;
    out->terminal_quat = s->dev.term_quat;
    out->timestamp = s->dev.sim_time;
    out->current_goal = s->dev.goal;
    out->active_mask = s->dev.active_mask;
    out->active_count = s->dev.active_count;
    out->max_sweeps = s->dev.max_sweeps;
    out->choice = s->dev.choice;
    out->ctrl = s->dev.ctrl_out;
    return LCR_OK;
}

int lcr_fetch_host(lcr_sim *s, lcr_host_view *out) {
    SIMCHK(s);
    if (!out) return fail(LCR_ERR_INVALID, "out is NULL");
    const size_t N = (size_t)s->dev.n;
    if (!s->host_mirror) {
        hipError_t e = hipHostMalloc((void **)&s->host_mirror, s->tobs_off + s->tobs_bytes, hipHostMallocDefault);
        if (e != hipSuccess) { s->host_mirror = nullptr; return fail(LCR_ERR_OOM, "hipHostMalloc(%zu) failed: %s", s->tobs_off + s->tobs_bytes, hipGetErrorString(e)); }
    }
    HIPCHK(hipMemcpyAsync(s->host_mirror, s->arena, s->fetch_bytes, hipMemcpyDeviceToHost, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    const char *hb = s->host_mirror, *db = (const char *)s->arena;
    auto H = [&](const void *dev) { return hb + ((const char *)dev - db); };
    const unsigned char *dres = (const unsigned char *)H(s->dev.did_reset);
    int any = 0;
    for (size_t i = 0; i < N; i++) any |= dres[i];
    if (any) {
        HIPCHK(hipMemcpyAsync(s->host_mirror + s->tobs_off, (const char *)s->arena + s->tobs_off, s->tobs_bytes, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    out->n_envs = s->dev.n;
    out->any_reset = any;
    out->arm_qpos = (const float *)H(s->dev.qpos);
    out->arm_qvel = (const float *)H(s->dev.qvel);
    out->cube_pos = (const float *)H(s->dev.qpos + 6 * N);
    out->aux_pos = s->dev.has_target ? (const float *)H(s->dev.target) : (s->cfg.task == LCR_TASK_STACK ? (const float *)H(s->dev.qpos + 13 * N) : nullptr);
    out->reward = (const float *)H(s->dev.reward);
    out->terminated = (const unsigned char *)H(s->dev.terminated);
    out->truncated = (const unsigned char *)H(s->dev.truncated);
    out->is_success = (const unsigned char *)H(s->dev.is_success);
    out->did_reset = dres;
    out->terminal_obs = (const float *)H(s->dev.term_obs);
    return LCR_OK;
}

int lcr_get_state(lcr_sim *s, double *qpos, double *qvel, double *ee_lag, float *target, int32_t *elapsed, uint64_t *rng,
                  int32_t *current_goal, double *sim_time, float *warm) {
    SIMCHK(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    const size_t N = (size_t)s->dev.n;
    std::vector<float> tmp;
    auto pull = [&](double *dst, const float *src, size_t cnt) -> hipError_t {
        tmp.resize(cnt);
        hipError_t e = hipMemcpy(tmp.data(), src, cnt * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess) for (size_t i = 0; i < cnt; i++) dst[i] = (double)tmp[i];
        return e;
    };
    if (qpos) HIPCHK(pull(qpos, s->dev.qpos, s->nq * N));
    if (qvel) HIPCHK(pull(qvel, s->dev.qvel, s->nv * N));
    if (ee_lag) HIPCHK(pull(ee_lag, s->dev.ee_lag, 3 * N));
    if (target) HIPCHK(hipMemcpy(target, s->dev.target, 3 * N * sizeof(float), hipMemcpyDeviceToHost));
    if (elapsed) HIPCHK(hipMemcpy(elapsed, s->dev.elapsed, N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (rng) HIPCHK(hipMemcpy(rng, s->dev.rng, 4 * N * sizeof(uint64_t), hipMemcpyDeviceToHost));
    if (current_goal) HIPCHK(hipMemcpy(current_goal, s->dev.goal, N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (sim_time) HIPCHK(hipMemcpy(sim_time, s->dev.sim_time, N * sizeof(double), hipMemcpyDeviceToHost));
    if (warm) {   // the carried constraint forces (mjData.qacc_warmstart of the reference's sim); zeros when nothing is carried
        if (s->dev.warm) HIPCHK(hipMemcpy(warm, s->dev.warm, sizeof(float) * LCR_NWARM * N, hipMemcpyDeviceToHost));
        else memset(warm, 0, sizeof(float) * LCR_NWARM * N);
    }
    return LCR_OK;
}

int lcr_set_state(lcr_sim *s, const double *qpos, const double *qvel, const double *ee_lag, const float *target,
                  const int32_t *elapsed, const uint64_t *rng, const int32_t *current_goal, const double *sim_time, const float *warm) {
    SIMCHK(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    const size_t N = (size_t)s->dev.n;
    std::vector<float> tmp;
    auto push = [&](float *dst, const double *src, size_t cnt) -> hipError_t {
        tmp.resize(cnt);
        for (size_t i = 0; i < cnt; i++) tmp[i] = (float)src[i];
        return hipMemcpy(dst, tmp.data(), cnt * sizeof(float), hipMemcpyHostToDevice);
    };
    if (qpos) HIPCHK(push(s->dev.qpos, qpos, s->nq * N));
    // a state set from outside without its constraint forces starts cold (zero forces in the first substep of the next step);
    // with them (warm != NULL: a checkpoint taken by lcr_get_state) the next step continues exactly where the saved sim would have
    if (s->dev.warm) {
        if (warm) HIPCHK(hipMemcpy(s->dev.warm, warm, sizeof(float) * LCR_NWARM * N, hipMemcpyHostToDevice));
        else if (qpos || qvel) HIPCHK(hipMemset(s->dev.warm, 0, sizeof(float) * LCR_NWARM * N));
    }
    if (qvel) HIPCHK(push(s->dev.qvel, qvel, s->nv * N));
    if (ee_lag) HIPCHK(push(s->dev.ee_lag, ee_lag, 3 * N));
    if (target) HIPCHK(hipMemcpy(s->dev.target, target, 3 * N * sizeof(float), hipMemcpyHostToDevice));
    if (elapsed) HIPCHK(hipMemcpy(s->dev.elapsed, elapsed, N * sizeof(int32_t), hipMemcpyHostToDevice));
    if (rng) HIPCHK(hipMemcpy(s->dev.rng, rng, 4 * N * sizeof(uint64_t), hipMemcpyHostToDevice));
    if (current_goal) HIPCHK(hipMemcpy(s->dev.goal, current_goal, N * sizeof(int32_t), hipMemcpyHostToDevice));
    if (sim_time) HIPCHK(hipMemcpy(s->dev.sim_time, sim_time, N * sizeof(double), hipMemcpyHostToDevice));
    return LCR_OK;
}

int lcr_malloc(lcr_sim *s, size_t bytes, void **dev_out) {
    SIMCHK(s);
    if (!dev_out) return fail(LCR_ERR_INVALID, "dev_out is NULL");
    hipError_t e = hipMalloc(dev_out, bytes);
    if (e != hipSuccess) return fail(LCR_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return LCR_OK;
}
int lcr_free(lcr_sim *s, void *dev) {
    SIMCHK(s);
    HIPCHK(hipFree(dev));
    return LCR_OK;
}
int lcr_memcpy_h2d(lcr_sim *s, void *dst_dev, const void *src_host, size_t bytes) {
    SIMCHK(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
    return LCR_OK;
}
int lcr_memcpy_d2h(lcr_sim *s, void *dst_host, const void *src_dev, size_t bytes) {
    SIMCHK(s);
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
    return LCR_OK;
}

int lcr_timer_begin(lcr_sim *s) {
    SIMCHK(s);
    HIPCHK(hipEventRecord(s->ev0, s->stream));
    return LCR_OK;
}
int lcr_timer_end(lcr_sim *s, float *ms_out) {
    SIMCHK(s);
    if (!ms_out) return fail(LCR_ERR_INVALID, "ms_out is NULL");
    HIPCHK(hipEventRecord(s->ev1, s->stream));
    HIPCHK(hipEventSynchronize(s->ev1));
    HIPCHK(hipEventElapsedTime(ms_out, s->ev0, s->ev1));
    return LCR_OK;
}

int lcr_fill_random_actions(lcr_sim *s, float *action_dev, uint64_t seed, uint64_t step) {
    SIMCHK_NOJOIN(s);   // (writes the caller's action buffer only)
    if (!action_dev) return fail(LCR_ERR_INVALID, "action is NULL");
    int rc = lcr_launch_fill_actions(action_dev, s->dev.n, s->k, s->dev.env_off, seed, step, s->stream);
    if (rc) return fail(LCR_ERR_HIP, "fill kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return LCR_OK;
}

int lcr_render(lcr_sim *s, int env, int camera, int width, int height, uint8_t *rgb_host) {
    SIMCHK(s);
    if (!rgb_host) return fail(LCR_ERR_INVALID, "rgb_host is NULL");
    if (env < 0 || env >= s->dev.n) return fail(LCR_ERR_INVALID, "env %d out of range", env);
    if (camera < 0 || camera > 2) return fail(LCR_ERR_INVALID, "camera must be 0 (front), 1 (top) or 2 (vizu)");
    if (width <= 0 || height <= 0 || (size_t)width * height > ((size_t)1 << 26)) return fail(LCR_ERR_INVALID, "bad frame size");
    LcrCam cam = camera == 0 ? s->cam_front : (camera == 1 ? s->cam_top : s->cam_vizu);
    cam.s = (float)(2.0 * std::tan(0.5 * 45.0 * M_PI / 180.0) / height);
    const size_t bytes = (size_t)width * height * 3;
    if (bytes > s->render_bytes) {
        if (s->render_dev) (void)hipFree(s->render_dev);
        s->render_dev = nullptr; s->render_bytes = 0;
        hipError_t e = hipMalloc((void **)&s->render_dev, bytes);
        if (e != hipSuccess) return fail(LCR_ERR_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        s->render_bytes = bytes;
    }
    int rc = lcr_launch_render_single(s->dev, cam, env, width, height, s->render_dev, s->stream);
    if (rc) return fail(LCR_ERR_HIP, "render launch failed: %s", hipGetErrorString((hipError_t)rc));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(rgb_host, s->render_dev, bytes, hipMemcpyDeviceToHost));
    return LCR_OK;
}

int lcr_render_state(lcr_sim *s, int camera, int width, int height, const double *qpos_host, const float *target_host, uint8_t *rgb_host) {
    SIMCHK(s);
    if (!rgb_host || !qpos_host) return fail(LCR_ERR_INVALID, "NULL argument");
    if (camera < 0 || camera > 2) return fail(LCR_ERR_INVALID, "camera must be 0 (front), 1 (top) or 2 (vizu)");
    if (width <= 0 || height <= 0 || (size_t)width * height > ((size_t)1 << 26)) return fail(LCR_ERR_INVALID, "bad frame size");
    LcrCam cam = camera == 0 ? s->cam_front : (camera == 1 ? s->cam_top : s->cam_vizu);
    cam.s = (float)(2.0 * std::tan(0.5 * 45.0 * M_PI / 180.0) / height);
    const size_t bytes = (size_t)width * height * 3, frame = (bytes + 255) & ~(size_t)255, need = frame + 256;
    if (need > s->render_bytes) {
        if (s->render_dev) (void)hipFree(s->render_dev);
        s->render_dev = nullptr; s->render_bytes = 0;
        hipError_t e = hipMalloc((void **)&s->render_dev, need);
        if (e != hipSuccess) return fail(LCR_ERR_OOM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        s->render_bytes = need;
    }
    // a one-env view of the handle whose state arrays are the caller's pose, staged behind the frame
    float st[32];
    for (int i = 0; i < s->nq; i++) st[i] = (float)qpos_host[i];
    for (int i = 0; i < 3; i++) st[s->nq + i] = target_host ? target_host[i] : 0.f;
    float *stage = (float *)(s->render_dev + frame);
    HIPCHK(hipMemcpyAsync(stage, st, sizeof(float) * (s->nq + 3), hipMemcpyHostToDevice, s->stream));
    LcrDev P1 = s->dev;
    P1.n = 1;
    P1.qpos = stage;
    P1.target = stage + s->nq;
    int rc = lcr_launch_render_single(P1, cam, 0, width, height, s->render_dev, s->stream);
    if (rc) return fail(LCR_ERR_HIP, "render launch failed: %s", hipGetErrorString((hipError_t)rc));
    HIPCHK(hipStreamSynchronize(s->stream));
    HIPCHK(hipMemcpy(rgb_host, s->render_dev, bytes, hipMemcpyDeviceToHost));
    return LCR_OK;
}

int lcr_render_terminal(lcr_sim *s, const int32_t *env_ids_host, int count, uint8_t *front_host, uint8_t *top_host) {
    SIMCHK(s);
    if (count < 0 || (count > 0 && (!env_ids_host || !front_host || !top_host))) return fail(LCR_ERR_INVALID, "NULL argument");
    if (!s->has_images) return fail(LCR_ERR_UNSUPPORTED, "terminal frames need observation_mode image / both (the frame background is only kept then)");
    for (int i = 0; i < count; i++)
        if (env_ids_host[i] < 0 || env_ids_host[i] >= s->dev.n) return fail(LCR_ERR_INVALID, "env id %d out of range", env_ids_host[i]);
    const size_t img = (size_t)LCR_IMG_H * LCR_IMG_W * 3;
    const int CHUNK = 1024;   // envs per pass: 2 x 225 KiB of frames each -> at most 450 MiB of staging
    const int cap = count < CHUNK ? count : CHUNK;
    if (cap == 0) return LCR_OK;
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t o_ids = 0, o_q = al(sizeof(int) * cap), o_t = o_q + al(sizeof(float) * s->nq * cap), o_f = o_t + al(sizeof(float) * 3 * cap),
                 o_tp = o_f + al(img * cap), need = o_tp + al(img * cap);
    if (need > s->term_stage_bytes) {
        if (s->term_stage) (void)hipFree(s->term_stage);
        s->term_stage = nullptr; s->term_stage_bytes = 0;
        hipError_t e = hipMalloc((void **)&s->term_stage, need);
        if (e != hipSuccess) return fail(LCR_ERR_OOM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e));
        s->term_stage_bytes = need;
    }
    char *base = (char *)s->term_stage;
    for (int done = 0; done < count; done += cap) {
        const int c = count - done < cap ? count - done : cap;
        HIPCHK(hipMemcpyAsync(base + o_ids, env_ids_host + done, sizeof(int) * c, hipMemcpyHostToDevice, s->stream));
        int rc = lcr_launch_gather_terminal(s->dev, (const int *)(base + o_ids), c, (float *)(base + o_q), (float *)(base + o_t), s->stream);
        if (rc) return fail(LCR_ERR_HIP, "gather launch failed: %s", hipGetErrorString((hipError_t)rc));
        LcrDev P1 = s->dev;   // a `c`-env view of the handle whose state arrays are the gathered terminal poses
        P1.n = c;
        P1.qpos = (float *)(base + o_q);
        P1.target = (float *)(base + o_t);
        P1.img_front = (unsigned char *)(base + o_f);
        P1.img_top = (unsigned char *)(base + o_tp);
        rc = lcr_launch_render_obs(P1, s->cam_front, s->cam_top, s->stream);
        if (rc) return fail(LCR_ERR_HIP, "render launch failed: %s", hipGetErrorString((hipError_t)rc));
        HIPCHK(hipMemcpyAsync(front_host + (size_t)done * img, base + o_f, img * c, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(top_host + (size_t)done * img, base + o_tp, img * c, hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return LCR_OK;
}

int lcr_calibrate_copy(lcr_sim *s, float *dst_dev, size_t n_floats) {
    SIMCHK(s);
    if (!dst_dev) return fail(LCR_ERR_INVALID, "dst is NULL");
    if (n_floats * sizeof(float) > s->arena_bytes) return fail(LCR_ERR_INVALID, "n_floats exceeds the state arena (%zu bytes)", s->arena_bytes);
    int rc = lcr_launch_calib_copy((const float *)s->arena, dst_dev, n_floats, s->stream);
    if (rc) return fail(LCR_ERR_HIP, "calibration kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
    return LCR_OK;
}

}  // extern "C"
