/*
 * lcr.h -- C ABI of the MI355X-native batched low-cost-robot simulator (liblcr_hip.so).
 *
 * Drop-in boundary for the ONE hot path of perezjln/gym-lowcostrobot: batched reset()/step() of the
 * ReachCube / LiftCube / PushCube / PickPlaceCube / StackTwoCubes environments.  Each entry point
 * cites the reference interface it replaces (paths relative to /root/reference/gym_lowcostrobot/).
 *
 * Conventions
 *   - one handle (lcr_sim) per GPU; a handle is NOT thread-safe, distinct handles may be driven from
 *     distinct threads / processes (one process per GPU is the intended multi-GPU mode, no collectives).
 *   - every function returns 0 on success or a negative lcr_status; nothing throws across the boundary;
 *     lcr_last_error() gives a thread-local message for the last failure.
 *   - "dev" pointers are HIP device pointers on the handle's device; "host" pointers are ordinary memory.
 *   - all per-env arrays are SoA  [component][env]  with env fastest (coalesced: lane == env), fp32.
 *   - work is enqueued on the handle's HIP stream (lcr_set_stream) and is asynchronous unless stated.
 *   - there is NO CPU fallback: without a gfx950 device lcr_create fails with LCR_ERR_NO_DEVICE.
 */
#ifndef LCR_H
#define LCR_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LCR_ABI_VERSION 5

typedef enum lcr_status {
    LCR_OK = 0,
    LCR_ERR_INVALID = -1,     /* bad argument / config (maps to ValueError in the Python facade) */
    LCR_ERR_NO_DEVICE = -2,   /* no HIP device / wrong architecture */
    LCR_ERR_HIP = -3,         /* a HIP runtime call failed (message has hipGetErrorString) */
    LCR_ERR_OOM = -4,
    LCR_ERR_UNSUPPORTED = -5
} lcr_status;

/* gym_lowcostrobot/__init__.py:9-43 registry ids */
typedef enum lcr_task {
    LCR_TASK_REACH = 0,       /* ReachCube-v0       envs/reach_cube_env.py */
    LCR_TASK_LIFT = 1,        /* LiftCube-v0        envs/lift_cube_env.py */
    LCR_TASK_PUSH = 2,        /* PushCube-v0        envs/push_cube_env.py */
    LCR_TASK_PICK_PLACE = 3,  /* PickPlaceCube-v0   envs/pick_place_cube_env.py */
    LCR_TASK_STACK = 4,       /* StackTwoCubes-v0   envs/stack_two_cubes_env.py */
    LCR_TASK_PUSH_LOOP = 5    /* PushCubeLoop-v0    envs/push_cube_loop_env.py */
} lcr_task;

enum { LCR_ACTION_JOINT = 0, LCR_ACTION_EE = 1 };            /* action_mode   reach_cube_env.py:80 */
enum { LCR_OBS_IMAGE = 0, LCR_OBS_STATE = 1, LCR_OBS_BOTH = 2 }; /* observation_mode reach_cube_env.py:79 */
enum { LCR_REWARD_SPARSE = 0, LCR_REWARD_DENSE = 1 };        /* reward_type   reach_cube_env.py:81 */

/* compat bits: every reference quirk is reproduced when the bit is CLEAR */
enum {
    LCR_COMPAT_ZERO_QVEL_ON_RESET = 1u << 0, /* set: zero qvel in reset (deviates from reach_cube_env.py:297-311) */
    LCR_COMPAT_COLD_SOLVE_EACH_STEP = 1u << 1 /* set: the contact solver starts every control step from zero forces, so that a step is a pure
                                                 function of (qpos, qvel, action).  Clear (default): the forces of the last substep warm-start
                                                 the next control step, as MuJoCo's mjData.qacc_warmstart does across env.step calls (the
                                                 reference never resets it); lcr_reset / auto-reset / lcr_set_state clear them */
};

#define LCR_IMG_H 240
#define LCR_IMG_W 320

/* Constructor kwargs of the reference env classes (reach_cube_env.py:77-87, lift_cube_env.py:77-88,
 * push_cube_env.py:79-90, pick_place_cube_env.py:79-91, stack_two_cubes_env.py:78-88) plus the batch
 * / device placement the reference does not have. */
typedef struct lcr_config {
    uint32_t struct_size;      /* = sizeof(lcr_config), ABI check */
    int32_t task;              /* lcr_task */
    int32_t n_envs;            /* envs on this GPU, 1 .. 67 108 864 (2^26) per handle */
    int32_t device;            /* HIP device ordinal */
    int64_t env_id_offset;     /* global id of env 0 (sharding: GPU g owns [g*N/G, (g+1)*N/G)) */
    int32_t action_mode;
    int32_t obs_mode;
    int32_t reward_type;
    int32_t block_gripper;     /* -1 = task default (reach/push: 1, others: 0) */
    double distance_threshold; /* 0.05  (doubles: the reset sampling boxes are built in fp64 exactly as the */
    double cube_xy_range;      /* 0.3    reference does, reach_cube_env.py:132-139)                        */
    double target_xy_range;    /* 0.3 */
    double goal_z_range;       /* 0.1  (pick_place) */
    double height_threshold;   /* 0.1  (lift) */
    double impratio;           /* 100, follower.xml:3 */
    int32_t n_substeps;        /* 20 */
    int32_t max_episode_steps; /* 50, gymnasium TimeLimit configured at __init__.py:12-42; <=0 disables */
    int32_t pgs_iters;         /* warm-started PGS sweeps per substep, 4; < 0: "converged" mode -- sweep until the largest force
                                  change of a sweep is <= pgs_tol (1 + largest |force|) in every env of the wave, at most 50 sweeps */
    uint32_t compat;
    int32_t auto_reset;        /* 1: SB3 VecEnv semantics fused in the step kernel */
    int32_t arm_collision;     /* 1 (default): the arm links collide with the floor / cube through sphere proxies (follower.xml:10,13:
                                  every arm geom collides in the reference); 0: finger tips only */
    uint64_t base_seed;        /* envs never explicitly seeded use SeedSequence(base_seed + global env id) */
    double pgs_tol;            /* 1e-6; used when pgs_iters < 0 */
    int32_t diagnostics;       /* 0 | 1: lcr_out_view.active_mask / active_count / max_sweeps / choice / ctrl are written by every step (the decision
                                  signature).  2, 3: profiling aids -- the same arrays carry per-wave cycle counters instead (2: one-wave kernels,
                                  which it also selects; 3: phases of the two-wave kernels).  Anything else: LCR_ERR_INVALID */
    int32_t finger_cube_condim; /* rows of a finger<->cube contact.  6 = MuJoCo's: normal, two tangents, torsion, two rolling (follower.xml:15
                                  condim="6" wins the max rule over the cube's 4; rolling coefficient = max of both geoms).  4 = without the
                                  rolling rows (sweep kernels only, 8-12 % faster there).  lcr_config_default (= LCR_PRESET_FAITHFUL): 6 on every task,
                                  4 is LCR_ERR_UNSUPPORTED under LCR_SOLVER_NEWTON.  LCR_PRESET_FAST: 6 for PushCubeLoop (coefficient 1.5 m) and
                                  StackTwoCubes (light cubes), 4 for the other tasks (deviation D4).  0 = the preset's default */
    int32_t step_kernel;       /* LCR_SOLVER_PGS (LCR_PRESET_FAST) only -- the Newton kernels of the default preset are ONE family (one wave per 64 envs; 2 is
                                  LCR_ERR_UNSUPPORTED with them).  Which step-kernel family runs lcr_step: 0 = by task and JOB size (global_envs below, never the shard size n_envs): two
                                  cooperating waves per 64 envs for ReachCube / LiftCube / PushCube / PickPlaceCube at every size and for StackTwoCubes
                                  jobs of <= 32 envs per SIMD of the device (MI355X: 32 768 envs), one wave per 64 envs for larger Stack jobs -- the faster family when the job runs as ONE shard
                                  on an MI355X; 1 = one wave per 64 envs always; 2 = two cooperating waves always (the faster family on shards of
                                  <= 32 768 envs whatever the job size: a Stack job sharded that finely pins 2).  PushCubeLoop has ONE kernel (one wave
                                  per 64 envs, its own row-wise solver): 0 and 1 run it, 2 is LCR_ERR_UNSUPPORTED.  The families
                                  regroup the same arithmetic and agree to fp32 rounding (~1e-7 per control step), not bit for bit; WITHIN a family
                                  results are bit-identical for every sharding.  Because 0 looks at the job and not at the shard, every sharding of a
                                  job whose shards declare the same global_envs runs the same family and gives identical bits; which
                                  build of the family a shard runs (one / two waves per SIMD, rows in LDS / global scratch) does follow its size and
                                  does not change a bit.  The Stack family boundary is counted in SIMDs of the device the handle lives on: a job replayed on
                                  another part may pick the other family (fp32 rounding apart, not bit for bit). */
    int32_t cc_points;         /* StackTwoCubes: cube<->cube manifold points kept per substep.  4 (default, 0 = default): the extremes along the diagonals
                                  of the reference face; 8: also the extremes along its two axes -- as many points as MuJoCo's box-box
                                  collider may return (stack_two_cubes.xml:25-35; narrows deviation D5).  8 runs on the
                                  two-cooperating-waves kernels (step_kernel = 1 with it: LCR_ERR_UNSUPPORTED); other tasks: LCR_ERR_INVALID */
    int64_t global_envs;       /* ABI v4: number of envs of the whole JOB this handle is one shard of (all GPUs together); 0 = n_envs (the handle is
                                  the job).  Must be >= env_id_offset + n_envs.  The reference has one independent MjData per env
                                  (reach_cube_env.py:89-90), so how a batch is cut into shards must not show in the results: every sharding of a job
                                  gives identical bits PROVIDED the shards are cut at wave boundaries -- a wave (64 consecutive env ids) skips work no
                                  lane needs, solves its coupled envs cooperatively and leaves the solver loops for all its lanes at once, so an env's
                                  low-order bits depend on its 63 wave-mates.  lcr_create therefore refuses (LCR_ERR_INVALID) a handle whose
                                  env_id_offset is not a multiple of 64 and a shard that, not being the job's last (env_id_offset + n_envs <
                                  global_envs), does not hold a multiple of 64 envs.  The step_kernel = 0 dispatch reads global_envs as well (see there) */
    /* ABI v5 (round 5): the solver of the constraint problem and the rows of a finger<->floor contact.  See lcr_config_preset. */
    int32_t solver;            /* lcr_solver.  LCR_SOLVER_NEWTON: Newton's method on the primal problem, all accelerations at once, warm-started from the carried
                                  constraint forces -- MuJoCo's default solver (follower.xml:3 names none); reaches the optimum of MuJoCo's convex constraint problem to
                                  float rounding (tools/kkt_distance.py).  LCR_SOLVER_PGS: pgs_iters warm-started sweeps of a block projected-gradient step on the dual
                                  problem (rounds 1-4; p90 2e-4 / p99 1e-2 rad per control step away from that optimum at four sweeps). */
    int32_t newton_iters;      /* LCR_SOLVER_NEWTON: most iterations per substep (30: what bounds a cold start on a hard contact set -- a finger set 5 mm into the floor -- warm-started, an env needs one on average and seven at the 99th percentile); a wave leaves the loop when every one of its envs has converged */
    int32_t ls_iters;          /* ... most evaluations of phi' per line search (8) */
    int32_t finger_floor_condim; /* rows of a finger<->floor contact: 6 = MuJoCo's (follower.xml:15 condim="6": + two rolling rows, coefficient 1e-4 m), 4 = without
                                  them.  0 = the preset's default.  6 is implemented by the Newton kernels (LCR_SOLVER_PGS with 6: LCR_ERR_UNSUPPORTED) */
    double newton_tol;         /* LCR_SOLVER_NEWTON: an env has converged when its Newton decrement -g'dx <= newton_tol^2 (1 + |a0|_M^2)   (1e-6) */
    double ls_tol;             /* ... its line search stops when |phi'(al)| <= ls_tol |phi'(0)|   (1e-2: MuJoCo's default ls_tolerance), plus a rounding floor 1e-5 (|M-part| + |force part|) of the two sums phi' is the difference of */
} lcr_config;

typedef enum lcr_solver { LCR_SOLVER_PGS = 0, LCR_SOLVER_NEWTON = 1 } lcr_solver;
/* LCR_PRESET_FAITHFUL (what lcr_config_default fills in): the reference's contact model as its MJCF states it -- six-row finger contacts against cube AND floor
 * (follower.xml:15), up to eight box-box points (stack_two_cubes.xml:25-35), elliptic cones -- solved by Newton's method (follower.xml:3).
 * What remains approximate under it: finger pads are boxes fitted to the hull tips and the other arm hulls five sphere proxies (deviation D3), each finger has one
 * world contact (floor or rail), Newton is capped at newton_iters = 30 iterations per substep, fp32 arithmetic.
 * LCR_PRESET_FAST: the rounds 1-4 configuration -- four block projected-gradient sweeps, rolling rows only where they change a step by more than the fp32
 * parity tolerance, four box-box points -- 8 x (ReachCube, PushCubeLoop) to 13 x (StackTwoCubes) the throughput of the default (measured per task: profiles/r06_quick_times.txt)
 * at p90 2e-4 / p99 1e-2 rad per control step from the optimum.  Options of the sweep kernels (step_kernel = 2, pgs_iters < 0,
 * diagnostics = 3, finger_cube_condim = 4) are refused under LCR_SOLVER_NEWTON: set them on a config filled by lcr_config_preset(.., LCR_PRESET_FAST). */
typedef enum lcr_preset { LCR_PRESET_FAITHFUL = 0, LCR_PRESET_FAST = 1 } lcr_preset;

typedef struct lcr_sim lcr_sim;

/* Read-only device views.  arm_qpos/arm_qvel/cube_pos (and cube_blue_pos for Stack) alias the state
 * arrays themselves (get_observation() reach_cube_env.py:281-295 returns exactly those qpos/qvel
 * slices cast to float32); target_pos aliases the per-env target (push_cube_env.py:297). */
typedef struct lcr_obs_view {
    int32_t n_envs;
    int32_t has_aux;            /* 1 if aux_pos is meaningful (push/pick_place: target_pos; stack: cube_blue_pos) */
    const float *arm_qpos;      /* [6][N] */
    const float *arm_qvel;      /* [6][N] */
    const float *cube_pos;      /* [3][N]  (stack: cube_red_pos) */
    const float *aux_pos;       /* [3][N]  or NULL */
    const uint8_t *image_front; /* [N][240][320][3] or NULL (observation_mode image/both); approximate ray-cast, see lcr_render.hip */
    const uint8_t *image_top;   /* [N][240][320][3] or NULL */
} lcr_obs_view;

/* step() return values (reach_cube_env.py:313-333) + SB3 auto-reset bookkeeping */
typedef struct lcr_out_view {
    int32_t n_envs;
    int32_t _pad;
    const float *reward;        /* [N] */
    const uint8_t *terminated;  /* [N] */
    const uint8_t *truncated;   /* [N]  TimeLimit */
    const uint8_t *is_success;  /* [N]  info["is_success"] (lift: always 0, reference returns info={}) */
    const uint8_t *did_reset;   /* [N]  1 where the env was auto-reset at the end of this step */
    const float *terminal_obs;  /* [18][N] arm_qpos6, arm_qvel6, cube_pos3, aux3 -- valid where did_reset */
    const float *terminal_quat; /* [8][N]  cube quaternion(s) of the terminal state (with terminal_obs: the full terminal pose) */
    const double *timestamp;    /* [N]  accumulated simulation time = info["timestamp"] of PushCubeLoop-v0 (push_cube_loop_env.py:328) */
    const int32_t *current_goal;/* [N]  PushCubeLoop-v0 goal side (0|1), persists across resets (push_cube_loop_env.py:136,341) */
    /* solver diagnostics of the last step, valid when lcr_config.diagnostics != 0 (else NULL): bit s of active_mask = constraint
     * slot s was active in some substep (0-7 floor<->cube, 8-11 cube<->cube / rails, 12-13 finger<->cube, 14-15 finger<->floor,
     * 16 arm-link proxies, 18+j joint limit j); active_count = number of (slot, substep) activations;
     * max_sweeps = most PGS sweeps of a substep; choice = wrapping sum over substeps s (weight 2s+1) and active constraints of
     * (slot + 1)(sel + 1) 2654435761 with sel the discrete choice behind the contact (vertex index, manifold candidate, box
     * face, proxy member, limit side) + 0x9E3779B1 x executed IK iterations: two runs that agree in these four words went
     * through the same sequence of discrete decisions */
    const uint32_t *active_mask, *active_count, *max_sweeps, *choice; /* [N] each */
    const float *ctrl;          /* [6][N] actuator targets data.ctrl as apply_action left them (reach_cube_env.py:273); diagnostics only */
} lcr_out_view;

/* Host-side mirror of everything a host vector env needs after a step (lcr_fetch_host): pointers into a pinned buffer owned
 * by the handle, SoA [component][N] like the device views, valid until the next lcr_fetch_host / lcr_destroy. */
typedef struct lcr_host_view {
    int32_t n_envs;
    int32_t any_reset;          /* 1 if some env was auto-reset in the last step (terminal_obs is then current) */
    const float *arm_qpos;      /* [6][N] */
    const float *arm_qvel;      /* [6][N] */
    const float *cube_pos;      /* [3][N] */
    const float *aux_pos;       /* [3][N] or NULL */
    const float *reward;        /* [N] */
    const uint8_t *terminated, *truncated, *is_success, *did_reset; /* [N] each */
    const float *terminal_obs;  /* [18][N], copied only when any_reset */
} lcr_host_view;

/* Environment variables the library reads (measurement / A-B overrides, none needed in production; all read at lcr_create unless stated):
 *   LCR_COOP_MAX=n        Newton kernels: coupled envs per wave solved cooperatively before the wave falls back to the coupled SIMT solves (default 8; 16 for
 *                         StackTwoCubes and PushCubeLoop; 0 = never cooperatively)
 *   LCR_RENDER_OVERLAP=0  image observations: frames on the handle's stream after the step kernel instead of on the second stream (see lcr_step)
 *   LCR_STEP_KERNEL=single|coop1|coop2, LCR_STACK_LDS=small|big   sweep kernels (LCR_PRESET_FAST): pin a kernel family / LDS variant
 *   LCR_RENDER_COUNT=1    frame kernel: count ray-cast passes into the diagnostics arrays (tools/render_work.py; read at the first frame launch) */
int lcr_abi_version(void);
const char *lcr_last_error(void);

/* Fill `cfg` with the reference constructor defaults for `task`. */
int lcr_config_default(lcr_config *cfg, int task);
/* The reference constructor defaults for `task` with the solver / contact-row settings of `preset` (lcr_preset). */
int lcr_config_preset(lcr_config *cfg, int task, int preset);
/* Number of action components k for a config: {joint:5, ee:3} + (0 if block_gripper else 1)  (reach_cube_env.py:95-96) */
int lcr_action_dim(const lcr_config *cfg);
int lcr_nq(int task); /* 13, stack 20 */
int lcr_nv(int task); /* 12, stack 18 */

/* == EnvClass.__init__ (reach_cube_env.py:77-139): allocate device state for n_envs envs.  The envs are
 * left in the post-reset state of seed (base_seed + global env id). */
int lcr_create(const lcr_config *cfg, lcr_sim **out);
void lcr_destroy(lcr_sim *sim); /* == close() reach_cube_env.py:357-363 */

/* Which step-kernel family this handle runs (decided at lcr_create from lcr_config.step_kernel and the shard size): 0 = one wave per 64 envs
 * (lcr_step_kernel), 1 / 2 = two cooperating waves per 64 envs (lcr_step2_kernel) compiled for one / two waves per SIMD. */
int lcr_step_kernel_family(lcr_sim *sim);

/* HIP stream (hipStream_t passed as void*) all later work is enqueued on; NULL = default stream. */
int lcr_set_stream(lcr_sim *sim, void *hip_stream);
int lcr_sync(lcr_sim *sim); /* hipStreamSynchronize on the handle's stream (after making it wait for frames still being ray-cast, see lcr_step) */

/* == reset(seed) (reach_cube_env.py:297-311, push:308-328, pick_place:316-336, stack:307-324).
 * mask_host: N bytes, nonzero = reset that env, NULL = all.  seeds_host: N uint64, env i is re-seeded
 * with numpy's Generator(PCG64(SeedSequence(seeds[i]))) before sampling; NULL = continue each env's
 * generator stream (gymnasium reset(seed=None) semantics). */
int lcr_reset(lcr_sim *sim, const uint8_t *mask_host, const uint64_t *seeds_host);

/* == step(action) (reach_cube_env.py:313-333) for all envs: apply_action (joint or ee+IK) -> 20 physics
 * substeps -> reward / terminated / truncated -> fused auto-reset.  action_dev: [k][N] float32.
 * Asynchronous.  With image observations the two frames of every env are ray-cast on a second, internal stream from a snapshot of the poses, so that the step kernel of
 * the NEXT lcr_step overlaps them (BASELINE config 5: 9.7 -> see DESIGN.md section 3.4).  Every other entry point of this API first makes the handle's stream wait for
 * those frames; a caller that reads lcr_obs_view.image_* with its own kernels on the handle's stream calls lcr_sync (or any other entry point) first.
 * LCR_RENDER_OVERLAP=0 in the environment: frames on the handle's stream, after the step kernel. */
int lcr_step(lcr_sim *sim, const float *action_dev);
/* Convenience for host callers (single-env facade): copies [k][N] host floats then steps. */
int lcr_step_host(lcr_sim *sim, const float *action_host);

int lcr_get_obs(lcr_sim *sim, lcr_obs_view *out);
/* == what DummyVecEnv.step_wait hands to SB3 (examples/gym_manipulation_sb3.py:34-39): state observations + rewards + flags of
 * all envs in ONE device-to-host copy into pinned memory (132 B/env), plus the terminal observations (72 B/env) only when some
 * env was reset.  Synchronises the handle's stream. */
int lcr_fetch_host(lcr_sim *sim, lcr_host_view *out);
int lcr_get_outputs(lcr_sim *sim, lcr_out_view *out);

/* Full simulator state (replaces poking env.data.qpos / env.data.qvel, e.g. examples/dynamixel_gym_leader.py:96-98;
 * also checkpoint/resume and the "(qpos, qvel, action) triple" parity tests).  Host pointers, any may be
 * NULL, SoA [component][N]; synchronous.
 * `warm` (ABI v3) is the solver state the reference keeps in mjData.qacc_warmstart between env.step calls
 * (reach_cube_env.py:276-279 never resets it): the constraint forces of the last substep, [LCR_NWARM][N] float32 --
 *   rows  0..31  floor<->cube      [cube c][vertex slot s][row k]   at 16 c + 4 s + k   (rows: normal, t1, t2, torsion)
 *   rows 32..61  arm-coupled slots [slot s][row k]                  at 32 + 6 s + k     (s: 0,1 finger<->cube, 2,3 finger<->floor,
 *                                                                                           4 arm-link proxies; k < 4, or 6 with rolling rows)
 *   rows 62..67  joint limits      [joint j]                        at 62 + j
 *   rows 68..83  rails (PushCubeLoop) [slot s][row k]               at 68 + 4 s + k
 *   rows 84..99  cube<->cube (Stack)  [slot s][row k]               at 84 + 4 s + k
 *   rows 100..103 cube<->cube slot s was active in the last substep (0.0 / 1.0)
 *   rows 104..119 cube<->cube slots 4..7 of the eight-point manifold (cc_points = 8) [slot s - 4][row k]  at 104 + 4 (s - 4) + k
 *   rows 120..123 their "was active" flags
 * lcr_get_state + lcr_set_state with all arrays including `warm` is an exact checkpoint: the next lcr_step is bit-identical to
 * the one the un-checkpointed sim would have made.  lcr_set_state with qpos or qvel but warm == NULL clears the carried forces
 * (cold solve in the first substep of the next step); with LCR_COMPAT_COLD_SOLVE_EACH_STEP nothing is carried: get returns zeros,
 * set ignores `warm`. */
#define LCR_NWARM 124
int lcr_get_state(lcr_sim *sim, double *qpos /*[nq][N]*/, double *qvel /*[nv][N]*/, double *ee_lag /*[3][N]*/,
                  float *target /*[3][N]*/, int32_t *elapsed /*[N]*/, uint64_t *rng /*[4][N]*/,
                  int32_t *current_goal /*[N]*/, double *sim_time /*[N]*/, float *warm /*[LCR_NWARM][N]*/);
int lcr_set_state(lcr_sim *sim, const double *qpos, const double *qvel, const double *ee_lag, const float *target,
                  const int32_t *elapsed, const uint64_t *rng, const int32_t *current_goal, const double *sim_time,
                  const float *warm);

/* small device-memory helpers so a ctypes/numpy caller needs no other GPU library */
int lcr_malloc(lcr_sim *sim, size_t bytes, void **dev_out);
int lcr_free(lcr_sim *sim, void *dev);
int lcr_memcpy_h2d(lcr_sim *sim, void *dst_dev, const void *src_host, size_t bytes); /* synchronous */
int lcr_memcpy_d2h(lcr_sim *sim, void *dst_host, const void *src_dev, size_t bytes); /* synchronous */

/* HIP-event timing on the handle's stream: begin, enqueue work, end (synchronises) -> milliseconds */
int lcr_timer_begin(lcr_sim *sim);
int lcr_timer_end(lcr_sim *sim, float *ms_out);

/* Fill action_dev [k][N] with U(-1,1) from a counter-based generator keyed (seed, global env id, step):
 * the synthetic policy of the benchmark; shard-invariant by construction. */
int lcr_fill_random_actions(lcr_sim *sim, float *action_dev, uint64_t seed, uint64_t step);

/* == render() with render_mode="rgb_array" (reach_cube_env.py:350-355: 640x640 frame of camera_vizu) and ad-hoc frames of
 * the observation cameras: ray-cast env `env` from camera 0 (camera_front), 1 (camera_top) or 2 (camera_vizu) at
 * width x height into rgb_host[height][width][3].  Synchronous. */
int lcr_render(lcr_sim *sim, int env, int camera, int width, int height, uint8_t *rgb_host);
/* The same for an arbitrary pose given by the caller (qpos_host[nq] as env.data.qpos, target_host[3] or NULL): e.g. the last
 * frame of an episode whose env the step kernel has already reset (terminal_obs + terminal_quat).  Does not touch the sim state. */
int lcr_render_state(lcr_sim *sim, int camera, int width, int height, const double *qpos_host, const float *target_host, uint8_t *rgb_host);

/* Batched last frames of finished episodes (what DummyVecEnv puts into infos[i]["terminal_observation"]["image_front" / "image_top"],
 * examples/gym_manipulation_sb3.py:34-39 with observation_mode image / both; reach_cube_env.py:288-292): the step kernel has already reset
 * those envs, so their frame buffers show the reset state; this draws camera_front / camera_top of the TERMINAL poses (terminal_obs +
 * terminal_quat of the last step) of the `count` listed envs with the observation ray-caster, as one batch, into
 * front_host / top_host [count][240][320][3].  Needs observation_mode image / both.  Synchronous.
 * PRECONDITION: every listed env was reset by the LAST lcr_step (out.did_reset[id] != 0) -- the terminal pose arrays are written only by lanes that auto-reset, an
 * env that did not finish shows the last frame of an OLDER episode (or zeros before its first reset).  The library does not re-read did_reset here; the Python
 * binding (VecSim.render_terminal) checks it and raises ValueError. */
int lcr_render_terminal(lcr_sim *sim, const int32_t *env_ids_host, int count, uint8_t *front_host, uint8_t *top_host);

/* Measurement support: copy n_floats floats from the start of the state arena to dst_dev with one dword load and
 * one dword store per lane (the step kernel's access pattern): a launch with a KNOWN byte count (4*n read, 4*n
 * written) against which rocprofv3 FETCH_SIZE / WRITE_SIZE are calibrated (MI355X_MICROARCH.md, HBM section). */
int lcr_calibrate_copy(lcr_sim *sim, float *dst_dev, size_t n_floats);

#ifdef __cplusplus
}
#endif
#endif /* LCR_H */
