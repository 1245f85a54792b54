// lcr_device.h -- kernel argument block shared by the C ABI (lcr_capi.hip) and the kernels (lcr_kernels.hip).
#pragma once
#include <stdint.h>

#define LCR_OBS_DIM 18

constexpr int LCR_DEV_NWARM = 124;   // floats per env in LcrDev::warm (layout: lcr_kernels.hip WARM_*; = LCR_NWARM of include/lcr.h)

struct LcrDev {
    int n;            // envs on this device
    int k;            // action components
    int task;         // lcr_task
    int n_substeps;
    int max_steps;    // TimeLimit, <=0 disabled
    int pgs_iters;
    int auto_reset;
    int gripper_active;   // lift / pick_place / stack
    int reward_type;
    int has_target;
    unsigned compat;
    int _pad0;
    long long env_off;
    float dist_thr, height_thr, inv_impratio;
    float cube_mass, cube_minv, cube_iinv;
    // friction of the cube geom (reach_cube.xml:26: 0.5 / torsional 0.005; push_cube_loop.xml:31: 1.5 / 1.5) and of the
    // finger<->cube pair (max rule): ratios used by the elliptic-cone regularisation and projection
    float rt_cube;       // mu_tan^2 / mu_tors^2 of cube contacts
    float inv_mu_c2;     // 1 / mu_tan^2
    float inv_mu_ct2;    // 1 / mu_tors^2
    float rt_fc;         // finger<->cube: mu_tan^2 / mu_tors^2
    float inv_mu_fct2;   // finger<->cube: 1 / mu_tors^2
    int walls;           // PushCubeLoop rails
    int arm_collision;   // link-proxy spheres collide (D3)
    int diag;            // write active_mask / active_count / max_sweeps
    float pgs_tol;       // converged mode (pgs_iters < 0): sweep until max |df| <= pgs_tol (1 + max |f|)
    // reset sampling boxes, fp64 exactly as the reference builds them (reach_cube_env.py:132-139, push:141-148)
    double cube_lo[3], cube_rng[3], tgt_lo[3], tgt_rng[3];
    // state, SoA [component][n]
    float *qpos;      // [nq][n]
    float *qvel;      // [nv][n]
    float *ee_lag;    // [3][n]
    float *target;    // [3][n]
    int *elapsed;     // [n]
    unsigned long long *rng;  // [4][n]  PCG64 state_hi, state_lo, inc_hi, inc_lo
    int *goal;        // [n]  PushCubeLoop current_goal (persists across resets)
    double *sim_time; // [n]  accumulated simulation time (data.time is never reset by the reference)
    // step outputs
    float *reward;
    unsigned char *terminated, *truncated, *is_success, *did_reset;
    float *term_obs;  // [18][n]
    float *term_quat; // [8][n]  cube quaternion(s) of the terminal state (valid where did_reset)
    unsigned *active_mask, *active_count, *max_sweeps, *choice;  // [n] each, diagnostics of the last step (diag != 0)
    float *ctrl_out;  // [6][n] actuator targets of the last step (diag != 0)
    float *scratch;   // one-wave Stack kernels: g rows of the arm-link proxy slot, [12][n] float2; two-wave kernels at two waves per SIMD: Wm records, [groups][64][36]
    // image observations
    unsigned char *img_front, *img_top;  // [n][240][320][3] or null
    unsigned char *img_bg;               // [2][240][320][3] env-independent background of camera_front / camera_top, or null
    // (appended: the offsets of everything above are what the tuned default kernels were compiled against)
    float rr_fc;         // finger<->cube: mu_tan^2 / mu_roll^2   (ROLL kernels: lcr_config.finger_cube_condim = 6)
    float inv_mu_fcr2;   // finger<->cube: 1 / mu_roll^2
    int roll;            // 1: finger<->cube slots carry the two rolling rows
    float *warm;         // [LCR_NWARM][n] constraint forces carried from one control step to the next (warm start), or null
    int cc8;             // Stack: eight-point cube<->cube manifold (lcr_config.cc_points = 8; two-wave kernels only)
    int coop;            // 0: one wave per 64 envs (lcr_kernels.hip); 1 / 2: two cooperating waves per 64 envs (lcr_kernels2.hip) compiled for
                         // one / two waves per SIMD (<= 512 / <= 256 registers per lane)
    int big_lds;         // Stack: the shard has at most three waves per CU -> the variant that keeps every g row in LDS (46 / 52 KiB per wave)
    // squared friction coefficients (round 4: the contact blocks take a projected-gradient step in the variables f_j / mu_j, lcr_step_common.h soc_step)
    float mu_c2, mu_ct2;             // cube geom: tangential, torsional
    float mu_fc2, mu_fct2, mu_fcr2;  // finger<->cube pair (max rule): tangential, torsional, rolling
    // the faithful preset (round 5): Newton on the primal (lcr_newton.h), six-row finger contacts against cube AND floor
    int newton;                      // 1: lcr_config.solver = LCR_SOLVER_NEWTON
    int newton_iters, ls_iters;      // most Newton iterations per substep / most evaluations of phi' per line search
    float newton_tol, ls_tol;
    int coop_max;                    // Newton kernels: a wave solves up to this many coupled (arm on cube, cube on cube) envs one by one with all its lanes (lcr_newton_coop.h); more: the 12-dimensional SIMT solve
};

// pinhole camera: position, world axes (camera looks along -Z), s = 2 tan(fovy/2) / height
struct LcrCam {
    float px, py, pz;
    float xx, xy, xz, yx, yy, yz, zx, zy, zz;
    float s;
};

// launchers implemented in lcr_kernels.hip / lcr_render.hip (plain C++ linkage, same shared object)
int lcr_launch_step(const LcrDev &P, const float *action_dev, int ee_mode, void *stream);
// PushCubeLoop (lcr_kernels_loop.hip: one wave per 64 envs, row-wise solver)
int lcr_launch_step_loop(const LcrDev &P, const float *action_dev, int ee_mode, void *stream);
// the Newton kernels of the faithful preset (lcr_kernels.hip, unit LCR_PART = 4)
int lcr_launch_step_newton(const LcrDev &P, const float *action_dev, int ee_mode, void *stream);
int lcr_launch_step_newton_stack(const LcrDev &P, const float *action_dev, int ee_mode, void *stream);   // (unit LCR_PART = 5)
// two-cooperating-waves family (lcr_kernels2.hip); occ = waves per SIMD the variant is compiled for
int lcr_launch_step2_one_cube(const LcrDev &P, const float *action_dev, int ee_mode, int occ, void *stream);
int lcr_launch_step2_stack(const LcrDev &P, const float *action_dev, int ee_mode, int occ, void *stream);
int lcr_launch_step2_stack_cc8(const LcrDev &P, const float *action_dev, int ee_mode, int occ, void *stream);
int lcr_launch_reset(const LcrDev &P, const unsigned char *mask_dev, const unsigned long long *seeds_dev, int seed_from_base,
                     unsigned long long base_seed, void *stream);
int lcr_launch_fill_actions(float *action_dev, int n, int k, long long env_off, unsigned long long seed,
                            unsigned long long step, void *stream);
int lcr_launch_render_obs(const LcrDev &P, const LcrCam &front, const LcrCam &top, void *stream);
int lcr_launch_gather_terminal(const LcrDev &P, const int *ids_dev, int count, float *qpos_out, float *target_out, void *stream);
int lcr_launch_render_bg(const LcrDev &P, const LcrCam &front, const LcrCam &top, void *stream);
int lcr_launch_render_single(const LcrDev &P, const LcrCam &cam, int env, int W, int H, unsigned char *out_dev, void *stream);
int lcr_launch_calib_copy(const float *src, float *dst, size_t n, void *stream);
