// lcr_newton.h -- building blocks of the FAITHFUL preset's contact solve: Newton's method on the primal problem, MuJoCo's default solver
// (follower.xml:3 names no solver).  CPU checker: newton_product of the test oracle (orc_params.solver = 2); decision record: profiles/r05_solver_decision.txt.
//
// The constrained accelerations x minimise the strictly convex, C^1, piecewise quadratic
//     F(x) = 1/2 (x - a0)' M (x - a0) + sum_b s_b(J_b x - aref_b),      s_b(z) = max_{f in K_b} ( -f'z - 1/2 f'R_b f ),
// the constraint forces are f_b = argmax.  Unknowns here: the arm in the coordinates y = L' qacc (M = L L': its metric is the identity) and per cube the linear
// and angular acceleration (metric: mass, isotropic inertia).  One iteration: gradient g and Hessian H = M + J'WJ at x (W: Jacobian of -f w.r.t. the row
// residuals), dx = -H^-1 g by Cholesky, line search on phi'(al) = grad F(x + al dx).dx (one gradient pass per evaluation gives phi' and phi'': safeguarded Newton
// steps on phi' from both ends of the bracket, see the loop), x += al dx.  Rounds 1-4 swept per-contact blocks of the DUAL problem; what those sweeps cannot resolve in any sane number of
// passes is the redundancy of contacts that share a body (two fingers on the floor, the four vertices of a resting cube) -- here that is one 12 x 12 factorisation.
#pragma once
#include "lcr_step_common.h"

namespace {

// ---- one contact block in the scaled variables (f_n, f_j / mu_j), w = (z_n, mu_j z_j), N = |w_t| (MJ-DOC: elliptic cone, regularised) ----
//   top    (w_n >= N):              f = 0
//   bottom (N Rn <= -w_n Rt):       f_n = -w_n / Rn,  f~_t = -w_t / Rt                  (inside the cone: plain quadratic)
//   middle (otherwise):             f_n = (N - w_n) / (Rn + Rt),  f~_t = -f_n w_t / N    (on the cone's surface)
// Rn: regulariser of the normal row; Rt = R_friction mu_tan^2 (the same for every friction row in the scaled variables).
// Everything is expressed with the SQUARED friction coefficients m2[r] (row r; m2[0] unused; an absent row has m2 = 0 and contributes nothing).
// J'WJ of the block = av v v' - gam w w' + sum_t kap m2[t] row_t row_t',  w = sum_t c[t] row_t,  v = row_n - w   (c = 0 outside the middle zone).
template <int NR>
struct BlkEval {
    float f[NR];   // forces (N resp. N m)
    float c[NR];   // middle zone: mu_t w_t / N, else 0   (c[0] unused)
    float av, kap, gam;
};
template <int NR>
DEV void blk_eval(const float (&z)[NR], float Rn, float Rt, const float (&m2)[NR], bool act, BlkEval<NR> &B) {
    const float w0 = z[0];
    float N2 = 0.f;
#pragma unroll
    for (int r = 1; r < NR; r++) N2 = fmaf(m2[r] * z[r], z[r], N2);
    const float rs = rsq(fmaxf(N2, 1e-30f)), N = N2 * rs;
    const bool top = !act || !(w0 < N);
    const bool bottom = !top && (N * Rn <= -w0 * Rt);
    const bool middle = !top && !bottom;
    const float iRn = rcp(Rn), iRt = rcp(Rt), iD = rcp(Rn + Rt);
    const float y0 = top ? 0.f : (bottom ? -w0 * iRn : (N - w0) * iD);
    const float st = top ? 0.f : (bottom ? -iRt : -y0 * rs);   // f~_t = st w_t
    B.f[0] = y0;
    B.c[0] = 0.f;
#pragma unroll
    for (int r = 1; r < NR; r++) {
        const float mz = m2[r] * z[r];
        B.f[r] = st * mz;
        B.c[r] = middle ? mz * rs : 0.f;
    }
    B.av = top ? 0.f : (bottom ? iRn : iD);
    B.kap = top ? 0.f : (bottom ? iRt : y0 * rs);
    B.gam = middle ? y0 * rs : 0.f;
}

// ---- packed lower-triangular symmetric matrix in registers: H(i, j), j <= i, at i (i + 1) / 2 + j (all indices are literals after unrolling) ----
constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
// H += w v v' restricted to the index range [LO, HI)
template <int LO, int HI, int NX>
DEV void h_rank1(float (&H)[NX * (NX + 1) / 2], const float (&v)[NX], float w) {
#pragma unroll
    for (int i = LO; i < HI; i++) {
        const float t = w * v[i];
#pragma unroll
        for (int j = LO; j <= i; j++) H[tri(i, j)] = fmaf(t, v[j], H[tri(i, j)]);
    }
}
// the block's J'WJ from its rows (dense vectors in x space, entries outside [LO, HI) are zero by construction)
template <int LO, int HI, int NX, int NR>
DEV void h_block(float (&H)[NX * (NX + 1) / 2], const float (&row)[NR][NX], const BlkEval<NR> &B, const float (&m2)[NR]) {
    float wv[NX], v[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) { wv[i] = 0.f; v[i] = 0.f; }
#pragma unroll
    for (int i = LO; i < HI; i++) {
        float a = 0.f;
#pragma unroll
        for (int r = 1; r < NR; r++) a = fmaf(B.c[r], row[r][i], a);
        wv[i] = a;
        v[i] = row[0][i] - a;
    }
    h_rank1<LO, HI, NX>(H, v, B.av);
    h_rank1<LO, HI, NX>(H, wv, -B.gam);
#pragma unroll
    for (int r = 1; r < NR; r++) h_rank1<LO, HI, NX>(H, row[r], B.kap * m2[r]);
}
// in-place Cholesky H = L L' (strictly lower part of L stays in H, id = 1 / L_ii); pivots are kept away from zero (H = M + PSD: positive in exact arithmetic)
template <int NX>
DEV void chol_packed(float (&H)[NX * (NX + 1) / 2], float (&id)[NX]) {
#pragma unroll
    for (int j = 0; j < NX; j++) {
        float d = H[tri(j, j)];
#pragma unroll
        for (int k = 0; k < j; k++) d = fmaf(-H[tri(j, k)], H[tri(j, k)], d);
        const float idj = rsq(fmaxf(d, 1e-30f));
        id[j] = idj;
#pragma unroll
        for (int i = j + 1; i < NX; i++) {
            float s = H[tri(i, j)];
#pragma unroll
            for (int k = 0; k < j; k++) s = fmaf(-H[tri(i, k)], H[tri(j, k)], s);
            H[tri(i, j)] = s * idj;
        }
    }
}
template <int NX>
DEV void solve_packed(const float (&H)[NX * (NX + 1) / 2], const float (&id)[NX], float (&x)[NX]) {   // x <- (L L')^-1 x
#pragma unroll
    for (int i = 0; i < NX; i++) {
        float s = x[i];
#pragma unroll
        for (int k = 0; k < i; k++) s = fmaf(-H[tri(i, k)], x[k], s);
        x[i] = s * id[i];
    }
#pragma unroll
    for (int i = NX - 1; i >= 0; i--) {
        float s = x[i];
#pragma unroll
        for (int k = i + 1; k < NX; k++) s = fmaf(-H[tri(k, i)], x[k], s);
        x[i] = s * id[i];
    }
}

// friction of the finger geoms against the floor (follower.xml:15: friction="1.5" + MuJoCo's default torsional 0.005 / rolling 0.0001; the finger class has priority 1)
constexpr float DEC_FLOOR = 3e-10f;   // a Newton decrement below DEC_FLOOR |x - a0|_M^2 that no longer shrinks is rounding (oracle: DEC_FLOOR)
constexpr float LS_NOISE = 1e-5f;   // relative rounding floor of phi'(al) in fp32 (oracle: LS_NOISE)
constexpr float MU_ROLL = 1e-4f;
constexpr float RR_FF = (MU_FINGER * MU_FINGER) / (MU_ROLL * MU_ROLL);   // regulariser scale of the rolling rows: mu_tan^2 / mu_roll^2

// rows and LDS placement of the arm-coupled slots in the Newton kernels: the finger slots 0-3 have six rows, the proxy slot four.  In LDS: the three LINEAR rows of
// each finger slot (rows 3 s + r), the three angular rows of each finger BODY (rows 12 + 3 sp + axis: B_a = L^-1 (z_j . e_a)_j -- the slot's torsion / rolling rows
// are d . B, and for a finger on the floor, frame (z, y, -x), they are B_z, B_y, -B_x themselves), the proxy slot's four rows (18 + r): 22 g rows, 33 KiB per wave
template <bool ROLL, bool NEWTON> constexpr int arm_rows_of(int s) { return (NEWTON && s < 4) ? 6 : as_rows<ROLL>(s); }
template <bool ROLL, int NC, bool BIG, bool NEWTON> constexpr int arm_row0_of(int s) { return NEWTON ? (s < 4 ? 3 * s : 18) : as_row0<ROLL, NC, BIG>(s); }
constexpr int NEWTON_G_ROWS = 22;
constexpr int NEWTON_BODY_ROW0 = 12;

// ================================================================================================
// The solve.  Everything a substep's set-up leaves behind is reached through NewtonCtx (references into the caller's registers / LDS).
// ================================================================================================
// Bodies: bit 0 the arm (y coordinates, 6), bit 1 cube 0, bit 2 cube 1 (linear + angular acceleration, 6 each).  newton_solve<.., MASK> minimises F over the
// bodies in MASK with every constraint that acts on them; the caller cuts the bodies of a wave into the connected components of its WAVE-UNIFORM coupling graph
// (an edge where some lane has a contact between two bodies) and solves them one after the other: independent problems, each as small as it can be -- most of
// the time the arm and the cube(s) do not touch, and a wave pays as many iterations as its slowest lane needs for THAT body.
// the scalars of LcrDev the solve reads (by value: the far entry below must not take the address of the kernel's argument block)
struct NewtonParams {
    float cube_iinv, cube_mass, inv_impratio, ls_tol, mu_c2, mu_ct2, mu_fc2, mu_fcr2, mu_fct2, newton_tol;
    int ls_iters, newton_iters;
};
DEV NewtonParams newton_params(const LcrDev &P) {
    return NewtonParams{P.cube_iinv, P.cube_mass, P.inv_impratio, P.ls_tol, P.mu_c2, P.mu_ct2, P.mu_fc2, P.mu_fcr2, P.mu_fct2, P.newton_tol, P.ls_iters, P.newton_iters
    };
}
template <int NC, int NRW, bool WALLS, int NCC>
struct NewtonCtx {
    const NewtonParams &P;
    const float *lds;   // g rows of the arm-coupled slots: row (row0[s] + r), float2 pairs [k][lane]
    int lane;
    const int (&row0)[NAS];
    ArmSlot<NRW> (&AS)[NAS];
    const bool (&slot_any)[NAS];
    bool link_on_cube;
    const int (&slot_cube)[3];
    FloorSlot (&FS)[NC][4];
    FloorSlot (&WS)[4];          // PushCubeLoop rails (lcr_kernels_loop.hip), pair coordinates
    const float (&wsg)[2];
    bool wall_any;
    float *ccl;                  // StackTwoCubes: cube<->cube records in LDS, field k of slot s at ccl[(s * CC_REC_NEWTON + k) * 64]
    const bool (&cc_act)[NCC];
    bool cc_any;
    f3 ccn, cct1, cct2;
    const f3 (&cp)[NC];
    const bool (&lim_act)[6];
    unsigned lim_wave;
    const float (&q)[6];
    const float (&qd)[6];
    const Chol6 &CL;
    float (&flim)[6];
    const float (&y0s)[6];
    int wave_its = 0;            // iterations this wave has executed in its solves of the substep (profiling aid, lcr_config.diagnostics = 2)
    int enable = 7;              // bit b clear: body b (0 arm, 1 cube 0, 2 cube 1) of this lane's env is solved elsewhere (lcr_newton_coop.h) -- a solve over bodies that are not all enabled
                                 // never counts the lane as live and leaves its accelerations as they are
};

template <int MASK> constexpr int nw_off(int body) {   // first compact index of a body of MASK
    int o = 0;
    for (int b = 0; b < body; b++) o += ((MASK >> b) & 1) ? 6 : 0;
    return o;
}
template <int MASK> constexpr int nw_dim() { return 6 * ((MASK & 1) + ((MASK >> 1) & 1) + ((MASK >> 2) & 1)); }

template <int NC, int NRW, bool WALLS, int NCC, int MASK>
DEV int newton_solve(NewtonCtx<NC, NRW, WALLS, NCC> &C, float (&y)[6], f3 (&ca)[NC], f3 (&cal)[NC]) {
    constexpr bool HAS_A = (MASK & 1) != 0;
    constexpr bool HAS_C[2] = {(MASK & 2) != 0, NC == 2 && (MASK & 4) != 0};
    constexpr int NX = nw_dim<MASK>(), NH = NX * (NX + 1) / 2;
    constexpr int OA = nw_off<MASK>(0), OC[2] = {nw_off<MASK>(1), nw_off<MASK>(2)};
    constexpr bool ARM_CUBE = HAS_A && (HAS_C[0] || HAS_C[1]);   // arm slots may carry a cube share
    constexpr bool HAS_CC = NC == 2 && HAS_C[0] && HAS_C[1];
    // residual rows: joint limits, arm slots (6 rows each, slot 4: 4), floor slots per cube, rails, cube<->cube
    constexpr int Z_LIM = 0, Z_ARM = 6, Z_FLOOR = 34, Z_WALL = Z_FLOOR + 16 * NC, Z_CC = Z_WALL + (WALLS ? 16 : 0), NZ = Z_CC + (NC == 2 ? 4 * NCC : 0);
    const NewtonParams &P = C.P;
    const int ln = C.lane;   // the lane's LDS column
    const float cm = P.cube_mass, ci = rcp(P.cube_iinv);
    auto mdiag = [&](int i) -> float { return (HAS_A && i < 6) ? 1.f : (((i - (HAS_A ? 6 : 0)) % 6) < 3 ? cm : ci); };
    const bool wave_lim = HAS_A && C.lim_wave != 0u;
    const bool wave_cube4 = ARM_CUBE && __any(C.AS[4].act && C.link_on_cube) != 0;
    bool fl_any[2] = {false, false};
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (!HAS_C[c]) continue;
        bool a = false;
#pragma unroll
        for (int s = 0; s < 4; s++) a = a || C.FS[c][s].act;
        fl_any[c] = __any(a) != 0;
    }
    float x[NX], x0[NX];
    if (HAS_A) {
#pragma unroll
        for (int j = 0; j < 6; j++) { x[OA + j] = y[j]; x0[OA + j] = C.y0s[j]; }
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (!HAS_C[c]) continue;
        x[OC[c] + 0] = ca[c].x; x[OC[c] + 1] = ca[c].y; x[OC[c] + 2] = ca[c].z; x[OC[c] + 3] = cal[c].x; x[OC[c] + 4] = cal[c].y; x[OC[c] + 5] = cal[c].z;
        x0[OC[c] + 0] = 0.f; x0[OC[c] + 1] = 0.f; x0[OC[c] + 2] = -GRAV; x0[OC[c] + 3] = 0.f; x0[OC[c] + 4] = 0.f; x0[OC[c] + 5] = 0.f;
    }
    // the tolerance scale 1 + |a0|_M^2 of the WHOLE system (oracle: one problem over all bodies), also when this call solves one component of it
    float scale = fmaf((float)NC * cm, GRAV * GRAV, 1.f);
#pragma unroll
    for (int j = 0; j < 6; j++) scale = fmaf(C.y0s[j], C.y0s[j], scale);
    // joint-limit rows: regulariser and reference acceleration once per solve
    float lim_aref[6], lim_iR[6];
#pragma unroll
    for (int j = 0; j < 6; j++) { lim_aref[j] = 0.f; lim_iR[j] = 0.f; }
    if (wave_lim) {
#pragma unroll
        for (int j = 0; j < 6; j++) {
            if (!((C.lim_wave >> j) & 1u)) continue;
            const bool lower = C.q[j] < JLO[j];
            const float pos = lower ? C.q[j] - JLO[j] : JHI[j] - C.q[j];
            const float imp = impedance(pos, D0_DEF, DW_DEF, 1.0f / W_DEF);
            lim_iR[j] = C.lim_act[j] ? rcp(fmaxf((1.f - imp) * rcp(imp) * INVW_DOF[j], 1e-15f)) : 0.f;
            lim_aref[j] = -B_DEF * (lower ? 1.f : -1.f) * C.qd[j] - K_DEF * imp * pos;
        }
    }
    auto lim_row = [&](int j, float (&g6)[6]) {
#pragma unroll
        for (int k = 0; k < 6; k++) g6[k] = k == j ? (C.q[j] < JLO[j] ? 1.f : -1.f) : 0.f;
        fsub(C.CL, g6);
    };
    // squared friction coefficients of an arm slot's rows (finger geoms: 1.5 / 0.005 / 1e-4; a finger on a cube: max rule; a link proxy: floor 1 and no torsion
    // row -- condim 3 --, on a cube the cube's coefficients)
    auto arm_m2 = [&](auto s_tag, float (&m2)[6]) {
        constexpr int s = decltype(s_tag)::value;
        const bool oncube = s < 2 || (s == 4 && C.link_on_cube);
        m2[0] = 1.f;
        m2[1] = m2[2] = s < 2 ? P.mu_fc2 : (s < 4 ? MU_FINGER * MU_FINGER : (oncube ? P.mu_c2 : 1.f));
        m2[3] = s < 2 ? P.mu_fct2 : (s < 4 ? MU_TORS * MU_TORS : (oncube ? P.mu_ct2 : 0.f));
        m2[4] = m2[5] = s < 2 ? P.mu_fcr2 : (s < 4 ? MU_ROLL * MU_ROLL : 0.f);
    };
    // which arm slots exist in this problem, and whether their rows carry a cube share here
    auto arm_here = [&](int s) -> bool { return HAS_A && C.slot_any[s]; };
    auto arm_cube_part = [&](int s) -> bool { return ARM_CUBE && (s < 2 || (s == 4 && wave_cube4)); };
    // the dense row r of arm slot s in the compact coordinates: g row from LDS, the cube's share in closed form
    auto arm_row = [&](auto s_tag, int r, bool cube_part, float (&row)[NX]) {
        constexpr int s = decltype(s_tag)::value;
        const ArmSlot<NRW> &T = C.AS[s];
#pragma unroll
        for (int i = 0; i < NX; i++) row[i] = 0.f;
        auto ld = [&](int lrow, float (&o)[6]) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float2v gp = *reinterpret_cast<const float2v *>(&C.lds[lrow * LDS_ROW + k * 128 + ln * 2]);
                o[2 * k] = gp.x; o[2 * k + 1] = gp.y;
            }
        };
        float g6[6];
        if (s < 4 && r >= 3) {   // torsion / rolling rows of a finger contact: d . B of the finger body (rows NEWTON_BODY_ROW0 + 3 sp + axis)
            constexpr int b0 = NEWTON_BODY_ROW0 + 3 * (s & 1);
            if (s >= 2 && !WALLS) {   // on the floor the frame is (z, y, -x): the body rows themselves (PushCubeLoop: a rail's side face has another frame)
                ld(b0 + (r == 3 ? 2 : (r == 4 ? 1 : 0)), g6);
                if (r == 5) {
#pragma unroll
                    for (int k = 0; k < 6; k++) g6[k] = -g6[k];
                }
            } else {
                const f3 dd = r == 3 ? T.n : (r == 4 ? T.t1 : T.t2);
                float bx[6], by[6], bz[6];
                ld(b0, bx); ld(b0 + 1, by); ld(b0 + 2, bz);
#pragma unroll
                for (int k = 0; k < 6; k++) g6[k] = fmaf(dd.x, bx[k], fmaf(dd.y, by[k], dd.z * bz[k]));
            }
        } else ld(C.row0[s] + r, g6);
#pragma unroll
        for (int k = 0; k < 6; k++) row[OA + k] = g6[k];
        if (ARM_CUBE && cube_part) {   // the cube's contact point moves with ca + cal x rc (rows 0-2); rows 3-5 see cal
            const f3 d = r == 0 ? T.n : (r == 1 ? T.t1 : (r == 2 ? T.t2 : (r == 3 ? T.n : (r == 4 ? T.t1 : T.t2))));
            const float on = (s < 2 || C.link_on_cube) ? -1.f : 0.f;
            const f3 lin = r < 3 ? on * d : mk(0.f, 0.f, 0.f), ang = r < 3 ? on * cross(T.rc, d) : on * d;
            const int which = NC == 2 ? C.slot_cube[s == 4 ? 2 : (s & 1)] : 0;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if (!HAS_C[c]) continue;
                const float m = (NC == 1 || which == c) ? 1.f : 0.f;
                row[OC[c] + 0] = m * lin.x; row[OC[c] + 1] = m * lin.y; row[OC[c] + 2] = m * lin.z;
                row[OC[c] + 3] = m * ang.x; row[OC[c] + 4] = m * ang.y; row[OC[c] + 5] = m * ang.z;
            }
        }
    };
    // a contact between a cube and the world: frame (n, t1, t2), lever r from the cube centre; rows [d ; r x d], torsion [0 ; n]
    struct CubeFrame { f3 n, t1, t2, r; };
    auto floor_frame = [&](int c, int s) -> CubeFrame { return CubeFrame{mk(0.f, 0.f, 1.f), mk(0.f, 1.f, 0.f), mk(-1.f, 0.f, 0.f), C.FS[c][s].r}; };
    auto wall_frame = [&](int s) -> CubeFrame {   // rails: pair coordinates (a, b, c) = (x, y, z) for the x pair, (y, z, x) for the y pair; n = sg a, t1 = b, t2 = sg c
        const float sg = C.wsg[s >> 1];
        const f3 r = C.WS[s].r;
        if ((s >> 1) == 0) return CubeFrame{mk(sg, 0.f, 0.f), mk(0.f, 1.f, 0.f), mk(0.f, 0.f, sg), r};
        return CubeFrame{mk(0.f, sg, 0.f), mk(0.f, 0.f, 1.f), mk(sg, 0.f, 0.f), mk(r.z, r.x, r.y)};
    };
    auto cube_rows = [&](const CubeFrame &Fm, int c, float sgn, float (&row)[4][NX], bool clear) {
        if (clear) {
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int i = 0; i < NX; i++) row[q][i] = 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const f3 d = q == 0 ? Fm.n : (q == 1 ? Fm.t1 : (q == 2 ? Fm.t2 : Fm.n));
            const f3 lin = q < 3 ? sgn * d : mk(0.f, 0.f, 0.f), ang = q < 3 ? sgn * cross(Fm.r, d) : sgn * d;
            const int o = c == 0 ? OC[0] : OC[1];
            row[q][o + 0] = lin.x; row[q][o + 1] = lin.y; row[q][o + 2] = lin.z; row[q][o + 3] = ang.x; row[q][o + 4] = ang.y; row[q][o + 5] = ang.z;
        }
    };
    auto cc_pos = [&](int s) -> f3 { return mk(C.ccl[(size_t)(s * CC_REC_NEWTON + 0) * 64], C.ccl[(size_t)(s * CC_REC_NEWTON + 1) * 64], C.ccl[(size_t)(s * CC_REC_NEWTON + 2) * 64]); };
    auto cc_rows = [&](int s, float (&row)[4][NX]) {
        const f3 pos = cc_pos(s);
        cube_rows(CubeFrame{C.ccn, C.cct1, C.cct2, pos - C.cp[NC - 1]}, NC - 1, 1.f, row, true);
        cube_rows(CubeFrame{C.ccn, C.cct1, C.cct2, pos - C.cp[0]}, 0, -1.f, row, false);
    };

    float zs[NZ], jd[NZ];
#pragma unroll
    for (int i = 0; i < NZ; i++) { zs[i] = 0.f; jd[i] = 0.f; }
    // out[row] = J_row . v (- aref with SUB): the residuals at the start (v = x) and the directional derivatives of a Newton step (v = dx)
    auto dots = [&](auto sub_tag, const float (&v)[NX], float (&out)[NZ]) {
        constexpr bool SUB = decltype(sub_tag)::value;
        if (wave_lim) {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                if (!((C.lim_wave >> j) & 1u)) continue;
                float g6[6];
                lim_row(j, g6);
                float a = SUB ? -lim_aref[j] : 0.f;
#pragma unroll
                for (int k = 0; k < 6; k++) a = fmaf(g6[k], v[OA + k], a);
                out[Z_LIM + j] = a;
            }
        }
        auto arm = [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            if (!arm_here(s)) return;
            constexpr int NR = s < 4 ? 6 : 4;
            const bool cube_part = arm_cube_part(s);
#pragma unroll
            for (int r = 0; r < NR; r++) {
                float row[NX];
                arm_row(s_tag, r, cube_part, row);
                float a = SUB ? -C.AS[s].aref[r] : 0.f;
#pragma unroll
                for (int i = 0; i < NX; i++) { if ((HAS_A && i >= OA && i < OA + 6) || cube_part) a = fmaf(row[i], v[i], a); }
                out[Z_ARM + 6 * s + r] = a;
            }
        };
        arm(std::integral_constant<int, 0>{}); arm(std::integral_constant<int, 1>{}); arm(std::integral_constant<int, 2>{});
        arm(std::integral_constant<int, 3>{}); arm(std::integral_constant<int, 4>{});
        auto cube_con = [&](const CubeFrame &Fm, int c, const float (&aref)[4], int zoff) {
            float row[4][NX];
            cube_rows(Fm, c, 1.f, row, true);
            const int o = c == 0 ? OC[0] : OC[1];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float a = SUB ? -aref[q] : 0.f;
#pragma unroll
                for (int i = 0; i < 6; i++) a = fmaf(row[q][o + i], v[o + i], a);
                out[zoff + q] = a;
            }
        };
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (!HAS_C[c] || !fl_any[c]) continue;
#pragma unroll
            for (int s = 0; s < 4; s++) cube_con(floor_frame(c, s), c, C.FS[c][s].aref, Z_FLOOR + 16 * c + 4 * s);
        }
        if constexpr (WALLS) {
            if (HAS_C[0] && C.wall_any) {
#pragma unroll
                for (int s = 0; s < 4; s++) cube_con(wall_frame(s), 0, C.WS[s].aref, Z_WALL + 4 * s);
            }
        }
        if constexpr (HAS_CC) {
            if (C.cc_any) {
#pragma unroll
                for (int s = 0; s < NCC; s++) {
                    float row[4][NX];
                    cc_rows(s, row);
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        float a = SUB ? -C.ccl[(size_t)(s * CC_REC_NEWTON + 7 + q) * 64] : 0.f;
#pragma unroll
                        for (int i = 0; i < NX; i++) { if (i >= OC[0]) a = fmaf(row[q][i], v[i], a); }
                        out[Z_CC + 4 * s + q] = a;
                    }
                }
            }
        }
    };
    // sum over the constraints of f(zs + al jd) . jd and of jd' W(zs + al jd) jd: the constraint parts of phi'(al) and phi''(al) -- registers only (cube<->cube: their
    // regulariser from LDS).  Per block  jd'W jd = av (jd_n - u)^2 - gam u^2 + kap sum_t m2_t jd_t^2,  u = sum_t c_t jd_t  (the same av / gam / kap / c as in h_block)
    struct LsVal { float f, h; };
    auto ls_eval = [&](float al) -> LsVal {
        float acc = 0.f, hac = 0.f;
        if (wave_lim) {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                if (!((C.lim_wave >> j) & 1u)) continue;
                const float z = fmaf(al, jd[Z_LIM + j], zs[Z_LIM + j]);
                const float wl = z < 0.f ? lim_iR[j] : 0.f;
                acc = fmaf(-z * wl, jd[Z_LIM + j], acc);
                hac = fmaf(wl * jd[Z_LIM + j], jd[Z_LIM + j], hac);
            }
        }
        auto curv = [&](auto nr_tag, const auto &B, const auto &m2, int zoff) {
            constexpr int NR = decltype(nr_tag)::value;
            float u = 0.f, tt = 0.f;
#pragma unroll
            for (int r = 1; r < NR; r++) { u = fmaf(B.c[r], jd[zoff + r], u); tt = fmaf(m2[r] * jd[zoff + r], jd[zoff + r], tt); }
            const float sn = jd[zoff] - u;
            hac = fmaf(B.av * sn, sn, fmaf(-B.gam * u, u, fmaf(B.kap, tt, hac)));
        };
        auto arm = [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            if (!arm_here(s)) return;
            constexpr int NR = s < 4 ? 6 : 4;
            float m2a[6], m2[NR], z[NR];
            arm_m2(s_tag, m2a);
#pragma unroll
            for (int r = 0; r < NR; r++) { m2[r] = m2a[r]; z[r] = fmaf(al, jd[Z_ARM + 6 * s + r], zs[Z_ARM + 6 * s + r]); }
            BlkEval<NR> B;
            blk_eval<NR>(z, C.AS[s].Rn, C.AS[s].Rn * P.inv_impratio * m2[1], m2, C.AS[s].act, B);
#pragma unroll
            for (int r = 0; r < NR; r++) acc = fmaf(B.f[r], jd[Z_ARM + 6 * s + r], acc);
            curv(std::integral_constant<int, NR>{}, B, m2, Z_ARM + 6 * s);
        };
        arm(std::integral_constant<int, 0>{}); arm(std::integral_constant<int, 1>{}); arm(std::integral_constant<int, 2>{});
        arm(std::integral_constant<int, 3>{}); arm(std::integral_constant<int, 4>{});
        auto cube_con = [&](float Rn, bool act, int zoff) {
            const float m2[4] = {1.f, P.mu_c2, P.mu_c2, P.mu_ct2};
            float z[4];
#pragma unroll
            for (int q = 0; q < 4; q++) z[q] = fmaf(al, jd[zoff + q], zs[zoff + q]);
            BlkEval<4> B;
            blk_eval<4>(z, Rn, Rn * P.inv_impratio * P.mu_c2, m2, act, B);
#pragma unroll
            for (int q = 0; q < 4; q++) acc = fmaf(B.f[q], jd[zoff + q], acc);
            curv(std::integral_constant<int, 4>{}, B, m2, zoff);
        };
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (!HAS_C[c] || !fl_any[c]) continue;
#pragma unroll
            for (int s = 0; s < 4; s++) cube_con(C.FS[c][s].Rn, C.FS[c][s].act, Z_FLOOR + 16 * c + 4 * s);
        }
        if constexpr (WALLS) {
            if (HAS_C[0] && C.wall_any) {
#pragma unroll
                for (int s = 0; s < 4; s++) cube_con(C.WS[s].Rn, C.WS[s].act, Z_WALL + 4 * s);
            }
        }
        if constexpr (HAS_CC) {
            if (C.cc_any) {
#pragma unroll
                for (int s = 0; s < NCC; s++) cube_con(C.ccl[(size_t)(s * CC_REC_NEWTON + CC_RN_NEWTON) * 64], C.cc_act[s], Z_CC + 4 * s);
            }
        }
        return LsVal{acc, hac};
    };
    // line-search fall-back (see the loop below): among the contact blocks whose N(al) = |w_t(zs + al jd)| has its minimiser inside the open bracket (lo, hi) AND that are
    // in their sticking zone there (N_min Rn <= -w_n Rt), the minimiser closest to `sec`; none: `sec`
    auto kink_cand = [&](float lo, float hi, float sec) -> float {
        float best = sec, bestd = -1.f;
        auto blk = [&](auto nr_tag, const auto &m2, float Rn, float Rt, bool act, int zoff) {
            constexpr int NR = decltype(nr_tag)::value;
            float qa = 0.f, qb = 0.f, qc = 0.f;
#pragma unroll
            for (int r = 1; r < NR; r++) {
                const float mj = m2[r] * jd[zoff + r];
                qa = fmaf(mj, jd[zoff + r], qa); qb = fmaf(mj, zs[zoff + r], qb); qc = fmaf(m2[r] * zs[zoff + r], zs[zoff + r], qc);
            }
            const float iqa = rcp(fmaxf(qa, 1e-30f));
            const float am = -qb * iqa;
            const float n2 = fmaxf(fmaf(-qb * qb, iqa, qc), 0.f), wn = fmaf(am, jd[zoff], zs[zoff]);
            const bool ok = act && qa > 0.f && am > lo && am < hi && wn < 0.f && !(n2 * Rn * Rn > wn * wn * Rt * Rt);
            const float dist = fabsf(am - sec);
            const bool take = ok && (bestd < 0.f || dist < bestd);
            best = take ? am : best;
            bestd = take ? dist : bestd;
        };
        auto arm = [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            if (!arm_here(s)) return;
            constexpr int NR = s < 4 ? 6 : 4;
            float m2a[6], m2[NR];
            arm_m2(s_tag, m2a);
#pragma unroll
            for (int r = 0; r < NR; r++) m2[r] = m2a[r];
            blk(std::integral_constant<int, NR>{}, m2, C.AS[s].Rn, C.AS[s].Rn * P.inv_impratio * m2[1], C.AS[s].act, Z_ARM + 6 * s);
        };
        arm(std::integral_constant<int, 0>{}); arm(std::integral_constant<int, 1>{}); arm(std::integral_constant<int, 2>{});
        arm(std::integral_constant<int, 3>{}); arm(std::integral_constant<int, 4>{});
        const float m2c[4] = {1.f, P.mu_c2, P.mu_c2, P.mu_ct2};
        auto cube_con = [&](float Rn, bool act, int zoff) { blk(std::integral_constant<int, 4>{}, m2c, Rn, Rn * P.inv_impratio * P.mu_c2, act, zoff); };
#pragma unroll
        for (int c = 0; c < NC; c++) {
            if (!HAS_C[c] || !fl_any[c]) continue;
#pragma unroll
            for (int s = 0; s < 4; s++) cube_con(C.FS[c][s].Rn, C.FS[c][s].act, Z_FLOOR + 16 * c + 4 * s);
        }
        if constexpr (WALLS) {
            if (HAS_C[0] && C.wall_any) {
#pragma unroll
                for (int s = 0; s < 4; s++) cube_con(C.WS[s].Rn, C.WS[s].act, Z_WALL + 4 * s);
            }
        }
        if constexpr (HAS_CC) {
            if (C.cc_any) {
#pragma unroll
                for (int s = 0; s < NCC; s++) cube_con(C.ccl[(size_t)(s * CC_REC_NEWTON + CC_RN_NEWTON) * 64], C.cc_act[s], Z_CC + 4 * s);
            }
        }
        return best;
    };
    // gradient and Hessian of F at x (residuals zs) -- or, with OUT, the forces at x into the slot records
    auto assemble = [&](auto out_tag, float (&g)[NX], float (&Hm)[NH]) {
        constexpr bool OUT = decltype(out_tag)::value;
        if (!OUT) {
#pragma unroll
            for (int i = 0; i < NX; i++) g[i] = mdiag(i) * (x[i] - x0[i]);
#pragma unroll
            for (int i = 0; i < NH; i++) Hm[i] = 0.f;
#pragma unroll
            for (int i = 0; i < NX; i++) Hm[tri(i, i)] = mdiag(i);
        }
        if (wave_lim) {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                if (!((C.lim_wave >> j) & 1u)) continue;
                const float z = zs[Z_LIM + j];
                const float wl = z < 0.f ? lim_iR[j] : 0.f, f = -z * wl;
                if (OUT) { C.flim[j] = f; continue; }
                float gl[NX], g6[6];
#pragma unroll
                for (int k = 0; k < NX; k++) gl[k] = 0.f;
                lim_row(j, g6);
#pragma unroll
                for (int k = 0; k < 6; k++) { gl[OA + k] = g6[k]; g[OA + k] = fmaf(-f, g6[k], g[OA + k]); }
                h_rank1<OA, OA + 6, NX>(Hm, gl, wl);
            }
        }
        auto arm = [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            if (!arm_here(s)) return;
            constexpr int NR = s < 4 ? 6 : 4;
            ArmSlot<NRW> &T = C.AS[s];
            const bool cube_part = arm_cube_part(s);
            float m2a[6], m2[NR], z[NR];
            arm_m2(s_tag, m2a);
#pragma unroll
            for (int r = 0; r < NR; r++) { m2[r] = m2a[r]; z[r] = zs[Z_ARM + 6 * s + r]; }
            BlkEval<NR> B;
            blk_eval<NR>(z, T.Rn, T.Rn * P.inv_impratio * m2[1], m2, T.act, B);
            if (OUT) {
#pragma unroll
                for (int r = 0; r < NR; r++) T.f[r] = B.f[r];
                return;
            }
            // rows streamed one at a time (a six-row slot of a coupled problem would otherwise hold 6 x 12 row entries at once -- the register peak of the kernel):
            // per row the gradient share and its kap m2 row row' term; the normal row and w = sum_t c_t row_t are kept for the two terms av v v' - gam w w', v = row_n - w
            auto stream = [&](auto lo_tag, auto hi_tag) {
                constexpr int LO = decltype(lo_tag)::value, HI = decltype(hi_tag)::value;
                float row0[NX], wv[NX];
#pragma unroll
                for (int i = 0; i < NX; i++) { row0[i] = 0.f; wv[i] = 0.f; }
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    float row[NX];
                    arm_row(s_tag, r, cube_part, row);
#pragma unroll
                    for (int i = LO; i < HI; i++) g[i] = fmaf(-B.f[r], row[i], g[i]);
                    if (r == 0) {
#pragma unroll
                        for (int i = LO; i < HI; i++) row0[i] = row[i];
                    } else {
#pragma unroll
                        for (int i = LO; i < HI; i++) wv[i] = fmaf(B.c[r], row[i], wv[i]);
                        h_rank1<LO, HI, NX>(Hm, row, B.kap * m2[r]);
                    }
                }
#pragma unroll
                for (int i = LO; i < HI; i++) row0[i] -= wv[i];
                h_rank1<LO, HI, NX>(Hm, row0, B.av);
                h_rank1<LO, HI, NX>(Hm, wv, -B.gam);
            };
            if (ARM_CUBE && cube_part) stream(std::integral_constant<int, 0>{}, std::integral_constant<int, NX>{});
            else stream(std::integral_constant<int, OA>{}, std::integral_constant<int, OA + 6>{});
        };
        arm(std::integral_constant<int, 0>{}); arm(std::integral_constant<int, 1>{}); arm(std::integral_constant<int, 2>{});
        arm(std::integral_constant<int, 3>{}); arm(std::integral_constant<int, 4>{});
        auto cube_con = [&](auto c_tag, const CubeFrame &Fm, FloorSlot &T, bool act, int zoff) {
            constexpr int c = decltype(c_tag)::value;
            const float m2[4] = {1.f, P.mu_c2, P.mu_c2, P.mu_ct2};
            float z[4];
#pragma unroll
            for (int q = 0; q < 4; q++) z[q] = zs[zoff + q];
            BlkEval<4> B;
            blk_eval<4>(z, T.Rn, T.Rn * P.inv_impratio * P.mu_c2, m2, act, B);
            if (OUT) {
#pragma unroll
                for (int q = 0; q < 4; q++) T.f[q] = B.f[q];
                return;
            }
            float row[4][NX];
            cube_rows(Fm, c, 1.f, row, true);
            constexpr int o = OC[c];
#pragma unroll
            for (int i = 0; i < 6; i++) {
                float a = g[o + i];
#pragma unroll
                for (int q = 0; q < 4; q++) a = fmaf(-B.f[q], row[q][o + i], a);
                g[o + i] = a;
            }
            h_block<o, o + 6, NX, 4>(Hm, row, B, m2);
        };
        if constexpr (HAS_C[0]) {
            if (fl_any[0] || OUT) {
#pragma unroll
                for (int s = 0; s < 4; s++) cube_con(std::integral_constant<int, 0>{}, floor_frame(0, s), C.FS[0][s], C.FS[0][s].act && fl_any[0], Z_FLOOR + 4 * s);
            }
            if constexpr (WALLS) {
                if (C.wall_any || OUT) {
#pragma unroll
                    for (int s = 0; s < 4; s++) cube_con(std::integral_constant<int, 0>{}, wall_frame(s), C.WS[s], C.WS[s].act && C.wall_any, Z_WALL + 4 * s);
                }
            }
        }
        if constexpr (HAS_C[1]) {
            if (fl_any[1] || OUT) {
#pragma unroll
                for (int s = 0; s < 4; s++) cube_con(std::integral_constant<int, 1>{}, floor_frame(1, s), C.FS[NC - 1][s], C.FS[NC - 1][s].act && fl_any[1], Z_FLOOR + 16 + 4 * s);
            }
        }
        if constexpr (HAS_CC) {
            if (C.cc_any) {
#pragma unroll
                for (int s = 0; s < NCC; s++) {
                    const float m2[4] = {1.f, P.mu_c2, P.mu_c2, P.mu_ct2};
                    const float Rn = C.ccl[(size_t)(s * CC_REC_NEWTON + CC_RN_NEWTON) * 64];
                    float z[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) z[q] = zs[Z_CC + 4 * s + q];
                    BlkEval<4> B;
                    blk_eval<4>(z, Rn, Rn * P.inv_impratio * P.mu_c2, m2, C.cc_act[s], B);
                    if (OUT) {
#pragma unroll
                        for (int q = 0; q < 4; q++) C.ccl[(size_t)(s * CC_REC_NEWTON + 3 + q) * 64] = B.f[q];
                        continue;
                    }
                    float row[4][NX];
                    cc_rows(s, row);
#pragma unroll
                    for (int i = OC[0]; i < NX; i++) {
                        float a = g[i];
#pragma unroll
                        for (int q = 0; q < 4; q++) a = fmaf(-B.f[q], row[q][i], a);
                        g[i] = a;
                    }
                    h_block<OC[0], NX, NX, 4>(Hm, row, B, m2);
                }
            }
        }
    };

    dots(std::true_type{}, x, zs);
    const float tol2 = P.newton_tol * P.newton_tol * scale;
    int lane_its = 0;   // iterations in which THIS env still moved (what the oracle counts per env)
    float dprev = 3.0e38f;
    for (int it = 0; it < P.newton_iters; it++) {
        float dx[NX], d0 = 0.f;
        {
            float Hm[NH], g[NX], hid[NX];
            assemble(std::false_type{}, g, Hm);
            chol_packed<NX>(Hm, hid);
#pragma unroll
            for (int i = 0; i < NX; i++) dx[i] = -g[i];
            solve_packed<NX>(Hm, hid, dx);
#pragma unroll
            for (int i = 0; i < NX; i++) d0 = fmaf(g[i], dx[i], d0);
        }
        // Newton decrement above the tolerance: this lane still moves.  Second exit (oracle: DEC_FLOOR): the gradient M (x - a0) - J'f carries the cancellation of stiff
        // rows (f = -z / R), its fp32 rounding leaves a decrement that no iteration removes and that can lie far above newton_tol^2 (1 + |a0|_M^2) -- recognised as a
        // decrement at rounding level relative to the problem (<= 3e-10 |x - a0|_M^2) that has stopped shrinking (not below a quarter of the previous iteration's)
        float dist2 = 0.f;
#pragma unroll
        for (int i = 0; i < NX; i++) dist2 = fmaf(mdiag(i) * (x[i] - x0[i]), x[i] - x0[i], dist2);
        const bool live = (C.enable & MASK) == MASK && -d0 > tol2 && !(-d0 <= DEC_FLOOR * dist2 && -d0 >= 0.25f * dprev);
        dprev = -d0;
        if (!__any(live)) break;
        lane_its += live ? 1 : 0;
        C.wave_its++;
        dots(std::false_type{}, dx, jd);
        // line search on phi'(al) = q0 + al q1 - sum f(zs + al jd) . jd (monotone increasing), with phi''(al) = q1 + sum jd'W jd from the same pass: first the full step
        // (exact while no contact changes zone); no bracket yet: the Newton step on phi' from the last point; bracket [lo, hi]: the Newton candidate from the end with the
        // smaller |phi'| (from the stale end after two updates of the same end in a row), a candidate outside the open bracket being none; both ends pointing outside:
        // a steep piece hides between two flat ones -- a sliding contact that comes to rest within the step --, try the minimiser of that block's N(al) (kink_cand),
        // else the Illinois secant point, else the midpoint.  Budget spent without meeting ls_tol: the lower end of the bracket (F decreases on [0, root]).
        // (oracle: newton_product)
        float q1 = 0.f;
#pragma unroll
        for (int i = 0; i < NX; i++) q1 = fmaf(mdiag(i) * dx[i], dx[i], q1);
        float q0 = 0.f;   // [M (x - x0)] . dx  (M is diagonal in these variables; the oracle's mpart at al = 0)
#pragma unroll
        for (int i = 0; i < NX; i++) q0 = fmaf(mdiag(i) * (x[i] - x0[i]), dx[i], q0);
        float al = 1.f, lo_a = 0.f, hi_a = -1.f, dlo = d0, dhi = 0.f, hlo = -d0, hhi = 0.f, dlo_m = d0, dhi_m = 0.f;
        int last_side = 0, same = 0;
        bool done = !live, conv = !live;
        for (int ls = 0; ls < P.ls_iters; ls++) {
            const LsVal e = ls_eval(al);
            const float mpart = fmaf(al, q1, q0);   // [M (x + al dx - x0)] . dx
            const float dphi = mpart - e.f, ddphi = q1 + e.h;
            float an = al, sec = al;
            bool need = false;
            if (!done) {
                // (the two partial sums cancel at the root: their fp32 rounding, not ls_tol, bounds what the search can resolve near convergence)
                const bool fin = fabsf(dphi) <= fmaf(P.ls_tol, fabsf(d0), LS_NOISE * (fabsf(mpart) + fabsf(e.f)));
                const int side = dphi < 0.f ? -1 : 1;
                if (dphi < 0.f) { if (last_side < 0) dhi_m *= 0.5f; lo_a = al; dlo = dphi; hlo = ddphi; dlo_m = dphi; }
                else { if (last_side > 0) dlo_m *= 0.5f; hi_a = al; dhi = dphi; hhi = ddphi; dhi_m = dphi; }
                same = side == last_side ? same + 1 : 0;
                last_side = side;
                if (hi_a < 0.f) an = lo_a - dlo * rcp(hlo);
                else {
                    const float cl = lo_a - dlo * rcp(hlo), ch = hi_a - dhi * rcp(hhi);
                    const float mg = 1e-4f * (hi_a - lo_a), blo = lo_a + mg, bhi = hi_a - mg;   // (inside by a margin: a candidate that repeats an end teaches nothing)
                    const bool vl = cl > blo && cl < bhi, vh = ch > blo && ch < bhi;
                    bool from_lo = fabsf(dlo) <= fabsf(dhi);
                    if (same >= 2) from_lo = side > 0;
                    sec = lo_a - dlo_m * (hi_a - lo_a) * rcp(dhi_m - dlo_m);
                    if (!(sec > lo_a && sec < hi_a)) sec = 0.5f * (lo_a + hi_a);
                    an = from_lo ? (vl ? cl : ch) : (vh ? ch : cl);
                    need = !fin && !vl && !vh;
                }
                conv = fin;
                done = fin;
            }
            if (__any(need)) {   // (rare: the block minimisers are computed only when some lane asks for them)
                const float mg = 1e-4f * (hi_a - lo_a);
                const float kc = kink_cand(lo_a + mg, hi_a - mg, sec);
                an = need ? kc : an;
            }
            al = done ? al : an;
            if (__all(done)) break;
        }
        if (!conv) al = lo_a > 0.f ? lo_a : hi_a;
        const float step = live ? al : 0.f;
#pragma unroll
        for (int i = 0; i < NX; i++) x[i] = fmaf(step, dx[i], x[i]);
#pragma unroll
        for (int i = 0; i < NZ; i++) zs[i] = fmaf(step, jd[i], zs[i]);
    }
    {   // the forces at the solution: carried to the next substep / control step
        float Hd[NH], gd[NX];
        assemble(std::true_type{}, gd, Hd);
    }
    // the integration uses the accelerations themselves (M (x - a0) = J'f at the optimum)
    if (HAS_A) {
#pragma unroll
        for (int j = 0; j < 6; j++) y[j] = x[OA + j];
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        if (!HAS_C[c]) continue;
        ca[c] = mk(x[OC[c] + 0], x[OC[c] + 1], x[OC[c] + 2]); cal[c] = mk(x[OC[c] + 3], x[OC[c] + 4], x[OC[c] + 5]);
    }
    return lane_its;
}


}  // namespace
