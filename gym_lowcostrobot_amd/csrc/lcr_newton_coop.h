// lcr_newton_coop.h -- the COUPLED arm + cube problem of ONE env solved by the whole wave (one-cube Newton kernels).
//
// Why.  One lane per env makes a wave pay for its slowest lane, and the coupled problem (12 unknowns, 78-entry Hessian, 50 constraint rows) is 2.5 x the cost of the
// two 6-dimensional ones per iteration with every lane of the wave dragged through it: round 5's launch time was the one wave in which ONE lane had a finger on its
// cube for all 20 substeps (7 of 12-dim iterations per substep at 45 k cycles each, profiles/r05_newton_phases.txt).  While few lanes of a wave are coupled
// (LcrDev::coop_max, default 3) the others keep their two small solves (newton_solve<.., 1> / <.., 2> with the coupled lanes disabled) and each coupled env -- the
// "patient", lane L -- is solved here by all 64 lanes: the same algorithm (lcr_newton.h; oracle: newton_product), laid out the other way round:
//
//   lane b < NB owns ONE constraint block of the patient: b = 0..4 the arm slots (finger<->cube 0 1, finger<->floor 2 3, link proxy 4), then the floor slots of
//   the cube(s), the eight cube<->cube slots (Stack), last the six joint limits (a limit is a block of one row: blk_eval with no friction rows is max(0, -z / R)):
//   15 blocks with one cube, 27 with two.  It has its block's rows as dense vectors in the NX = 12 / 18 unknowns, evaluates its cone zone, its share of the gradient
//   J'f and of the Hessian J'WJ (the SIMT code of ONE slot: blk_eval, h_block), and its share of phi'(al) / phi''(al) in the line search;
//   the shares are summed through LDS in chunks of 64 numbers ([block][64] written, column e summed by lane e, the totals read back by every lane) or by DPP (the
//   two sums of a line-search evaluation);
//   every lane then holds the whole NX x NX system and factorises / solves it redundantly (chol_packed<NX>), so the iteration logic -- tolerances, the bracketing
//   line search, the exits -- is the SIMT code with wave-uniform values.
//   Two cubes: a patient is an env with ANY coupling (arm on a cube, cube on cube) and all three bodies are solved as one 18-dimensional problem; the rows are
//   rebuilt where they are used (twice per iteration) instead of being kept next to the 171-entry Hessian.
//
// The patient's data reach the other lanes through an LDS staging area it writes itself (block records, x, a0, the Cholesky factor of M for the limit rows); its
// results (accelerations, the force of every row) go back the same way.  ~2 k instructions per iteration and patient instead of ~7 k for all lanes.
#pragma once
#include <type_traits>

#include "lcr_newton.h"

namespace {

template <int NC, bool WALLS = false> constexpr int coop_nb() { return 5 + 4 * NC + (NC == 2 ? 8 : 0) + (WALLS ? 4 : 0) + 6; }   // blocks of one patient: 15 / 27 (PushCubeLoop, its four rails: 19)
template <int NC> constexpr int coop_nx() { return 6 + 6 * NC; }
constexpr int COOP_REC = 32;                     // floats of a block record in the staging area
constexpr int COOP_CH = 64;                      // numbers per chunk of the LDS reduction
// LDS floats of the cooperative solve: [NB][64] shares of one chunk + the totals (gradient + packed Hessian, padded to whole chunks); the staging area overlays the shares
template <int NC> constexpr int coop_red() { return coop_nx<NC>() + coop_nx<NC>() * (coop_nx<NC>() + 1) / 2; }          // 90 / 189
template <int NC> constexpr int coop_chunks() { return (coop_red<NC>() + COOP_CH - 1) / COOP_CH; }                        // 2 / 3
template <int NC, bool WALLS = false> constexpr int coop_stage_floats() { return coop_nb<NC, WALLS>() * COOP_REC + 21 + coop_nx<NC>() + 6 + 2; }
template <int NC, bool WALLS = false> constexpr int coop_floats() {
    return (coop_nb<NC, WALLS>() * COOP_CH > coop_stage_floats<NC, WALLS>() ? coop_nb<NC, WALLS>() * COOP_CH : coop_stage_floats<NC, WALLS>()) + coop_chunks<NC>() * COOP_CH;
}

// LDS stores complete before the loads that follow.  Needed, not belt and braces: the column sums of a chunk are ds_read2st64 loads issued right behind the sixteen
// ds_write_b128 of the shares, and the compiler lets them return into the registers those stores send (profiles/r06_lds_store_hazard.txt: lanes 15-25 of the second
// chunk read wrong sums on the MI355X until the stores were waited for -- the solve then diverged by radians)
DEV void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// sum over the 64 lanes, the same value in every lane (DPP row reductions, then the two row broadcasts of GFX9; the total arrives in lane 63)
DEV float wave_sum(float v) {
    auto dpp = [](float x, auto ctrl_tag, auto rmask_tag) -> float {
        constexpr int ctrl = decltype(ctrl_tag)::value, rmask = decltype(rmask_tag)::value;
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, rmask, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xb1>{}, std::integral_constant<int, 0xf>{});    // quad_perm [1,0,3,2]
    v += dpp(v, std::integral_constant<int, 0x4e>{}, std::integral_constant<int, 0xf>{});    // quad_perm [2,3,0,1]
    v += dpp(v, std::integral_constant<int, 0x124>{}, std::integral_constant<int, 0xf>{});   // row_ror:4
    v += dpp(v, std::integral_constant<int, 0x128>{}, std::integral_constant<int, 0xf>{});   // row_ror:8   (every lane of a row holds the row's sum)
    v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});   // row_bcast:15 -> rows 1, 3
    v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});   // row_bcast:31 -> rows 2, 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// H += w v v' restricted to the packed entries [E0, E1)  (the Hessian share of a block is produced chunk by chunk)
template <int NX, int E0, int E1>
DEV void h_rank1_part(float (&H)[E1 - E0], const float (&v)[NX], float w) {
#pragma unroll
    for (int i = 0; i < NX; i++) {
        const float t = w * v[i];
#pragma unroll
        for (int j = 0; j <= i; j++) {
            if (tri(i, j) >= E0 && tri(i, j) < E1) H[tri(i, j) - E0] = fmaf(t, v[j], H[tri(i, j) - E0]);
        }
    }
}

// The patient (lane L) writes its problem into the staging area.  Record of block b at stage[b * COOP_REC]: n t1 t2 (0-8), lever from the centre of cube 0 / cube 1
// (9-11 / 12-14), aref (15-20), Rn (21), Rt (22), m2 of the tangential / torsional / rolling rows (23-25), coefficient of the share of cube 0 / cube 1 (26 / 27:
// -1 arm on that cube, +1 that cube on the floor, -1 / +1 cube 0 / cube 1 of a cube<->cube contact, 0 none), act (28), sign of a limit row (29)
// NCS = cubes of the problem solved here: all NC of the env, or (Stack) ONE -- cube `cs` -- when the patient's only coupling is its arm on that cube (the other cube
// stays with its own 6-dimensional SIMT solve: 12 unknowns instead of 18)
template <int NC, int NRW, int NCC, int NCS, bool WALLS>
DEV void coop_stage(const NewtonCtx<NC, NRW, WALLS, NCC> &C, float *stage, const float (&y)[6], const f3 (&ca)[NC], const f3 (&cal)[NC], int cs) {
    constexpr int NB = coop_nb<NCS, WALLS>();
    auto cube_of = [&](int ci) -> int { return NCS == NC ? ci : cs; };   // env cube behind the problem's cube ci
    const NewtonParams &P = C.P;
    const f3 zero3 = mk(0.f, 0.f, 0.f);
    auto rec = [&](int b, f3 n, f3 t1, f3 t2, f3 rc0, f3 rc1, const float *aref, int naref, float Rn, float Rt, float m2t, float m2s, float m2r, float c0, float c1, bool act, float sign) {
        float *r = stage + b * COOP_REC;
        r[0] = n.x; r[1] = n.y; r[2] = n.z; r[3] = t1.x; r[4] = t1.y; r[5] = t1.z; r[6] = t2.x; r[7] = t2.y; r[8] = t2.z;
        r[9] = rc0.x; r[10] = rc0.y; r[11] = rc0.z; r[12] = rc1.x; r[13] = rc1.y; r[14] = rc1.z;
#pragma unroll
        for (int k = 0; k < 6; k++) r[15 + k] = k < naref ? aref[k] : 0.f;
        r[21] = act ? Rn : 1.f; r[22] = act ? Rt : 1.f; r[23] = m2t; r[24] = m2s; r[25] = m2r; r[26] = c0; r[27] = c1; r[28] = act ? 1.f : 0.f; r[29] = sign;
    };
#pragma unroll
    for (int s = 0; s < NAS; s++) {
        const ArmSlot<NRW> &T = C.AS[s];
        const bool oncube = s < 2 || (s == 4 && C.link_on_cube);
        const bool second = NC == 2 && C.slot_cube[s == 4 ? 2 : (s & 1)] == 1;   // which cube the slot talks to (Stack)
        const float m2t = s < 2 ? P.mu_fc2 : (s < 4 ? MU_FINGER * MU_FINGER : (oncube ? P.mu_c2 : 1.f));
        const float m2s = s < 2 ? P.mu_fct2 : (s < 4 ? MU_TORS * MU_TORS : (oncube ? P.mu_ct2 : 0.f));
        const float m2r = s < 2 ? P.mu_fcr2 : (s < 4 ? MU_ROLL * MU_ROLL : 0.f);
        const bool may_cube = s < 2 || s == 4;
        const int sc = second ? 1 : 0;                                   // the env's cube the slot is on, and which of the problem's cubes that is (-1: none)
        const int sci = (may_cube && oncube) ? (NCS == NC ? sc : (sc == cs ? 0 : -1)) : -1;
        rec(s, T.n, T.t1, T.t2, T.rc, T.rc, T.aref, NRW, T.Rn, T.Rn * P.inv_impratio * m2t, m2t, m2s, m2r, sci == 0 ? -1.f : 0.f, sci == 1 ? -1.f : 0.f, T.act, 0.f);
    }
#pragma unroll
    for (int ci = 0; ci < NCS; ci++)
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const FloorSlot T = cube_of(ci) == 0 ? C.FS[0][s] : C.FS[NC - 1][s];
            rec(5 + 4 * ci + s, mk(0.f, 0.f, 1.f), mk(0.f, 1.f, 0.f), mk(-1.f, 0.f, 0.f), T.r, T.r, T.aref, 4, T.Rn, T.Rn * P.inv_impratio * P.mu_c2, P.mu_c2, P.mu_ct2, 0.f,
                ci == 0 ? 1.f : 0.f, ci == 1 ? 1.f : 0.f, T.act, 0.f);
        }
    if constexpr (NCS == 2) {   // cube<->cube: the force acts at the contact point on cube 1 and, negated, on cube 0 (records of the patient's own LDS column)
#pragma unroll
        for (int s = 0; s < NCC; s++) {
            const bool act = C.cc_any && C.cc_act[s];
            const f3 pos = act ? mk(C.ccl[(size_t)(s * CC_REC_NEWTON + 0) * 64], C.ccl[(size_t)(s * CC_REC_NEWTON + 1) * 64], C.ccl[(size_t)(s * CC_REC_NEWTON + 2) * 64]) : zero3;
            float aref[4];
#pragma unroll
            for (int q = 0; q < 4; q++) aref[q] = act ? C.ccl[(size_t)(s * CC_REC_NEWTON + 7 + q) * 64] : 0.f;
            const float Rn = act ? C.ccl[(size_t)(s * CC_REC_NEWTON + CC_RN_NEWTON) * 64] : 1.f;
            rec(13 + s, C.ccn, C.cct1, C.cct2, pos - C.cp[0], pos - C.cp[NC - 1], aref, 4, Rn, Rn * P.inv_impratio * P.mu_c2, P.mu_c2, P.mu_ct2, 0.f, -1.f, 1.f, act, 0.f);
        }
    }
    if constexpr (WALLS) {   // PushCubeLoop's rails: pair coordinates (a, b, c) = (x, y, z) for the x pair, (y, z, x) for the y pair; n = sg a, t1 = b, t2 = sg c (newton_solve: wall_frame)
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const FloorSlot &T = C.WS[s];
            const float sg = C.wsg[s >> 1];
            const bool xp = (s >> 1) == 0;
            const f3 n = xp ? mk(sg, 0.f, 0.f) : mk(0.f, sg, 0.f), t1 = xp ? mk(0.f, 1.f, 0.f) : mk(0.f, 0.f, 1.f), t2 = xp ? mk(0.f, 0.f, sg) : mk(sg, 0.f, 0.f);
            const f3 r = xp ? T.r : mk(T.r.z, T.r.x, T.r.y);
            rec(5 + 4 * NCS + s, n, t1, t2, r, r, T.aref, 4, T.Rn, T.Rn * P.inv_impratio * P.mu_c2, P.mu_c2, P.mu_ct2, 0.f, 1.f, 0.f, T.act && C.wall_any, 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {   // joint limits: regulariser and reference acceleration as newton_solve computes them
        const bool lower = C.q[j] < JLO[j];
        const float pos = lower ? C.q[j] - JLO[j] : JHI[j] - C.q[j];
        const float imp = impedance(pos, D0_DEF, DW_DEF, 1.0f / W_DEF);
        const float Rn = fmaxf((1.f - imp) * rcp(imp) * INVW_DOF[j], 1e-15f);
        const float aref = -B_DEF * (lower ? 1.f : -1.f) * C.qd[j] - K_DEF * imp * pos;
        rec(NB - 6 + j, zero3, zero3, zero3, zero3, zero3, &aref, 1, Rn, 1.f, 0.f, 0.f, 0.f, 0.f, 0.f, C.lim_act[j], lower ? 1.f : -1.f);
    }
    float *g = stage + NB * COOP_REC;   // the factor of M (strictly lower part row by row, then 1 / L_ii), a0 of the arm, x
    int o = 0;
#pragma unroll
    for (int i = 1; i < 6; i++)
#pragma unroll
        for (int k = 0; k < i; k++) g[o++] = C.CL.L[i][k];
#pragma unroll
    for (int i = 0; i < 6; i++) g[15 + i] = C.CL.id[i];
#pragma unroll
    for (int j = 0; j < 6; j++) { g[21 + j] = C.y0s[j]; g[27 + j] = y[j]; }
#pragma unroll
    for (int ci = 0; ci < NCS; ci++) {
        float *xc = g + 33 + 6 * ci;
        const f3 a = cube_of(ci) == 0 ? ca[0] : ca[NC - 1], al = cube_of(ci) == 0 ? cal[0] : cal[NC - 1];
        xc[0] = a.x; xc[1] = a.y; xc[2] = a.z; xc[3] = al.x; xc[4] = al.y; xc[5] = al.z;
    }
}

// the solve; `lane` = this lane, L = the patient's lane (wave-uniform).  Returns the Newton iterations it took; the patient's y / ca / cal and the forces of its slots are updated.
template <int NC, int NRW, int NCC, int NCS = NC, bool WALLS = false>
DEV int coop_solve(NewtonCtx<NC, NRW, WALLS, NCC> &C, float *stage, int lane, int L, float (&y)[6], f3 (&ca)[NC], f3 (&cal)[NC], int cs = 0) {
    constexpr int NX = coop_nx<NCS>(), NH = NX * (NX + 1) / 2, NB = coop_nb<NCS, WALLS>(), RED = coop_red<NCS>(), NCH = coop_chunks<NCS>();
    constexpr bool REBUILD = NCS == 2;
    // the rows are rebuilt where they are used instead of living next to the 171-entry Hessian
    const NewtonParams &P = C.P;
    if (lane == L) coop_stage<NC, NRW, NCC, NCS, WALLS>(C, stage, y, ca, cal, cs);
    lds_fence();
    // ---- every lane: its block's record, the shared vectors ----
    const int b = lane < NB ? lane : NB - 1;   // (the other lanes shadow the last block with act = false)
    const float *r = stage + b * COOP_REC;
    const f3 dn = mk(r[0], r[1], r[2]), dt1 = mk(r[3], r[4], r[5]), dt2 = mk(r[6], r[7], r[8]);
    const f3 rcs[2] = {mk(r[9], r[10], r[11]), mk(r[12], r[13], r[14])};
    float aref[6];
#pragma unroll
    for (int k = 0; k < 6; k++) aref[k] = r[15 + k];
    const float Rn = r[21], Rt = r[22], lsign = r[29];
    const float coefs[2] = {r[26], r[27]};
    const float m2[6] = {1.f, r[23], r[23], r[24], r[25], r[25]};
    const bool act = lane < NB && r[28] != 0.f;
    const float *gs = stage + NB * COOP_REC;
    float x[NX], x0[NX];
#pragma unroll
    for (int i = 0; i < 6; i++) x0[i] = gs[21 + i];
#pragma unroll
    for (int i = 6; i < NX; i++) x0[i] = (i - 6) % 6 == 2 ? -GRAV : 0.f;
#pragma unroll
    for (int i = 0; i < NX; i++) x[i] = gs[27 + i];
    float lrow6[6];   // a joint limit's row L^-1 (+-e_j): kept (the factor of M lives in the staging area, which the reduction overwrites)
#pragma unroll
    for (int k = 0; k < 6; k++) lrow6[k] = 0.f;
    if (act && b >= NB - 6) {
        const int j = b - (NB - 6);
        float Ls[15], id[6];
#pragma unroll
        for (int k = 0; k < 15; k++) Ls[k] = gs[k];
#pragma unroll
        for (int k = 0; k < 6; k++) { id[k] = gs[15 + k]; lrow6[k] = k == j ? lsign : 0.f; }
#pragma unroll
        for (int i = 0; i < 6; i++) {   // forward substitution (fsub)
            float s = lrow6[i];
#pragma unroll
            for (int k = 0; k < i; k++) s = fmaf(-Ls[i * (i - 1) / 2 + k], lrow6[k], s);
            lrow6[i] = s * id[i];
        }
    }
    const float cm = P.cube_mass, ci = rcp(P.cube_iinv);
    auto mdiag = [&](int i) -> float { return i < 6 ? 1.f : ((i - 6) % 6 < 3 ? cm : ci); };
    // ---- the rows of this lane's block, dense in the NX unknowns ----
    auto ldrow = [&](int lrow, float (&o6)[6]) {   // g row `lrow` of the patient's LDS column
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float2v gp = *reinterpret_cast<const float2v *>(&C.lds[lrow * LDS_ROW + k * 128 + L * 2]);
            o6[2 * k] = gp.x; o6[2 * k + 1] = gp.y;
        }
    };
    auto build_rows = [&](float (&J)[6][NX]) {
#pragma unroll
        for (int q = 0; q < 6; q++)
#pragma unroll
            for (int i = 0; i < NX; i++) J[q][i] = 0.f;
        if (act) {   // (a block that is off keeps zero rows: the g rows of a slot no lane of the wave touches were never written)
            if (b < 4) {   // a finger slot: three linear rows, then torsion / rolling = d . B of the finger body's three angular rows
                float bx[6], by[6], bz[6];
                const int b0 = NEWTON_BODY_ROW0 + 3 * (b & 1);
                ldrow(b0, bx); ldrow(b0 + 1, by); ldrow(b0 + 2, bz);
#pragma unroll
                for (int q = 0; q < 3; q++) {
                    float g6[6];
                    ldrow(3 * b + q, g6);
                    const f3 dd = q == 0 ? dn : (q == 1 ? dt1 : dt2);
#pragma unroll
                    for (int k = 0; k < 6; k++) { J[q][k] = g6[k]; J[3 + q][k] = fmaf(dd.x, bx[k], fmaf(dd.y, by[k], dd.z * bz[k])); }
                }
            } else if (b == 4) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    float g6[6];
                    ldrow(18 + q, g6);
#pragma unroll
                    for (int k = 0; k < 6; k++) J[q][k] = g6[k];
                }
            } else if (b >= NB - 6) {
#pragma unroll
                for (int k = 0; k < 6; k++) J[0][k] = lrow6[k];
            }
            // a cube's share: its contact point moves with ca + cal x rc (rows 0-2); rows 3-5 see cal
#pragma unroll
            for (int c = 0; c < NCS; c++)
#pragma unroll
                for (int q = 0; q < 6; q++) {
                    const f3 d = (q % 3) == 0 ? dn : ((q % 3) == 1 ? dt1 : dt2);
                    const f3 lin = q < 3 ? coefs[c] * d : mk(0.f, 0.f, 0.f), ang = q < 3 ? coefs[c] * cross(rcs[c], d) : coefs[c] * d;
                    J[q][6 + 6 * c + 0] = lin.x; J[q][6 + 6 * c + 1] = lin.y; J[q][6 + 6 * c + 2] = lin.z;
                    J[q][6 + 6 * c + 3] = ang.x; J[q][6 + 6 * c + 4] = ang.y; J[q][6 + 6 * c + 5] = ang.z;
                }
        }
    };
    float Jk[REBUILD ? 1 : 6][REBUILD ? 1 : NX];   // (one cube: the rows stay in registers)
    auto with_rows = [&](auto &&fn) {
        if constexpr (REBUILD) {
            float J[6][NX];
            build_rows(J);
            fn(J);
        } else fn(Jk);
    };
    if constexpr (!REBUILD) build_rows(Jk);
    float scale = fmaf((float)NC * cm, GRAV * GRAV, 1.f);
#pragma unroll
    for (int j = 0; j < 6; j++) scale = fmaf(x0[j], x0[j], scale);
    const float tol2 = P.newton_tol * P.newton_tol * scale;
    float zs[6], jd[6];
    with_rows([&](const float (&J)[6][NX]) {
#pragma unroll
        for (int q = 0; q < 6; q++) {
            float a = -aref[q];
#pragma unroll
            for (int i = 0; i < NX; i++) a = fmaf(J[q][i], x[i], a);
            zs[q] = act ? a : 0.f;
            jd[q] = 0.f;
        }
    });
    float *shares = stage, *totals = stage + (NB * COOP_CH > coop_stage_floats<NCS, WALLS>() ? NB * COOP_CH : coop_stage_floats<NCS, WALLS>());
    int its = 0;
    float dprev = 3.0e38f;
    for (int it = 0; it < P.newton_iters; it++) {
        float dx[NX], d0 = 0.f;
        {
            // this block's share of the gradient and of the Hessian, chunk by chunk of 64 numbers: the shares [block][64] are written, column e is summed by lane e,
            // the totals [RED] collect
            BlkEval<6> B;
            blk_eval<6>(zs, Rn, Rt, m2, act, B);
            with_rows([&](const float (&J)[6][NX]) {
                float wv[NX], v[NX];   // J'WJ = av v v' - gam w w' + sum_t kap m2[t] row_t row_t',  w = sum_t c[t] row_t,  v = row_n - w
#pragma unroll
                for (int i = 0; i < NX; i++) {
                    float a = 0.f;
#pragma unroll
                    for (int q = 1; q < 6; q++) a = fmaf(B.c[q], J[q][i], a);
                    wv[i] = a;
                    v[i] = J[0][i] - a;
                }
                auto chunk = [&](auto ch_tag) {
                    constexpr int CHI = decltype(ch_tag)::value;
                    constexpr int V0 = CHI * COOP_CH, V1 = (CHI + 1) * COOP_CH < RED ? (CHI + 1) * COOP_CH : RED;     // values [V0, V1) of (gradient | packed Hessian)
                    constexpr int E0 = V0 > NX ? V0 - NX : 0, E1 = V1 > NX ? V1 - NX : 0;                           // ... packed Hessian entries [E0, E1)
                    float vals[COOP_CH];
#pragma unroll
                    for (int i = 0; i < COOP_CH; i++) vals[i] = 0.f;
                    if constexpr (V0 < NX) {
#pragma unroll
                        for (int i = V0; i < (V1 < NX ? V1 : NX); i++) {
                            float a = 0.f;
#pragma unroll
                            for (int q = 0; q < 6; q++) a = fmaf(-B.f[q], J[q][i], a);
                            vals[i - V0] = a;
                        }
                    }
                    if constexpr (E1 > E0) {
                        float Hl[E1 - E0];
#pragma unroll
                        for (int i = 0; i < E1 - E0; i++) Hl[i] = 0.f;
                        h_rank1_part<NX, E0, E1>(Hl, v, B.av);
                        h_rank1_part<NX, E0, E1>(Hl, wv, -B.gam);
#pragma unroll
                        for (int q = 1; q < 6; q++) h_rank1_part<NX, E0, E1>(Hl, J[q], B.kap * m2[q]);
#pragma unroll
                        for (int i = 0; i < E1 - E0; i++) vals[E0 + NX - V0 + i] = Hl[i];
                    }
                    if (lane < NB) {
                        // (plain float stores, which the compiler merges into wide LDS writes itself: stores through a float4 lvalue are, to its alias analysis, unrelated
                        //  to the float loads of the staging records and of the column sums -- it moved those loads across them, and the solve read clobbered records)
                        float *w = shares + lane * COOP_CH;
#pragma unroll
                        for (int i = 0; i < COOP_CH; i++) w[i] = vals[i];
                    }
                    lds_fence();
                    float t = 0.f;
#pragma unroll
                    for (int k = 0; k < NB; k++) t += shares[k * COOP_CH + lane];
                    totals[CHI * COOP_CH + lane] = t;
                    lds_fence();
                };
                chunk(std::integral_constant<int, 0>{});
                chunk(std::integral_constant<int, 1>{});
                if constexpr (NCH > 2) chunk(std::integral_constant<int, 2>{});
            });
            float Hm[NH], g[NX], hid[NX];
#pragma unroll
            for (int i = 0; i < NX; i++) g[i] = fmaf(mdiag(i), x[i] - x0[i], totals[i]);
#pragma unroll
            for (int i = 0; i < NH; i++) Hm[i] = totals[NX + i];
#pragma unroll
            for (int i = 0; i < NX; i++) Hm[tri(i, i)] += mdiag(i);
            chol_packed<NX>(Hm, hid);
#pragma unroll
            for (int i = 0; i < NX; i++) dx[i] = -g[i];
            solve_packed<NX>(Hm, hid, dx);
#pragma unroll
            for (int i = 0; i < NX; i++) d0 = fmaf(g[i], dx[i], d0);
        }
        float dist2 = 0.f;
#pragma unroll
        for (int i = 0; i < NX; i++) dist2 = fmaf(mdiag(i) * (x[i] - x0[i]), x[i] - x0[i], dist2);
        const bool live = -d0 > tol2 && !(-d0 <= DEC_FLOOR * dist2 && -d0 >= 0.25f * dprev);   // (wave-uniform: every lane holds the same numbers)
        dprev = -d0;
        if (!live) break;
        its++;
        with_rows([&](const float (&J)[6][NX]) {
#pragma unroll
            for (int q = 0; q < 6; q++) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < NX; i++) a = fmaf(J[q][i], dx[i], a);
                jd[q] = act ? a : 0.f;
            }
        });
        // the line search of newton_solve with this block's share of phi' / phi'' summed over the wave
        float q1 = 0.f, q0 = 0.f;
#pragma unroll
        for (int i = 0; i < NX; i++) { q1 = fmaf(mdiag(i) * dx[i], dx[i], q1); q0 = fmaf(mdiag(i) * (x[i] - x0[i]), dx[i], q0); }
        auto ls_eval = [&](float al, float &ef, float &eh) {
            float z[6];
#pragma unroll
            for (int q = 0; q < 6; q++) z[q] = fmaf(al, jd[q], zs[q]);
            BlkEval<6> B;
            blk_eval<6>(z, Rn, Rt, m2, act, B);
            float acc = 0.f, u = 0.f, tt = 0.f;
#pragma unroll
            for (int q = 0; q < 6; q++) acc = fmaf(B.f[q], jd[q], acc);
#pragma unroll
            for (int q = 1; q < 6; q++) { u = fmaf(B.c[q], jd[q], u); tt = fmaf(m2[q] * jd[q], jd[q], tt); }
            const float sn = jd[0] - u;
            const float hac = fmaf(B.av * sn, sn, fmaf(-B.gam * u, u, B.kap * tt));
            ef = wave_sum(acc);
            eh = wave_sum(hac);
        };
        auto kink_cand = [&](float lo, float hi, float sec) -> float {   // (rare) the sticking-zone minimiser of a block's N(al) closest to `sec`; ties: the first block
            float qa = 0.f, qb = 0.f, qc = 0.f;
#pragma unroll
            for (int q = 1; q < 6; q++) {
                const float mj = m2[q] * jd[q];
                qa = fmaf(mj, jd[q], qa); qb = fmaf(mj, zs[q], qb); qc = fmaf(m2[q] * zs[q], zs[q], qc);
            }
            const float iqa = rcp(fmaxf(qa, 1e-30f));
            const float am = -qb * iqa;
            const float n2 = fmaxf(fmaf(-qb * qb, iqa, qc), 0.f), wn = fmaf(am, jd[0], zs[0]);
            const bool ok = act && lane < NB - 6 && qa > 0.f && am > lo && am < hi && wn < 0.f && !(n2 * Rn * Rn > wn * wn * Rt * Rt);
            const float dist = fabsf(am - sec);
            float best = sec, bestd = -1.f;
            for (int k = 0; k < NB - 6; k++) {
                const bool okk = __builtin_amdgcn_readlane((int)ok, k) != 0;
                const float amk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(am), k)), dk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dist), k));
                const bool take = okk && (bestd < 0.f || dk < bestd);
                best = take ? amk : best;
                bestd = take ? dk : bestd;
            }
            return best;
        };
        float al = 1.f, lo_a = 0.f, hi_a = -1.f, dlo = d0, dhi = 0.f, hlo = -d0, hhi = 0.f, dlo_m = d0, dhi_m = 0.f;
        int last_side = 0, same = 0;
        bool conv = false;
        for (int ls = 0; ls < P.ls_iters; ls++) {
            float ef, eh;
            ls_eval(al, ef, eh);
            const float mpart = fmaf(al, q1, q0);
            const float dphi = mpart - ef, ddphi = q1 + eh;
            float an = al, sec = al;
            bool need = false;
            const bool fin = fabsf(dphi) <= fmaf(P.ls_tol, fabsf(d0), LS_NOISE * (fabsf(mpart) + fabsf(ef)));
            const int side = dphi < 0.f ? -1 : 1;
            if (dphi < 0.f) { if (last_side < 0) dhi_m *= 0.5f; lo_a = al; dlo = dphi; hlo = ddphi; dlo_m = dphi; }
            else { if (last_side > 0) dlo_m *= 0.5f; hi_a = al; dhi = dphi; hhi = ddphi; dhi_m = dphi; }
            same = side == last_side ? same + 1 : 0;
            last_side = side;
            if (hi_a < 0.f) an = lo_a - dlo * rcp(hlo);
            else {
                const float cl = lo_a - dlo * rcp(hlo), ch = hi_a - dhi * rcp(hhi);
                const float mg = 1e-4f * (hi_a - lo_a), blo = lo_a + mg, bhi = hi_a - mg;
                const bool vl = cl > blo && cl < bhi, vh = ch > blo && ch < bhi;
                bool from_lo = fabsf(dlo) <= fabsf(dhi);
                if (same >= 2) from_lo = side > 0;
                sec = lo_a - dlo_m * (hi_a - lo_a) * rcp(dhi_m - dlo_m);
                if (!(sec > lo_a && sec < hi_a)) sec = 0.5f * (lo_a + hi_a);
                an = from_lo ? (vl ? cl : ch) : (vh ? ch : cl);
                need = !fin && !vl && !vh;
            }
            conv = fin;
            if (fin) break;
            if (need) {
                const float mg = 1e-4f * (hi_a - lo_a);
                an = kink_cand(lo_a + mg, hi_a - mg, sec);
            }
            al = an;
        }
        if (!conv) al = lo_a > 0.f ? lo_a : hi_a;
#pragma unroll
        for (int i = 0; i < NX; i++) x[i] = fmaf(al, dx[i], x[i]);
#pragma unroll
        for (int q = 0; q < 6; q++) zs[q] = fmaf(al, jd[q], zs[q]);
    }
    // ---- the forces at the solution and the accelerations go back to the patient ----
    {
        BlkEval<6> B;
        blk_eval<6>(zs, Rn, Rt, m2, act, B);
        if (lane < NB) {
#pragma unroll
            for (int q = 0; q < 6; q++) stage[lane * 8 + q] = B.f[q];
        }
        lds_fence();
        if (lane == L) {
#pragma unroll
            for (int s = 0; s < NAS; s++)
#pragma unroll
                for (int q = 0; q < (s < 4 ? 6 : 4); q++) C.AS[s].f[q] = stage[s * 8 + q];
#pragma unroll
            for (int ci = 0; ci < NCS; ci++)
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float fv = stage[(5 + 4 * ci + s) * 8 + q];
                        if ((NCS == NC ? ci : cs) == 0) C.FS[0][s].f[q] = fv; else C.FS[NC - 1][s].f[q] = fv;
                    }
            if constexpr (NCS == 2) {
                if (C.cc_any) {
#pragma unroll
                    for (int s = 0; s < NCC; s++)
#pragma unroll
                        for (int q = 0; q < 4; q++) C.ccl[(size_t)(s * CC_REC_NEWTON + 3 + q) * 64] = stage[(13 + s) * 8 + q];
                }
            }
            if constexpr (WALLS) {
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int q = 0; q < 4; q++) C.WS[s].f[q] = stage[(5 + 4 * NCS + s) * 8 + q];
            }
#pragma unroll
            for (int j = 0; j < 6; j++) C.flim[j] = stage[(NB - 6 + j) * 8];
#pragma unroll
            for (int j = 0; j < 6; j++) y[j] = x[j];
#pragma unroll
            for (int ci = 0; ci < NCS; ci++) {
                const f3 a = mk(x[6 + 6 * ci], x[7 + 6 * ci], x[8 + 6 * ci]), al = mk(x[9 + 6 * ci], x[10 + 6 * ci], x[11 + 6 * ci]);
                if ((NCS == NC ? ci : cs) == 0) { ca[0] = a; cal[0] = al; } else { ca[NC - 1] = a; cal[NC - 1] = al; }
            }
        }
    }
    C.wave_its += its;
    return its;
}

// ================================================================================================
// One cube (15 blocks): up to FOUR patients at a time, one per 16-lane row of the wave.  Lane 16 p + b owns block b of patient p; the shares of a row are summed
// inside the row by DPP (no LDS, no waits), so each row is its own little solve and the rows run the iteration loop together like the lanes of the SIMT solver do: a
// row that has converged idles (step 0) until the last one has.  The launch time of PushCube / LiftCube / PickPlaceCube was the wave with the most coupled envs (three
// per substep, 61 cooperative solves per control step): four per pass turn its k solves into ceil(k / 4) -- PushCube 3.31 -> 2.64 ms, PickPlace-ee 3.32 -> 2.53.  StackTwoCubes' arm + ONE cube patients
// (12 unknowns, the same 15 blocks) are solved here too: 32 768 envs 6.81 -> 6.03 ms.  PushCubeLoop (rails: 19 blocks) fits a row with its six joint limits in ONE lane
// (LIM1 below): 5.88 -> 4.90 ms.  -DLCR_ROWS_TRACE: every replicated quantity of every iteration is compared with
// the row's first lane and the first differences are printed (how the re-association in row_sum was found).
// ================================================================================================
// sum over the 16 lanes of a row, THE SAME BITS in every lane of the row: every step adds a lane and its partner under an involution (i ^ 1, i ^ 2, mirror of the half
// row, mirror of the row), so both compute a + b = b + a.  (With rotations -- row_ror:4, row_ror:8 -- the quads are added in a different order in each quad; the
// lanes of a row then factorise Hessians that differ in the last bit and their copies of the iterate drift apart.)
DEV float row_sum(float v) {
    auto dpp = [](float x, auto ctrl_tag) -> float {
        constexpr int ctrl = decltype(ctrl_tag)::value;
        return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), ctrl, 0xf, 0xf, false));
    };
    // (each partial sum is made opaque: under -ffast-math the compiler may otherwise re-associate a step with the next one or with the CALLER's arithmetic --
    //  phi'(al) = mpart - (h0 + h1) became (mpart - h0) - h1 in one half of the row and (mpart - h1) - h0 in the other: profiles/r06_rows_trace.txt)
    v += dpp(v, std::integral_constant<int, 0xb1>{});    // quad_perm [1,0,3,2]
    asm("" : "+v"(v));
    v += dpp(v, std::integral_constant<int, 0x4e>{});    // quad_perm [2,3,0,1]
    asm("" : "+v"(v));
    v += dpp(v, std::integral_constant<int, 0x141>{});   // row_half_mirror
    asm("" : "+v"(v));
    v += dpp(v, std::integral_constant<int, 0x140>{});   // row_mirror
    asm("" : "+v"(v));
    return v;
}

// pmask: the patients of this pass (at most four lanes of the wave, ascending: the p-th set bit is row p's patient); c1mask (Stack): the patients whose cube is cube 1
// (each is solved as arm + ITS cube, 12 unknowns, 15 blocks: the other cube keeps its SIMT solve)
#ifdef LCR_ROWS_TRACE
__device__ int lcr_rows_trace_n = 0;
DEV void rows_chk(float v, int lane, bool has, int it, int tag, int idx) {
    const float r = __int_as_float(__builtin_amdgcn_ds_bpermute((lane & 48) << 2, __float_as_int(v)));
    if (has && __float_as_int(r) != __float_as_int(v)) {
        if (atomicAdd(&lcr_rows_trace_n, 1) < 60) printf("ROWS it %d lane %d tag %d idx %d mine %a first %a\n", it, lane, tag, idx, (double)v, (double)r);
    }
}
#define ROWS_CHK(v, tag, idx) rows_chk(v, lane, has, it, tag, idx)
#else
#define ROWS_CHK(v, tag, idx)
#endif
template <int NC, int NRW, int NCC, bool WALLS = false>
DEV void coop_solve_rows(NewtonCtx<NC, NRW, WALLS, NCC> &C, float *stage, int lane, unsigned long long pmask, unsigned long long c1mask, float (&y)[6], f3 (&ca)[NC], f3 (&cal)[NC], int &sweeps_done) {
    // With rails (PushCubeLoop) the patient has 19 blocks: the six joint limits -- blocks of ONE row each -- then share ONE lane (LIM1), whose six rows are six blocks:
    // 13 contact lanes + 1 limit lane per row.  Its evaluation is per row (f_j = max(0, -z_j / R_j)); its Hessian share has the form of a contact's with av = w_0, no
    // cone terms and the weights w_1..5 in place of kap m2[r] (kw below), so everything after the evaluation is shared code.
    constexpr bool LIM1 = WALLS;
    constexpr int NX = coop_nx<1>(), NH = NX * (NX + 1) / 2, NBR = coop_nb<1, WALLS>(), NBC = NBR - 6, NB = LIM1 ? NBC + 1 : NBR;   // records, contact blocks, lanes in use
    static_assert(NB <= 16, "a patient's blocks must fit a 16-lane row");
    const NewtonParams &P = C.P;
    const int row = lane >> 4, li = lane & 15, b = LIM1 ? (li < NBC ? li : NBC - 1) : (li < NB ? li : NB - 1);
    const bool islim = LIM1 && li == NBC;
    int Lp = 0;           // this row's patient lane
    bool has = false;     // this row has a patient
    f3 dn = mk(0.f, 0.f, 1.f), dt1 = mk(0.f, 1.f, 0.f), dt2 = mk(-1.f, 0.f, 0.f), rc = mk(0.f, 0.f, 0.f);
    float aref[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, Rn = 1.f, Rt = 1.f, coef = 0.f, lsign = 0.f, m2t = 0.f, m2s = 0.f, m2r = 0.f;
    bool act = false;
    float x[NX], x0[NX], lrow6[6];
    float limrow[LIM1 ? 6 : 1][6], lim_iR[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // (LIM1, the limit lane: the rows L^-1 (+-e_j) of the active limits, 1 / R_j)
    bool lim_on[6] = {false, false, false, false, false, false};
#pragma unroll
    for (int i = 0; i < NX; i++) { x[i] = 0.f; x0[i] = 0.f; }
#pragma unroll
    for (int k = 0; k < 6; k++) lrow6[k] = 0.f;
    {
        int p = 0;
        for (unsigned long long m = pmask; m != 0ull; m &= m - 1ull, p++) {
            const int L = __builtin_ctzll(m);
            if (lane == L) coop_stage<NC, NRW, NCC, 1, WALLS>(C, stage, y, ca, cal, NC == 2 ? (int)(c1mask >> L & 1ull) : 0);
            lds_fence();
            if (row == p) {
                const float *r = stage + b * COOP_REC;
                const float *gs = stage + NBR * COOP_REC;
                Lp = L; has = true;
                dn = mk(r[0], r[1], r[2]); dt1 = mk(r[3], r[4], r[5]); dt2 = mk(r[6], r[7], r[8]); rc = mk(r[9], r[10], r[11]);
#pragma unroll
                for (int k = 0; k < 6; k++) aref[k] = r[15 + k];
                Rn = r[21]; Rt = r[22]; m2t = r[23]; m2s = r[24]; m2r = r[25]; coef = r[26]; lsign = r[29];
                act = (LIM1 ? li < NBC : li < NB) && r[28] != 0.f;
#pragma unroll
                for (int i = 0; i < 6; i++) x0[i] = gs[21 + i];
#pragma unroll
                for (int i = 0; i < NX; i++) x[i] = gs[27 + i];
                if constexpr (LIM1) {
                    if (islim) {
                        float Ls[15], id[6];
#pragma unroll
                        for (int k = 0; k < 15; k++) Ls[k] = gs[k];
#pragma unroll
                        for (int k = 0; k < 6; k++) id[k] = gs[15 + k];
                        coef = 0.f;
#pragma unroll
                        for (int j = 0; j < 6; j++) {
                            const float *rj = stage + (NBC + j) * COOP_REC;
                            lim_on[j] = rj[28] != 0.f;
                            act = act || lim_on[j];
                            aref[j] = rj[15]; lim_iR[j] = rcp(rj[21]);
                            const float sg = lim_on[j] ? rj[29] : 0.f;
#pragma unroll
                            for (int i = 0; i < 6; i++) {
                                float sacc = i == j ? sg : 0.f;
#pragma unroll
                                for (int k = 0; k < i; k++) sacc = fmaf(-Ls[i * (i - 1) / 2 + k], limrow[j][k], sacc);
                                limrow[j][i] = sacc * id[i];
                            }
                        }
                    }
                }
                if (!LIM1 && act && b >= NB - 6) {   // a joint limit's row L^-1 (+-e_j)
                    const int j = b - (NB - 6);
                    float Ls[15], id[6];
#pragma unroll
                    for (int k = 0; k < 15; k++) Ls[k] = gs[k];
#pragma unroll
                    for (int k = 0; k < 6; k++) { id[k] = gs[15 + k]; lrow6[k] = k == j ? lsign : 0.f; }
#pragma unroll
                    for (int i = 0; i < 6; i++) {
                        float sacc = lrow6[i];
#pragma unroll
                        for (int k = 0; k < i; k++) sacc = fmaf(-Ls[i * (i - 1) / 2 + k], lrow6[k], sacc);
                        lrow6[i] = sacc * id[i];
                    }
                }
            }
            lds_fence();   // (the record is in registers before the next patient overwrites it)
        }
    }
#pragma unroll
    for (int i = 6; i < NX; i++) x0[i] = i == 8 ? -GRAV : 0.f;
    const float m2[6] = {1.f, m2t, m2t, m2s, m2r, m2r};
    const float cm = P.cube_mass, ci = rcp(P.cube_iinv);
    auto mdiag = [&](int i) -> float { return i < 6 ? 1.f : (i < 9 ? cm : ci); };
    float J[6][NX];
#pragma unroll
    for (int q = 0; q < 6; q++)
#pragma unroll
        for (int i = 0; i < NX; i++) J[q][i] = 0.f;
    auto ldrow = [&](int lrow, float (&o6)[6]) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float2v gp = *reinterpret_cast<const float2v *>(&C.lds[lrow * LDS_ROW + k * 128 + Lp * 2]);
            o6[2 * k] = gp.x; o6[2 * k + 1] = gp.y;
        }
    };
    if (act) {
        if (b < 4) {
            float bx[6], by[6], bz[6];
            const int b0 = NEWTON_BODY_ROW0 + 3 * (b & 1);
            ldrow(b0, bx); ldrow(b0 + 1, by); ldrow(b0 + 2, bz);
#pragma unroll
            for (int q = 0; q < 3; q++) {
                float g6[6];
                ldrow(3 * b + q, g6);
                const f3 dd = q == 0 ? dn : (q == 1 ? dt1 : dt2);
#pragma unroll
                for (int k = 0; k < 6; k++) { J[q][k] = g6[k]; J[3 + q][k] = fmaf(dd.x, bx[k], fmaf(dd.y, by[k], dd.z * bz[k])); }
            }
        } else if (b == 4) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                float g6[6];
                ldrow(18 + q, g6);
#pragma unroll
                for (int k = 0; k < 6; k++) J[q][k] = g6[k];
            }
        } else if (!LIM1 && b >= NB - 6) {
#pragma unroll
            for (int k = 0; k < 6; k++) J[0][k] = lrow6[k];
        }
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const f3 d = (q % 3) == 0 ? dn : ((q % 3) == 1 ? dt1 : dt2);
            const f3 lin = q < 3 ? coef * d : mk(0.f, 0.f, 0.f), ang = q < 3 ? coef * cross(rc, d) : coef * d;
            J[q][6] = lin.x; J[q][7] = lin.y; J[q][8] = lin.z; J[q][9] = ang.x; J[q][10] = ang.y; J[q][11] = ang.z;
        }
        if constexpr (LIM1) {
            if (islim) {
#pragma unroll
                for (int j = 0; j < 6; j++)
#pragma unroll
                    for (int k = 0; k < 6; k++) J[j][k] = limrow[j][k];
            }
        }
    }
    // this lane's forces and curvature weights at the residuals z: a contact block's cone zone (blk_eval), or -- the limit lane -- six one-row blocks
    auto eval = [&](const float (&z)[6], BlkEval<6> &B, float (&kw)[6]) {
        if (!islim) {
            blk_eval<6>(z, Rn, Rt, m2, act, B);
#pragma unroll
            for (int q = 0; q < 6; q++) kw[q] = B.kap * m2[q];
        } else {
#pragma unroll
            for (int q = 0; q < 6; q++) {
                const float w = (lim_on[q] && z[q] < 0.f) ? lim_iR[q] : 0.f;
                B.f[q] = -z[q] * w; B.c[q] = 0.f; kw[q] = w;
            }
            B.av = kw[0]; B.gam = 0.f; B.kap = 0.f;
        }
    };
    float scale = fmaf((float)NC * cm, GRAV * GRAV, 1.f);
#pragma unroll
    for (int j = 0; j < 6; j++) scale = fmaf(x0[j], x0[j], scale);
    const float tol2 = P.newton_tol * P.newton_tol * scale;
    float zs[6], jd[6];
#pragma unroll
    for (int q = 0; q < 6; q++) {
        float a = -aref[q];
#pragma unroll
        for (int i = 0; i < NX; i++) a = fmaf(J[q][i], x[i], a);
        zs[q] = act ? a : 0.f;
        jd[q] = 0.f;
    }
    int its = 0;
    float dprev = 3.0e38f;
    for (int it = 0; it < P.newton_iters; it++) {
        float dx[NX], d0 = 0.f;
        {
            BlkEval<6> B;
            float kw[6];
            if constexpr (LIM1) eval(zs, B, kw); else blk_eval<6>(zs, Rn, Rt, m2, act, B);
            float Hm[NH], g[NX], hid[NX];
#pragma unroll
            for (int i = 0; i < NH; i++) Hm[i] = 0.f;
            if constexpr (LIM1) {   // (h_block with the weights kw[r] in place of kap m2[r])
                float wv[NX], v[NX];
#pragma unroll
                for (int i = 0; i < NX; i++) {
                    float acc = 0.f;
#pragma unroll
                    for (int r = 1; r < 6; r++) acc = fmaf(B.c[r], J[r][i], acc);
                    wv[i] = acc; v[i] = J[0][i] - acc;
                }
                h_rank1<0, NX, NX>(Hm, v, B.av);
                h_rank1<0, NX, NX>(Hm, wv, -B.gam);
#pragma unroll
                for (int r = 1; r < 6; r++) h_rank1<0, NX, NX>(Hm, J[r], kw[r]);
            } else h_block<0, NX, NX, 6>(Hm, J, B, m2);
#pragma unroll
            for (int i = 0; i < NH; i++) Hm[i] = row_sum(Hm[i]);
#pragma unroll
            for (int i = 0; i < NH; i++) ROWS_CHK(Hm[i], 0, i);
#pragma unroll
            for (int i = 0; i < NX; i++) ROWS_CHK(x[i], 6, i);
#pragma unroll
            for (int i = 0; i < NX; i++) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < 6; q++) a = fmaf(-B.f[q], J[q][i], a);
                g[i] = fmaf(mdiag(i), x[i] - x0[i], row_sum(a));
            }
#pragma unroll
            for (int i = 0; i < NX; i++) Hm[tri(i, i)] += mdiag(i);
#pragma unroll
            for (int i = 0; i < NX; i++) ROWS_CHK(g[i], 1, i);
#pragma unroll
            for (int i = 0; i < NH; i++) ROWS_CHK(Hm[i], 9, i);
            chol_packed<NX>(Hm, hid);
#pragma unroll
            for (int i = 0; i < NH; i++) ROWS_CHK(Hm[i], 2, i);
#pragma unroll
            for (int i = 0; i < NX; i++) ROWS_CHK(hid[i], 3, i);
#pragma unroll
            for (int i = 0; i < NX; i++) dx[i] = -g[i];
            solve_packed<NX>(Hm, hid, dx);
#pragma unroll
            for (int i = 0; i < NX; i++) ROWS_CHK(dx[i], 4, i);
#pragma unroll
            for (int i = 0; i < NX; i++) d0 = fmaf(g[i], dx[i], d0);
            ROWS_CHK(d0, 5, 0);
            // (every lane of the row now holds the same dx and d0 to the bit: computed from the same sums by the same instructions.  Until row_sum made its partial sums
            //  opaque the two halves of a row disagreed in the last bit of phi'(al) -- see there -- and the copies of x drifted apart)
        }
        float dist2 = 0.f;
#pragma unroll
        for (int i = 0; i < NX; i++) dist2 = fmaf(mdiag(i) * (x[i] - x0[i]), x[i] - x0[i], dist2);
        const bool live = has && -d0 > tol2 && !(-d0 <= DEC_FLOOR * dist2 && -d0 >= 0.25f * dprev);   // (the same in every lane of a row)
        dprev = -d0;
        if (!__any(live)) break;
        its += live ? 1 : 0;
#pragma unroll
        for (int q = 0; q < 6; q++) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < NX; i++) a = fmaf(J[q][i], dx[i], a);
            jd[q] = act ? a : 0.f;
        }
        float q1 = 0.f, q0 = 0.f;
#pragma unroll
        for (int i = 0; i < NX; i++) { q1 = fmaf(mdiag(i) * dx[i], dx[i], q1); q0 = fmaf(mdiag(i) * (x[i] - x0[i]), dx[i], q0); }
        auto ls_eval = [&](float al, float &ef, float &eh) {
            float z[6];
#pragma unroll
            for (int q = 0; q < 6; q++) z[q] = fmaf(al, jd[q], zs[q]);
            BlkEval<6> B;
            float kw[6];
            if constexpr (LIM1) eval(z, B, kw); else blk_eval<6>(z, Rn, Rt, m2, act, B);
            float acc = 0.f, u = 0.f, tt = 0.f;
#pragma unroll
            for (int q = 0; q < 6; q++) acc = fmaf(B.f[q], jd[q], acc);
#pragma unroll
            for (int q = 1; q < 6; q++) { u = fmaf(B.c[q], jd[q], u); tt = LIM1 ? fmaf(kw[q] * jd[q], jd[q], tt) : fmaf(m2[q] * jd[q], jd[q], tt); }
            const float sn = jd[0] - u;
            ef = row_sum(acc);
            eh = row_sum(fmaf(B.av * sn, sn, fmaf(-B.gam * u, u, LIM1 ? tt : B.kap * tt)));
        };
        auto kink_cand = [&](float lo, float hi, float sec) -> float {   // (rare)
            float qa = 0.f, qb = 0.f, qc = 0.f;
#pragma unroll
            for (int q = 1; q < 6; q++) {
                const float mj = m2[q] * jd[q];
                qa = fmaf(mj, jd[q], qa); qb = fmaf(mj, zs[q], qb); qc = fmaf(m2[q] * zs[q], zs[q], qc);
            }
            const float iqa = rcp(fmaxf(qa, 1e-30f));
            const float am = -qb * iqa;
            const float n2 = fmaxf(fmaf(-qb * qb, iqa, qc), 0.f), wn = fmaf(am, jd[0], zs[0]);
            const bool ok = act && li < NBC && qa > 0.f && am > lo && am < hi && wn < 0.f && !(n2 * Rn * Rn > wn * wn * Rt * Rt);
            const float dist = fabsf(am - sec), okf = ok ? 1.f : 0.f;
            float best = sec;
            for (int rr = 0; rr < 4; rr++) {   // (v_readlane with wave-uniform lane numbers; the row that is meant keeps the result)
                float bst = sec, bd = -1.f;
                for (int k = 0; k < NBC; k++) {
                    const bool okk = __builtin_amdgcn_readlane(__float_as_int(okf), 16 * rr + k) != 0;
                    const float amk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(am), 16 * rr + k)), dk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(dist), 16 * rr + k));
                    const bool take = okk && (bd < 0.f || dk < bd);
                    bst = take ? amk : bst;
                    bd = take ? dk : bd;
                }
                best = row == rr ? bst : best;
            }
            return best;
        };
        float al = 1.f, lo_a = 0.f, hi_a = -1.f, dlo = d0, dhi = 0.f, hlo = -d0, hhi = 0.f, dlo_m = d0, dhi_m = 0.f;
        int last_side = 0, same = 0;
        bool done = !live, conv = !live;
        for (int ls = 0; ls < P.ls_iters; ls++) {
            float ef, eh;
            ls_eval(al, ef, eh);
            const float mpart = fmaf(al, q1, q0);
            const float dphi = mpart - ef, ddphi = q1 + eh;
            float an = al, sec = al;
            bool need = false;
            if (!done) {
                const bool fin = fabsf(dphi) <= fmaf(P.ls_tol, fabsf(d0), LS_NOISE * (fabsf(mpart) + fabsf(ef)));
                const int side = dphi < 0.f ? -1 : 1;
                if (dphi < 0.f) { if (last_side < 0) dhi_m *= 0.5f; lo_a = al; dlo = dphi; hlo = ddphi; dlo_m = dphi; }
                else { if (last_side > 0) dlo_m *= 0.5f; hi_a = al; dhi = dphi; hhi = ddphi; dhi_m = dphi; }
                same = side == last_side ? same + 1 : 0;
                last_side = side;
                if (hi_a < 0.f) an = lo_a - dlo * rcp(hlo);
                else {
                    const float cl = lo_a - dlo * rcp(hlo), ch = hi_a - dhi * rcp(hhi);
                    const float mg = 1e-4f * (hi_a - lo_a), blo = lo_a + mg, bhi = hi_a - mg;
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
            if (__any(need)) {
                const float mg = 1e-4f * (hi_a - lo_a);
                const float kc = kink_cand(lo_a + mg, hi_a - mg, sec);
                an = need ? kc : an;
            }
            al = done ? al : an;
            if (__all(done)) break;
        }
        if (!conv) al = lo_a > 0.f ? lo_a : hi_a;
        const float step = live ? al : 0.f;
        ROWS_CHK(step, 7, 0);
#pragma unroll
        for (int i = 0; i < NX; i++) x[i] = fmaf(step, dx[i], x[i]);
#pragma unroll
        for (int q = 0; q < 6; q++) zs[q] = fmaf(step, jd[q], zs[q]);
    }
    // ---- forces and accelerations back to the patients: [lane][8] forces, then [row][12] x ----
    {
        BlkEval<6> B;
        float kw[6];
        if constexpr (LIM1) eval(zs, B, kw); else blk_eval<6>(zs, Rn, Rt, m2, act, B);
#pragma unroll
        for (int q = 0; q < 6; q++) stage[lane * 8 + q] = B.f[q];
        if ((lane & 15) == 0) {
#pragma unroll
            for (int i = 0; i < NX; i++) stage[512 + row * NX + i] = x[i];
            stage[512 + 4 * NX + row] = (float)its;
        }
        lds_fence();
        int p = 0;
        for (unsigned long long m = pmask; m != 0ull; m &= m - 1ull, p++) {
            const int L = __builtin_ctzll(m);
            if (lane == L) {
                const float *fr = stage + p * 16 * 8;
#pragma unroll
                for (int s = 0; s < NAS; s++)
#pragma unroll
                    for (int q = 0; q < (s < 4 ? 6 : 4); q++) C.AS[s].f[q] = fr[s * 8 + q];
                const bool second = NC == 2 && (c1mask >> L & 1ull) != 0ull;
#pragma unroll
                for (int s = 0; s < 4; s++)
#pragma unroll
                    for (int q = 0; q < 4; q++) { if (second) C.FS[NC - 1][s].f[q] = fr[(5 + s) * 8 + q]; else C.FS[0][s].f[q] = fr[(5 + s) * 8 + q]; }
#pragma unroll
                for (int j = 0; j < 6; j++) C.flim[j] = LIM1 ? fr[NBC * 8 + j] : fr[(NB - 6 + j) * 8];
                if constexpr (WALLS) {
#pragma unroll
                    for (int s = 0; s < 4; s++)
#pragma unroll
                        for (int q = 0; q < 4; q++) C.WS[s].f[q] = fr[(9 + s) * 8 + q];
                }
                const float *xr = stage + 512 + p * NX;
#pragma unroll
                for (int j = 0; j < 6; j++) y[j] = xr[j];
                if (second) { ca[NC - 1] = mk(xr[6], xr[7], xr[8]); cal[NC - 1] = mk(xr[9], xr[10], xr[11]); }
                else { ca[0] = mk(xr[6], xr[7], xr[8]); cal[0] = mk(xr[9], xr[10], xr[11]); }
                sweeps_done = (int)stage[512 + 4 * NX + p];
            }
        }
        lds_fence();
        C.wave_its += its;
    }
}

}  // namespace
