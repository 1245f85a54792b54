// lcr_newton_coop.h -- the COUPLED arm + cube problem of ONE env solved by the whole wave (one-cube Newton kernels).
//
// Why.  One lane per env makes a wave pay for its slowest lane, and the coupled problem (12 unknowns, 78-entry Hessian, 50 constraint rows) is 2.5 x the cost of the
// two 6-dimensional ones per iteration with every lane of the wave dragged through it: round 5's launch time was the one wave in which ONE lane had a finger on its
// cube for all 20 substeps (7 of 12-dim iterations per substep at 45 k cycles each, profiles/r05_newton_phases.txt).  While few lanes of a wave are coupled
// (LcrDev::coop_max, default 3) the others keep their two small solves (newton_solve<.., 1> / <.., 2> with the coupled lanes disabled) and each coupled env -- the
// "patient", lane L -- is solved here by all 64 lanes: the same algorithm (lcr_newton.h; oracle: newton_product), laid out the other way round:
//
//   lane b < 15 owns ONE constraint block of the patient: b = 0..4 the arm slots (finger<->cube 0 1, finger<->floor 2 3, link proxy 4), 5..8 the cube's floor
//   slots, 9..14 the joint limits (a limit is a block of one row: blk_eval with no friction rows is max(0, -z / R)).  It keeps its block's rows as dense vectors
//   in the 12 unknowns (6 x 12 registers), evaluates its cone zone, its share of the gradient J'f and of the Hessian J'WJ (the SIMT code of ONE slot: blk_eval,
//   h_block), and its share of phi'(al) / phi''(al) in the line search;
//   the shares are summed through LDS (gradient + Hessian: 90 numbers per block, [block][90] written, summed column-wise by lanes 0..63 / 0..25, the totals read
//   back by every lane) or by DPP (the two sums of a line-search evaluation);
//   every lane then holds the whole 12 x 12 system and factorises / solves it redundantly (chol_packed<12>), so the iteration logic -- tolerances, the bracketing
//   line search, the exits -- is the SIMT code with wave-uniform values.
//
// The patient's data reach the other lanes through an LDS staging area it writes itself (block records, x, a0, the Cholesky factor of M for the limit rows); its
// results (accelerations, the force of every row) go back the same way.  ~2 k instructions per iteration and patient instead of ~7 k for all lanes.
#pragma once
#include <type_traits>

#include "lcr_newton.h"

namespace {

constexpr int COOP_NB = 15;                      // blocks of one patient (one cube, no rails)
constexpr int COOP_NX = 12;
constexpr int COOP_NH = COOP_NX * (COOP_NX + 1) / 2;
constexpr int COOP_RED = COOP_NX + COOP_NH;      // numbers a block contributes to one Newton step: gradient + packed Hessian
constexpr int COOP_REC = 28;                     // floats of a block record in the staging area
constexpr int COOP_FLOATS = COOP_NB * COOP_RED + COOP_RED + 6;   // LDS floats of the cooperative solve: [15][90] shares + [90] totals (the staging area overlays them) -> 5.8 KiB
static_assert(COOP_NB * COOP_REC + 21 + 12 + 6 + 4 <= COOP_NB * COOP_RED, "staging area fits under the shares");

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

// The patient (lane L) writes its problem into the staging area.  Record of block b at stage[b * COOP_REC]: n t1 t2 (0-8), rc (9-11), aref (12-17), Rn (18), Rt (19),
// m2 of the tangential / torsional / rolling rows (20-22), coef of the cube part (23: -1 arm on the cube, +1 cube on the floor, 0 none), act (24), sign of a limit row (25)
template <int NC, int NRW, int NCC>
DEV void coop_stage(const NewtonCtx<NC, NRW, false, NCC> &C, float *stage, const float (&y)[6], const f3 (&ca)[NC], const f3 (&cal)[NC]) {
    const NewtonParams &P = C.P;
    auto rec = [&](int b, f3 n, f3 t1, f3 t2, f3 rc, const float *aref, int naref, float Rn, float Rt, float m2t, float m2s, float m2r, float coef, bool act, float sign) {
        float *r = stage + b * COOP_REC;
        r[0] = n.x; r[1] = n.y; r[2] = n.z; r[3] = t1.x; r[4] = t1.y; r[5] = t1.z; r[6] = t2.x; r[7] = t2.y; r[8] = t2.z; r[9] = rc.x; r[10] = rc.y; r[11] = rc.z;
#pragma unroll
        for (int k = 0; k < 6; k++) r[12 + k] = k < naref ? aref[k] : 0.f;
        r[18] = Rn; r[19] = Rt; r[20] = m2t; r[21] = m2s; r[22] = m2r; r[23] = coef; r[24] = act ? 1.f : 0.f; r[25] = sign;
    };
#pragma unroll
    for (int s = 0; s < NAS; s++) {
        const ArmSlot<NRW> &T = C.AS[s];
        const bool oncube = s < 2 || (s == 4 && C.link_on_cube);
        const float m2t = s < 2 ? P.mu_fc2 : (s < 4 ? MU_FINGER * MU_FINGER : (oncube ? P.mu_c2 : 1.f));
        const float m2s = s < 2 ? P.mu_fct2 : (s < 4 ? MU_TORS * MU_TORS : (oncube ? P.mu_ct2 : 0.f));
        const float m2r = s < 2 ? P.mu_fcr2 : (s < 4 ? MU_ROLL * MU_ROLL : 0.f);
        rec(s, T.n, T.t1, T.t2, T.rc, T.aref, NRW, T.Rn, T.Rn * P.inv_impratio * m2t, m2t, m2s, m2r, oncube ? -1.f : 0.f, T.act, 0.f);
    }
#pragma unroll
    for (int s = 0; s < 4; s++) {
        const FloorSlot &T = C.FS[0][s];
        rec(5 + s, mk(0.f, 0.f, 1.f), mk(0.f, 1.f, 0.f), mk(-1.f, 0.f, 0.f), T.r, T.aref, 4, T.Rn, T.Rn * P.inv_impratio * P.mu_c2, P.mu_c2, P.mu_ct2, 0.f, 1.f, T.act, 0.f);
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {   // joint limits: regulariser and reference acceleration as newton_solve computes them
        const bool lower = C.q[j] < JLO[j];
        const float pos = lower ? C.q[j] - JLO[j] : JHI[j] - C.q[j];
        const float imp = impedance(pos, D0_DEF, DW_DEF, 1.0f / W_DEF);
        const float Rn = fmaxf((1.f - imp) * rcp(imp) * INVW_DOF[j], 1e-15f);
        const float aref = -B_DEF * (lower ? 1.f : -1.f) * C.qd[j] - K_DEF * imp * pos;
        rec(9 + j, mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 0.f), mk(0.f, 0.f, 0.f), &aref, 1, Rn, 1.f, 0.f, 0.f, 0.f, 0.f, C.lim_act[j], lower ? 1.f : -1.f);
    }
    float *g = stage + COOP_NB * COOP_REC;   // the factor of M (strictly lower part row by row, then 1 / L_ii), x, a0 of the arm
    int o = 0;
#pragma unroll
    for (int i = 1; i < 6; i++)
#pragma unroll
        for (int k = 0; k < i; k++) g[o++] = C.CL.L[i][k];
#pragma unroll
    for (int i = 0; i < 6; i++) g[15 + i] = C.CL.id[i];
#pragma unroll
    for (int j = 0; j < 6; j++) { g[21 + j] = y[j]; g[33 + j] = C.y0s[j]; }
    g[27] = ca[0].x; g[28] = ca[0].y; g[29] = ca[0].z; g[30] = cal[0].x; g[31] = cal[0].y; g[32] = cal[0].z;
}

// the solve; `lane` = this lane, L = the patient's lane (wave-uniform).  Returns the Newton iterations it took; the patient's y / ca / cal and the forces of its slots are updated.
template <int NC, int NRW, int NCC>
DEV int coop_solve(NewtonCtx<NC, NRW, false, NCC> &C, float *stage, int lane, int L, float (&y)[6], f3 (&ca)[NC], f3 (&cal)[NC]) {
    static_assert(NC == 1, "one cube");
    constexpr int NX = COOP_NX, NH = COOP_NH;
    const NewtonParams &P = C.P;
    if (lane == L) coop_stage<NC, NRW, NCC>(C, stage, y, ca, cal);
    // ---- every lane: its block's record, the shared vectors ----
    const int b = lane < COOP_NB ? lane : COOP_NB - 1;   // (lanes 15..63 shadow the last block with act = false)
    const float *r = stage + b * COOP_REC;
    const f3 dn = mk(r[0], r[1], r[2]), dt1 = mk(r[3], r[4], r[5]), dt2 = mk(r[6], r[7], r[8]), rc = mk(r[9], r[10], r[11]);
    float aref[6];
#pragma unroll
    for (int k = 0; k < 6; k++) aref[k] = r[12 + k];
    const float Rn = r[18], Rt = r[19], coef = r[23], lsign = r[25];
    const float m2[6] = {1.f, r[20], r[20], r[21], r[22], r[22]};
    const bool act = lane < COOP_NB && r[24] != 0.f;
    const float *gs = stage + COOP_NB * COOP_REC;
    float x[NX], x0[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) x[i] = gs[21 + i];
#pragma unroll
    for (int i = 0; i < 6; i++) { x0[i] = gs[33 + i]; x0[6 + i] = i == 2 ? -GRAV : 0.f; }
    const float cm = P.cube_mass, ci = rcp(P.cube_iinv);
    auto mdiag = [&](int i) -> float { return i < 6 ? 1.f : (i < 9 ? cm : ci); };
    // ---- the rows of this lane's block, dense in the 12 unknowns ----
    float J[6][NX];
#pragma unroll
    for (int q = 0; q < 6; q++)
#pragma unroll
        for (int i = 0; i < NX; i++) J[q][i] = 0.f;
    auto ldrow = [&](int lrow, float (&o6)[6]) {   // g row `lrow` of the patient's LDS column
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const float2v gp = *reinterpret_cast<const float2v *>(&C.lds[lrow * LDS_ROW + k * 128 + L * 2]);
            o6[2 * k] = gp.x; o6[2 * k + 1] = gp.y;
        }
    };
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
        } else if (b >= 9) {   // a joint limit: row L^-1 (+-e_j)
            const int j = b - 9;
            float g6[6], Ls[15], id[6];
#pragma unroll
            for (int k = 0; k < 15; k++) Ls[k] = gs[k];
#pragma unroll
            for (int k = 0; k < 6; k++) { id[k] = gs[15 + k]; g6[k] = k == j ? lsign : 0.f; }
#pragma unroll
            for (int i = 0; i < 6; i++) {   // forward substitution (fsub)
                float s = g6[i];
#pragma unroll
                for (int k = 0; k < i; k++) s = fmaf(-Ls[i * (i - 1) / 2 + k], g6[k], s);
                g6[i] = s * id[i];
            }
#pragma unroll
            for (int k = 0; k < 6; k++) J[0][k] = g6[k];
        }
        // the cube's share: its contact point moves with ca + cal x rc (rows 0-2); rows 3-5 see cal
#pragma unroll
        for (int q = 0; q < 6; q++) {
            const f3 d = (q % 3) == 0 ? dn : ((q % 3) == 1 ? dt1 : dt2);
            const f3 lin = q < 3 ? coef * d : mk(0.f, 0.f, 0.f), ang = q < 3 ? coef * cross(rc, d) : coef * d;
            J[q][6] = lin.x; J[q][7] = lin.y; J[q][8] = lin.z; J[q][9] = ang.x; J[q][10] = ang.y; J[q][11] = ang.z;
        }
    }
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
    float *shares = stage, *totals = stage + COOP_NB * COOP_RED;
    int its = 0;
    float dprev = 3.0e38f;
    for (int it = 0; it < P.newton_iters; it++) {
        float dx[NX], d0 = 0.f;
        {
            // this block's share of the gradient and of the Hessian
            BlkEval<6> B;
            blk_eval<6>(zs, Rn, Rt, m2, act, B);
            float Hl[NH], gl[NX];
#pragma unroll
            for (int i = 0; i < NH; i++) Hl[i] = 0.f;
#pragma unroll
            for (int i = 0; i < NX; i++) {
                float a = 0.f;
#pragma unroll
                for (int q = 0; q < 6; q++) a = fmaf(-B.f[q], J[q][i], a);
                gl[i] = a;
            }
            h_block<0, NX, NX, 6>(Hl, J, B, m2);
            if (lane < COOP_NB) {
                float *w = shares + lane * COOP_RED;
#pragma unroll
                for (int i = 0; i < NX; i++) w[i] = gl[i];
#pragma unroll
                for (int i = 0; i < NH; i++) w[NX + i] = Hl[i];
            }
            // column sums: lane e sums entry e (and entry 64 + e) over the blocks
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int k = 0; k < COOP_NB; k++) {
                t0 += shares[k * COOP_RED + lane];
                t1 += lane < COOP_RED - 64 ? shares[k * COOP_RED + 64 + lane] : 0.f;
            }
            totals[lane] = t0;
            if (lane < COOP_RED - 64) totals[64 + lane] = t1;
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
#pragma unroll
        for (int q = 0; q < 6; q++) {
            float a = 0.f;
#pragma unroll
            for (int i = 0; i < NX; i++) a = fmaf(J[q][i], dx[i], a);
            jd[q] = act ? a : 0.f;
        }
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
            const bool ok = act && lane < 9 && qa > 0.f && am > lo && am < hi && wn < 0.f && !(n2 * Rn * Rn > wn * wn * Rt * Rt);
            const float dist = fabsf(am - sec);
            float best = sec, bestd = -1.f;
            for (int k = 0; k < 9; k++) {
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
        if (lane < COOP_NB) {
#pragma unroll
            for (int q = 0; q < 6; q++) stage[lane * 8 + q] = B.f[q];
        }
        if (lane == L) {
#pragma unroll
            for (int s = 0; s < NAS; s++)
#pragma unroll
                for (int q = 0; q < (s < 4 ? 6 : 4); q++) C.AS[s].f[q] = stage[s * 8 + q];
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int q = 0; q < 4; q++) C.FS[0][s].f[q] = stage[(5 + s) * 8 + q];
#pragma unroll
            for (int j = 0; j < 6; j++) C.flim[j] = stage[(9 + j) * 8];
#pragma unroll
            for (int j = 0; j < 6; j++) y[j] = x[j];
            ca[0] = mk(x[6], x[7], x[8]); cal[0] = mk(x[9], x[10], x[11]);
        }
    }
    C.wave_its += its;
    return its;
}

}  // namespace
