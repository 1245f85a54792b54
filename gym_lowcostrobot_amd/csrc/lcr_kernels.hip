// lcr_kernels.hip -- hand-written gfx950 kernels of the batched low-cost-robot simulator.
//
// Mapping: ONE wavefront lane == ONE environment.  A workgroup is one wave (64 lanes), so nothing in the
// step kernel needs a barrier; 65 536 envs = 1024 waves = one wave per SIMD of the 256 CUs.
// State is SoA [component][env] in HBM: every per-lane scalar load/store is one fully coalesced 256-B
// wave transaction.  State is read once at kernel entry, kept in VGPRs across all n_substeps physics
// substeps, and written once at exit; reward / termination / TimeLimit / auto-reset are fused at the tail.
// LDS holds the per-lane contact rows that couple into the arm (g = L^-1 J^T, 5 slots x 4..6 rows x 6 floats, laid out
// [slot][row][k][lane] => bank-conflict free), the parked constants of the floor slots and, for Stack, the cube<->cube
// contact records: 38-52 KiB per wave (LdsSize).  MFMA is not used: the largest contraction is 6x6.
// Template variants of the step kernel: NC cubes (1|2), EE (ee-IK action mode), ADAPT (converged
// solver mode), ROLL (six-row finger<->cube contacts), BIG (Stack shards of <= 3 waves per CU: every row in LDS).
//
// What is restated here (reference file:line, relative to /root/reference/gym_lowcostrobot/):
//   apply_action joint mode   envs/reach_cube_env.py:248-273 (+ lift_cube_env.py:258-282 gripper)
//   apply_action ee mode + IK envs/reach_cube_env.py:236-247, 148-221 (incl. the qpos overwrite)
//   20 x mujoco.mj_step       envs/reach_cube_env.py:276-279 -> substep() below; the MuJoCo pipeline itself
//                             (CRBA+armature, RNE, position actuators, soft contacts, implicitfast) follows
//                             MuJoCo's public documentation; constants from assets/low_cost_robot_6dof/*.xml
//   reward / success / done   envs/reach_cube_env.py:313-348, lift:322-346, push:330-361, pick_place:338-369,
//                             stack_two_cubes_env.py:326-363; TimeLimit(50) from __init__.py:9-43
//   reset                     envs/reach_cube_env.py:297-311, push:308-328, pick_place:316-336, stack:307-324
#include "lcr_step_common.h"
#include "lcr_newton.h"
#include "lcr_newton_coop.h"

// translation-unit selection, see the launchers at the end of the file
#ifndef LCR_PART
#define LCR_PART (-1)
#endif
#define LCR_HAS_PART(k) (LCR_PART == -1 || LCR_PART == (k))

namespace {

// ------------------------------------------------------------------------------------------------
// one physics substep (== mujoco.mj_step, reach_cube_env.py:277)
// ------------------------------------------------------------------------------------------------
// NEWTON (the faithful preset): every finger contact has MuJoCo's six rows -- against the floor too (follower.xml:15 condim="6") -- and the constraint problem is solved
// by Newton's method on the primal (lcr_newton.h) instead of sweeps on the dual
// CPL (Newton kernels): the substep exists in two copies so that the coupled problems (arm + cube: 12 unknowns, 78-entry Hessian; Stack's three bodies: 18 unknowns,
// 171 entries) do not set the register allocation of the substeps that do not need them.  CPL_FAST: every body is its own 6-dimensional SIMT problem; the envs of
// the wave in which bodies touch (a finger or a gripper-body proxy on a cube, cube on cube) are solved one by one by the whole wave (lcr_newton_coop.h).  With more
// than LcrDev::coop_max such envs the copy returns false -- the env state untouched -- and the substep is run by CPL_SLOW: the wave-uniform coupled SIMT solves of
// round 5; it returns whether the wave still has that many (the caller goes back to the fast copy when not).  CPL_BOTH: one copy with both (PushCubeLoop).
constexpr int CPL_BOTH = 0, CPL_FAST = 1, CPL_SLOW = 2;
#ifndef LCR_COOP_ROWS
#define LCR_COOP_ROWS 4   // one cube: patients per pass of the cooperative solve, one per 16-lane row (0: one at a time through the LDS reduction, as Stack and PushCubeLoop do)
#endif
template <int NC, bool ADAPT, bool ROLL, bool BIG, bool NEWTON = false, int CPL = CPL_BOTH>
DEV bool substep(const LcrDev &P, EnvState<NC> &S, const float (&ctrl)[6], float *lds, int lane, int env, f3 &lag_ee, f3 (&lag_cube)[NC], Warm<NC, ROLL ? 6 : 4> &W, Diag &DGtot, int sub_index) {
    static_assert(!NEWTON || (ROLL && !ADAPT && (NC == 1 || BIG)), "the Newton kernels carry six-row finger slots and keep every g row in LDS");
    static_assert(CPL == CPL_BOTH || NEWTON, "two copies of the substep: the Newton kernels");
    constexpr int NCC = NEWTON ? 8 : 4;   // cube<->cube manifold points (Stack): the Newton kernels carry eight slots, 4-7 in use with lcr_config.cc_points = 8
    constexpr int CCB = NEWTON ? NEWTON_G_ROWS : cc_base_rows<NC, BIG, ROLL>();   // first LDS row of the cube<->cube records
    constexpr int CCR = NEWTON ? CC_REC_NEWTON : CC_REC, CCRN = NEWTON ? CC_RN_NEWTON : 15;   // floats of a cube<->cube record, index of its Rn
    constexpr int NRW = ROLL ? 6 : 4;   // rows an arm slot may have
    Diag DG = {0u, 0u, 0u, 0u};   // this substep's share
    using namespace lcrm;
    // ---- position stage -------------------------------------------------------------------------
    CubeRot CR[NC];
    float cq_in[NC][4];   // (CPL_FAST) the quaternion as it came in: a bail-out leaves the state as it found it
#pragma unroll
    for (int c = 0; c < NC; c++) {
#pragma unroll
        for (int k = 0; k < 4; k++) cq_in[c][k] = S.cq[c][k];
        float n2 = S.cq[c][0] * S.cq[c][0] + S.cq[c][1] * S.cq[c][1] + S.cq[c][2] * S.cq[c][2] + S.cq[c][3] * S.cq[c][3];
        float in = rsq(n2);
#pragma unroll
        for (int k = 0; k < 4; k++) S.cq[c][k] *= in;
        CR[c] = quat_to_cols(S.cq[c]);
        lag_cube[c] = S.cp[c];  // P8: body xpos as left behind by this mj_step
    }
    ArmFrames F;
    arm_frames(S.q, F);
    lag_ee = site_pos(F);
    f3 z[6], v0[6];  // joint axes and p x z (linear velocity of the world origin per unit joint rate)
#pragma unroll
    for (int j = 0; j < 6; j++) { z[j] = joint_axis(F, j); v0[j] = cross(F.p[j], z[j]); }

    // link coms and world inertias
    f3 com[6];
    Sym3 Iw[6];
    com[0] = local_point(F, 0, C1x, C1y, C1z); Iw[0] = world_inertia(F.X[0], F.Y[0], F.Z[0], I1_xx, I1_xy, I1_xz, I1_yy, I1_yz, I1_zz);
    com[1] = local_point(F, 1, C2x, C2y, C2z); Iw[1] = world_inertia(F.X[1], F.Y[1], F.Z[1], I2_xx, I2_xy, I2_xz, I2_yy, I2_yz, I2_zz);
    com[2] = local_point(F, 2, C3x, C3y, C3z); Iw[2] = world_inertia(F.X[2], F.Y[2], F.Z[2], I3_xx, I3_xy, I3_xz, I3_yy, I3_yz, I3_zz);
    com[3] = local_point(F, 3, C4x, C4y, C4z); Iw[3] = world_inertia(F.X[3], F.Y[3], F.Z[3], I4_xx, I4_xy, I4_xz, I4_yy, I4_yz, I4_zz);
    com[4] = local_point(F, 4, C5x, C5y, C5z); Iw[4] = world_inertia(F.X[4], F.Y[4], F.Z[4], I5_xx, I5_xy, I5_xz, I5_yy, I5_yz, I5_zz);
    com[5] = local_point(F, 5, C6x, C6y, C6z); Iw[5] = world_inertia(F.X[5], F.Y[5], F.Z[5], I6_xx, I6_xy, I6_xz, I6_yy, I6_yz, I6_zz);
    const float mass[6] = {M1, M2, M3, M4, M5, M6};

    // ---- joint-space inertia: composite rigid bodies referenced to the WORLD ORIGIN ----------------
    // composite (m, h = sum m c, I_O = sum I_w + m(|c|^2 1 - c c^T)); column i of M from the composite of links i..6
    float Mm[6][6];
    {
        float mc = 0.f;
        f3 hc = mk(0.f, 0.f, 0.f);
        Sym3 Ic = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            const float m = mass[i];
            const f3 c = com[i];
            const float cc = dot(c, c);
            mc += m;
            hc = axpy(m, c, hc);
            Ic.xx += Iw[i].xx + m * (cc - c.x * c.x);
            Ic.yy += Iw[i].yy + m * (cc - c.y * c.y);
            Ic.zz += Iw[i].zz + m * (cc - c.z * c.z);
            Ic.xy += Iw[i].xy - m * c.x * c.y;
            Ic.xz += Iw[i].xz - m * c.x * c.z;
            Ic.yz += Iw[i].yz - m * c.y * c.z;
            // spatial momentum of the composite for unit rate of joint i
            f3 l = axpy(mc, v0[i], cross(z[i], hc));
            f3 n = symv(Ic, z[i]) + cross(hc, v0[i]);
#pragma unroll
            for (int j = 0; j <= i; j++) Mm[i][j] = dot(z[j], n) + dot(v0[j], l);
            Mm[i][i] += ARMATURE;
        }
    }
    Chol6 CL;
    chol6(Mm, CL);

    // ---- bias forces: recursive Newton-Euler, zero joint acceleration, base accelerating at -g ------
    float tau[6];
    {
        f3 w = mk(0.f, 0.f, 0.f), wd = mk(0.f, 0.f, 0.f), a = mk(0.f, 0.f, GRAV), pprev = mk(0.f, 0.f, 0.f);
        f3 Fi[6], Ni[6];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            f3 r = F.p[i] - pprev;
            a = a + cross(wd, r) + wxwxr(w, r, dot(w, w));  // acceleration of this link's origin (rigid with the parent)
            f3 zq = S.qd[i] * z[i];
            wd = wd + cross(w, zq);
            w = w + zq;
            f3 rc = com[i] - F.p[i];
            f3 ac = a + cross(wd, rc) + wxwxr(w, rc, dot(w, w));
            Fi[i] = mass[i] * ac;
            Ni[i] = symv(Iw[i], wd) + cross(w, symv(Iw[i], w));
            pprev = F.p[i];
        }
        f3 f = mk(0.f, 0.f, 0.f), n = mk(0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 5; i >= 0; i--) {
            f3 nn = Ni[i] + cross(com[i] - F.p[i], Fi[i]);
            if (i < 5) nn = nn + n + cross(F.p[i + 1] - F.p[i], f);
            f = f + Fi[i];
            n = nn;
            float bias = dot(z[i], n);
            // passive damping + position actuator (ctrlrange == joint range via inheritrange; joint-level force clamp)
            float c = clampf(ctrl[i], JLO[i], JHI[i]);
            float fa = clampf(fmaf(KP, c - S.q[i], -KV * S.qd[i]), -FRC, FRC);
            tau[i] = fa - DAMPING * S.qd[i] - bias;
        }
    }
    // y = L^T a  (scaled arm acceleration);  y_smooth = L^-1 tau
    float y[6];
#pragma unroll
    for (int j = 0; j < 6; j++) y[j] = tau[j];
    fsub(CL, y);
    float y0s[6];   // (NEWTON) the unconstrained arm acceleration in y coordinates: a0 of the primal problem
#pragma unroll
    for (int j = 0; j < 6; j++) y0s[j] = y[j];
    // cube accelerations, world frame (isotropic inertia: no gyroscopic term)
    f3 ca[NC], cal[NC];
    f3 cww[NC];  // cube angular velocity in world frame
#pragma unroll
    for (int c = 0; c < NC; c++) {
        ca[c] = mk(0.f, 0.f, -GRAV);
        cal[c] = mk(0.f, 0.f, 0.f);
        cww[c] = axpy(S.cw[c].x, CR[c].X, axpy(S.cw[c].y, CR[c].Y, S.cw[c].z * CR[c].Z));
    }
    const float minv = P.cube_minv, iinv = P.cube_iinv;

    // ---- collision: floor <-> cube (MuJoCo plane-box: penetrating vertices in index order, at most 4) ----
    FloorSlot FS[NC][4];
#pragma unroll
    for (int c = 0; c < NC; c++) {
#pragma unroll
        for (int s = 0; s < 4; s++) { FS[c][s].act = false; FS[c][s].r = mk(0.f, 0.f, 0.f); FS[c][s].Rn = 1.f;
#pragma unroll
            for (int k = 0; k < 4; k++) { FS[c][s].f[k] = 0.f; FS[c][s].aref[k] = 0.f; FS[c][s].inv[k] = 0.f; } }
        float sdist[4] = {0.f, 0.f, 0.f, 0.f};
        // vertex i = (+-h, +-h, +-h) by bits 0,1,2 of i, relative to the cube centre
        const f3 hx = CH * CR[c].X, hy = CH * CR[c].Y, hz = CH * CR[c].Z;
        f3 rv[8];
        float vd[8];
        bool lower_same = true, upper_none = true;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            f3 a = (i & 1) ? hx : neg(hx), b = (i & 2) ? hy : neg(hy), d = (i & 4) ? hz : neg(hz);
            rv[i] = a + b + d;
            vd[i] = S.cp[c].z + rv[i].z;
            if (i >= 4) upper_none = upper_none && !(vd[i] < 0.f);
            else if (i > 0) lower_same = lower_same && ((vd[i] < 0.f) == (vd[0] < 0.f));
        }
        if (__all(lower_same && upper_none)) {
            // common case (upright cubes resting or airborne in every lane): the penetrating set is exactly {0,1,2,3} or
            // empty, so slot s holds vertex s -- the same assignment as the general rule, without the select chains
#pragma unroll
            for (int s = 0; s < 4; s++) {
                FS[c][s].r = mk(rv[s].x, rv[s].y, rv[s].z - 0.5f * vd[s]);
                sdist[s] = vd[s];
                FS[c][s].act = vd[s] < 0.f;
                if (P.diag) diag_choice(DG, FS[c][s].act, 4 * c + s, s);
            }
        } else {
            int cnt = 0;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float dist = vd[i];
                bool pen = dist < 0.f && cnt < 4;
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    bool take = pen && cnt == s;
                    FS[c][s].r.x = take ? rv[i].x : FS[c][s].r.x;
                    FS[c][s].r.y = take ? rv[i].y : FS[c][s].r.y;
                    FS[c][s].r.z = take ? rv[i].z - 0.5f * dist : FS[c][s].r.z;  // contact point midway between vertex and plane
                    sdist[s] = take ? dist : sdist[s];
                    FS[c][s].act = FS[c][s].act || take;
                    if (P.diag) diag_choice(DG, take, 4 * c + s, i);
                }
                cnt += pen ? 1 : 0;
            }
        }
#pragma unroll
        for (int s = 0; s < 4; s++) {
            FloorSlot &T = FS[c][s];
            float imp = impedance(sdist[s], D0_DEF, DW_DEF, 1.0f / W_DEF);
            float Rn = fmaxf((1.f - imp) * rcp(imp) * minv, 1e-15f);  // diagApprox = cube body_invweight0 (translational)
            float Rf = Rn * P.inv_impratio;                          // elliptic cone: friction rows R/impratio
            float Rt = Rf * P.rt_cube;
            T.Rn = Rn;
            f3 vp = S.cv[c] + cross(cww[c], T.r);  // velocity of the contact point
            T.aref[0] = -B_DEF * vp.z - K_DEF * imp * sdist[s];
            T.aref[1] = -B_DEF * vp.y;
            T.aref[2] = B_DEF * vp.x;       // t2 = -x
            T.aref[3] = -B_DEF * cww[c].z;  // torsion about n
            // (an inactive slot gets inv = 0: with f = 0 its row updates then come out as exactly zero in the sweeps, no per-sweep selects)
            {   // k[] of the block step (lcr_step_common.h soc_step): Ln = 2 (A + R)_nn, Lt = 2 (mu^2 ((A + R)_11 + (A + R)_22) + mu_tors^2 (A + R)_33)
                const float KF = 2.f;   // (two groups, see soc_step)
                const float Ln = KF * (minv + iinv * (T.r.x * T.r.x + T.r.y * T.r.y) + Rn);
                const float a12 = 2.f * minv + iinv * (2.f * T.r.z * T.r.z + T.r.x * T.r.x + T.r.y * T.r.y) + 2.f * Rf;
                const float Ls = KF * P.mu_ct2 * (iinv + Rt);
                const float Lt = fmaf(KF * P.mu_c2, a12, Ls);
                const float iLt = T.act ? rcp(Lt) : 0.f, iLs = iLt;
                T.inv[0] = T.act ? rcp(Ln) : 0.f; T.inv[1] = P.mu_c2 * iLt; T.inv[2] = Ln * rcp(Ln + Lt); T.inv[3] = P.mu_ct2 * iLs;
            }
            if (c == 0 && NC == 1 && !ROLL) {
                float4v *pk = reinterpret_cast<float4v *>(lds + LDS_G_FLOATS) + (size_t)(s * 2) * 64 + lane;
                pk[0] = float4v{T.aref[0], T.aref[1], T.aref[2], T.aref[3]};
                pk[64] = float4v{T.inv[0], T.inv[1], T.inv[2], T.inv[3]};
            }
            // warm start: forces of the previous substep if this slot was active then (inactive slots were zeroed)
#pragma unroll
            for (int k = 0; k < 4; k++) { T.f[k] = T.act ? W.floor[c][s][k] : 0.f; }
            const f3 r = T.r;
            ca[c].z = fmaf(minv, T.f[0], ca[c].z);
            ca[c].y = fmaf(minv, T.f[1], ca[c].y);
            ca[c].x = fmaf(-minv, T.f[2], ca[c].x);
            cal[c].x = fmaf(iinv, r.y * T.f[0] - r.z * T.f[1], cal[c].x);
            cal[c].y = fmaf(iinv, -r.x * T.f[0] - r.z * T.f[2], cal[c].y);
            cal[c].z = fmaf(iinv, r.x * T.f[1] + r.y * T.f[2] + T.f[3], cal[c].z);
        }
    }


    // ---- collision: cube <-> cube (Stack).  Face-axis SAT (6 axes), then vertices of each box below the other's
    //      reference face and inside its footprint; first 4 found (box1's vertices first).  Records live in LDS. ----
    bool cc_act[NCC];
#pragma unroll
    for (int s = 0; s < NCC; s++) cc_act[s] = false;
    bool cc_any = false;
    const bool real_lane = (int)(blockIdx.x * 64 + lane) < P.n;   // (the tail lanes of a ragged batch shadow the last env and store nothing: they are nobody's patient)
    int cc_count = 0;       // (Stack) cube<->cube points of this lane
    bool coupled = false;   // (NEWTON) bodies of this lane's env touch -- its arm a cube, cube on cube --: they are ONE problem
    f3 ccn = mk(0.f, 0.f, 1.f), cct1 = mk(0.f, 1.f, 0.f), cct2 = mk(-1.f, 0.f, 0.f);
    float *ccl = lds + CCB * LDS_ROW + lane;   // Stack: record field k of slot s at ccl[(s*CCR + k)*64]
    const size_t CS = 64;
    if constexpr (NC == 2) {
        const f3 dc = S.cp[1] - S.cp[0];
        const f3 ax0[3] = {CR[0].X, CR[0].Y, CR[0].Z}, ax1[3] = {CR[1].X, CR[1].Y, CR[1].Z};
        float best = -1e30f, bsgn = 1.f;
        int bax = 0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
            const f3 n = a < 3 ? ax0[a] : ax1[a - 3];
            float ext = 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) ext += fabsf(dot(n, a < 3 ? ax1[j] : ax0[j])) * CH;
            float dd = dot(n, dc);
            float sep = fabsf(dd) - CH - ext;
            if (sep > best) { best = sep; bax = a; bsgn = dd < 0.f ? -1.f : 1.f; }
        }
        const bool touching = best < 0.f;
        int cnt = 0;
        f3 cpos[NCC];
        float cdist[NCC];
#pragma unroll
        for (int s = 0; s < NCC; s++) { cpos[s] = mk(0.f, 0.f, 0.f); cdist[s] = 0.f; }
        if (__any(touching)) {  // wave-uniform: the manifold construction is skipped unless some env has overlapping cubes
            // reference box A (owner of the best axis), incident box B; per-lane selection by compare/select
            const bool Ais0 = bax < 3;
            const int k = Ais0 ? bax : bax - 3;
            auto sel3 = [](int i, f3 a, f3 b, f3 c) { return i == 0 ? a : (i == 1 ? b : c); };
            const f3 AX = Ais0 ? CR[0].X : CR[1].X, AY = Ais0 ? CR[0].Y : CR[1].Y, AZ = Ais0 ? CR[0].Z : CR[1].Z;
            const f3 BX = Ais0 ? CR[1].X : CR[0].X, BY = Ais0 ? CR[1].Y : CR[0].Y, BZ = Ais0 ? CR[1].Z : CR[0].Z;
            const f3 cA = Ais0 ? S.cp[0] : S.cp[1], cB = Ais0 ? S.cp[1] : S.cp[0];
            ccn = bsgn * sel3(k, AX, AY, AZ);                 // cube0 -> cube1
            const f3 m = Ais0 ? ccn : neg(ccn);               // A -> B
            const f3 u = sel3(k, AY, AZ, AX), v = sel3(k, AZ, AX, AY);   // columns (k+1)%3, (k+2)%3
            const float md0 = dot(m, BX), md1 = dot(m, BY), md2 = dot(m, BZ);
            int kb = 0; float bd = fabsf(md0);
            if (fabsf(md1) > bd) { bd = fabsf(md1); kb = 1; }
            if (fabsf(md2) > bd) { bd = fabsf(md2); kb = 2; }
            const float mdk = kb == 0 ? md0 : (kb == 1 ? md1 : md2);
            const float sB = mdk > 0.f ? -1.f : 1.f;
            const f3 nb = sB * sel3(kb, BX, BY, BZ);
            const f3 pa = sel3(kb, BY, BZ, BX), qa = sel3(kb, BZ, BX, BY);
            const f3 fB = axpy(CH, nb, cB);
            const float mnb = dot(m, nb);
            const float tol = 1e-4f;
            const float SP[4] = {1.f, -1.f, -1.f, 1.f}, SQ[4] = {1.f, 1.f, -1.f, -1.f};
            f3 V[4];
            float Vu[4], Vv[4], Vd[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                V[i] = axpy(CH * SP[i], pa, axpy(CH * SQ[i], qa, fB));
                f3 d = V[i] - cA;
                Vu[i] = dot(d, u); Vv[i] = dot(d, v); Vd[i] = dot(d, m) - CH;
            }
            float skey[NCC];
            int sidx[NCC];
#pragma unroll
            for (int s = 0; s < NCC; s++) { skey[s] = 0.f; sidx[s] = -1; }
            const bool eight = NCC == 8 && P.cc8;   // (the axis extremes join the diagonal ones: lcr_config.cc_points = 8)
            auto consider = [&](bool ok, int cand, f3 P, float dist, float cu, float cv) {
                const float key[8] = {cu + cv, -cu + cv, -cu - cv, cu - cv, cu, cv, -cu, -cv};   // diagonals of the reference face, then its axes
#pragma unroll
                for (int s = 0; s < NCC; s++) {
                    bool t = ok && (s < 4 || eight) && (sidx[s] < 0 || key[s] > skey[s]);
                    skey[s] = t ? key[s] : skey[s];
                    sidx[s] = t ? cand : sidx[s];
                    cdist[s] = t ? dist : cdist[s];
                    cpos[s].x = t ? P.x : cpos[s].x; cpos[s].y = t ? P.y : cpos[s].y; cpos[s].z = t ? P.z : cpos[s].z;
                }
            };
            // (a) vertices of B's incident face inside A's footprint, below A's face
#pragma unroll
            for (int i = 0; i < 4; i++) {
                bool ok = touching && !(fabsf(Vu[i]) > CH + tol || fabsf(Vv[i]) > CH + tol) && Vd[i] < 0.f;
                consider(ok, i, axpy(-0.5f * Vd[i], m, V[i]), Vd[i], Vu[i], Vv[i]);
            }
            // (b) vertices of A's face inside B's incident face (near-parallel faces only)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                f3 a = axpy(CH, m, axpy(CH * SP[j], u, axpy(CH * SQ[j], v, cA)));
                f3 d = a - fB;
                float t = -dot(d, nb) * rcp(mnb);
                bool ok = touching && mnb < -0.5f && !(fabsf(dot(d, pa)) > CH + tol || fabsf(dot(d, qa)) > CH + tol) && t < 0.f;
                consider(ok, 4 + j, axpy(0.5f * t, m, a), t, SP[j] * CH, SQ[j] * CH);
            }
            // (c) crossings of B's face edges with A's face boundary lines
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int e2 = (e + 1) & 3;
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    const bool on_u = l < 2;
                    const float sg = (l & 1) ? -1.f : 1.f;
                    const float cP = on_u ? Vu[e] : Vv[e], cQ = on_u ? Vu[e2] : Vv[e2];
                    const float oP = on_u ? Vv[e] : Vu[e], oQ = on_u ? Vv[e2] : Vu[e2];
                    const float fP = cP - sg * CH, fQ = cQ - sg * CH;
                    const bool cross_ = (fP < 0.f && fQ > 0.f) || (fP > 0.f && fQ < 0.f);
                    const float t = fP * rcp(cross_ ? fP - fQ : 1.f);
                    const float ot = fmaf(t, oQ - oP, oP);
                    const float d = fmaf(t, Vd[e2] - Vd[e], Vd[e]);
                    bool ok = touching && cross_ && !(fabsf(ot) > CH + tol) && d < 0.f;
                    f3 X = axpy(t, V[e2] - V[e], V[e]);
                    consider(ok, 8 + 4 * e + l, axpy(-0.5f * d, m, X), d, on_u ? sg * CH : ot, on_u ? ot : sg * CH);
                }
            }
#pragma unroll
            for (int s = 0; s < NCC; s++) {
                bool dup = false;
#pragma unroll
                for (int s2 = 0; s2 < s; s2++) dup = dup || (sidx[s2] == sidx[s]);
                cc_act[s] = sidx[s] >= 0 && !dup;
                cnt += cc_act[s] ? 1 : 0;
                if (P.diag) diag_choice(DG, cc_act[s], s < 4 ? 8 + s : 24 + (s - 4), sidx[s] + 32 * (bax + 6 * kb) + 1024 * (bsgn < 0.f ? 1 : 0));
            }
        }
        cc_any = __any(cnt > 0) != 0;
        coupled = cnt > 0 && real_lane;
        cc_count = cnt;
        if (cc_any) {
            make_frame(ccn, cct1, cct2);
#pragma unroll
            for (int s = 0; s < NCC; s++) {
                const f3 r0 = cpos[s] - S.cp[0], r1 = cpos[s] - S.cp[1];
                float imp = impedance(cdist[s], D0_DEF, DW_DEF, 1.0f / W_DEF);
                float Rn = fmaxf((1.f - imp) * rcp(imp) * (2.f * minv), 1e-15f);
                float Rf = Rn * P.inv_impratio;
                float Rt = Rf * P.rt_cube;
                f3 vrel = (S.cv[1] + cross(cww[1], r1)) - (S.cv[0] + cross(cww[0], r0));
                f3 wrel = cww[1] - cww[0];
                ccl[(size_t)(s * CCR + 0) * CS] = cpos[s].x; ccl[(size_t)(s * CCR + 1) * CS] = cpos[s].y; ccl[(size_t)(s * CCR + 2) * CS] = cpos[s].z;
                float ccLn = 1.f, ccLt = 0.f;   // metric of the block step (soc_step)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const f3 d = r == 0 ? ccn : (r == 1 ? cct1 : (r == 2 ? cct2 : ccn));
                    float vel = r < 3 ? dot(d, vrel) : dot(d, wrel);
                    float aref = -B_DEF * vel - (r == 0 ? K_DEF * imp * cdist[s] : 0.f);
                    float diag;
                    if (r < 3) { f3 a0 = cross(r0, d), a1 = cross(r1, d); diag = 2.f * minv + iinv * (dot(a0, a0) + dot(a1, a1)); }
                    else diag = 2.f * iinv;
                    float Rr = r == 0 ? Rn : (r == 3 ? Rt : Rf);
                    {   // f: keep the previous substep's force if the slot was active then, and apply it
                        float fw = (cc_act[s] && W.cc_prev[s]) ? ccl[(size_t)(s * CCR + 3 + r) * CS] : 0.f;
                        ccl[(size_t)(s * CCR + 3 + r) * CS] = fw;
                        if (r < 3) {
                            f3 a0 = cross(r0, d), a1 = cross(r1, d);
                            ca[1] = axpy(minv * fw, d, ca[1]); ca[0] = axpy(-minv * fw, d, ca[0]);
                            cal[1] = axpy(iinv * fw, a1, cal[1]); cal[0] = axpy(-iinv * fw, a0, cal[0]);
                        } else { cal[1] = axpy(iinv * fw, d, cal[1]); cal[0] = axpy(-iinv * fw, d, cal[0]); }
                    }
                    ccl[(size_t)(s * CCR + 7 + r) * CS] = aref;
                    if (r == 0) ccLn = 2.f * (diag + Rr); else ccLt = fmaf(2.f * (r == 3 ? P.mu_ct2 : P.mu_c2), diag + Rr, ccLt);
                }
                if constexpr (!NEWTON) {   // k[] of soc_step in the record's four "inverse diagonal" fields
                    const float iLt = cc_act[s] ? rcp(ccLt) : 0.f;
                    ccl[(size_t)(s * CCR + 11) * CS] = cc_act[s] ? rcp(ccLn) : 0.f;
                    ccl[(size_t)(s * CCR + 12) * CS] = P.mu_c2 * iLt;
                    ccl[(size_t)(s * CCR + 13) * CS] = ccLn * rcp(ccLn + ccLt);
                    ccl[(size_t)(s * CCR + 14) * CS] = P.mu_ct2 * iLt;
                }
                ccl[(size_t)(s * CCR + CCRN) * CS] = Rn;
            }
        }
    }

    // ---- collision: arm-coupled contact slots (NAS): finger spheres vs cube / floor, arm-link proxies (D3); rows g = L^-1 J^T
    //      go to LDS.  Every slot is skipped wave-uniformly when no lane of the wave touches. ----
    ArmSlot<NRW> AS[NAS];
    bool slot_any[NAS];
    bool link_on_cube = false;       // slot 4: this lane's contact is against a cube (else the floor)
    int link_nj = 3;             // slot 4: number of joints that move the contact point (proxy on link_3: 3 ... link_6: 6)
    int link_bi = 0;             // slot 4: which proxy
    int slot_cube[3] = {0, 0, 0};  // which cube the cube slots 0, 1 and 4 refer to (Stack)
    {
    const f3 sph[2] = {local_point(F, 4, SPH0x, SPH0y, SPH0z), local_point(F, 5, SPH1x, SPH1y, SPH1z)};
    const float srad[2] = {SPH0r, SPH1r};
    // (NEWTON = the faithful preset) the finger pads as boxes: centre, link axes, half extents (lcr_step_common.h pad_box / pad_floor; oracle finger_geom = 1)
    const f3 padc[2] = {local_point(F, 4, PAD0cx, PAD0cy, PAD0cz), local_point(F, 5, PAD1cx, PAD1cy, PAD1cz)};
    const f3 padh[2] = {mk(PAD0hx, PAD0hy, PAD0hz), mk(PAD1hx, PAD1hy, PAD1hz)};
    const float padr[2] = {0.01293f, 0.01369f};   // |half extents|: bounding radius for the broad phase
    bool body_done[2] = {false, false};   // (NEWTON) the angular rows of finger body 0 / 1 are in LDS (wave-uniform)
#pragma unroll
    for (int s = 0; s < NAS; s++) {
        const int sp = s & 1;
        const bool may_cube = s < 2 || s == 4;   // literal after unrolling
        ArmSlot<NRW> &T = AS[s];
        f3 pos = mk(0.f, 0.f, 0.f), n = mk(0.f, 0.f, 1.f);
        float dist = 1.f;
        int cidx = 0;
        bool oncube = s < 2;
        float invw_link = sp == 0 ? INVW_TRAN_L5 : INVW_TRAN_L6;
        int sel = 0;   // discrete choice behind the contact (diagnostics)
        if (s < 2) {
            // sphere vs box; broad phase (wave-uniform): a sphere farther than r + h*sqrt(3) from every cube centre cannot touch
            float bestd = 1e30f;
            bool near_any = false;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const f3 dd = (NEWTON ? padc[sp] : sph[sp]) - S.cp[c];
                const float reach_ = (NEWTON ? padr[sp] : srad[sp]) + 1.7321f * CH;
                near_any = near_any || dot(dd, dd) < reach_ * reach_;
            }
            if (__any(near_any))
#pragma unroll
            for (int c = 0; c < NC; c++) {
                if constexpr (NEWTON) {
                    const int L = sp == 0 ? 4 : 5;
                    const SBHit hit = pad_box(padc[sp], F.X[L], F.Y[L], F.Z[L], padh[sp], S.cp[c], CR[c]);
                    if (hit.dist < bestd) { bestd = hit.dist; cidx = c; n = hit.n; pos = hit.pos; sel = c + 4 * hit.code; }  // deepest cube wins (tie: cube 0)
                } else {
                    const SBHit hit = sphere_box(sph[sp], srad[sp], S.cp[c], CR[c]);
                    if (hit.dist < bestd) { bestd = hit.dist; cidx = c; n = hit.n; pos = hit.pos; sel = 8 * c + hit.code; }  // deepest cube wins (tie: cube 0)
                }
            }
            dist = bestd;
            slot_cube[sp] = cidx;
        } else if (s < 4) {
            if constexpr (NEWTON) {
                const int L = sp == 0 ? 4 : 5;
                const PadFloorHit hit = pad_floor<false>(PadBox{padc[sp], padh[sp].x * F.X[L], padh[sp].y * F.Y[L], padh[sp].z * F.Z[L]});
                dist = hit.dist; pos = hit.pos; sel = hit.code;
            } else {
                dist = sph[sp].z - srad[sp];
                pos = mk(sph[sp].x, sph[sp].y, 0.5f * dist);
                sel = 0;
            }
        } else if (P.arm_collision) {
            // arm-link proxies (D3): both ends of link_3, link_4 motor, link_5 motor body, link_6 jaw root.  One contact: the
            // deepest candidate in the order proxy 0 floor, proxy 1 floor, proxy 2 floor, proxy 3 floor, proxy 3 cubes, proxy 4
            // floor, proxy 4 cubes (ties: first).  Floor candidates need only the height of the proxy centre.
            const int plink[5] = {2, 2, 3, 4, 5};
            const float px[5] = {LPX0x, LPX1x, LPX2x, LPX3x, LPX4x}, py[5] = {LPX0y, LPX1y, LPX2y, LPX3y, LPX4y}, pz[5] = {LPX0z, LPX1z, LPX2z, LPX3z, LPX4z};
            const float pr[5] = {LPX0r, LPX1r, LPX2r, LPX3r, LPX4r};
            // wave-uniform broad phase for the gripper body vs the cubes: both proxies lie within 27 mm of the link_5 / link_6 origins
            bool near_any = false;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const f3 dd = F.p[4] - S.cp[c];
                near_any = near_any || dot(dd, dd) < (0.0350f + 1.7321f * CH) * (0.0350f + 1.7321f * CH);
            }
            const bool wave_near = __any(near_any) != 0;
            float bestd = 1e30f;
            int bi = 0;
#pragma unroll
            for (int i = 0; i < 5; i++) {
                const int L = plink[i];
                const float cz = fmaf(px[i], F.X[L].z, fmaf(py[i], F.Y[L].z, fmaf(pz[i], F.Z[L].z, F.p[L].z)));
                const float df = cz - pr[i];
                if (df < bestd) { bestd = df; bi = i; oncube = false; }
                if (i >= 3 && wave_near) {
                    const f3 ci = local_point(F, L, px[i], py[i], pz[i]);
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const SBHit hit = sphere_box(ci, pr[i], S.cp[c], CR[c]);
                        if (hit.dist < bestd) { bestd = hit.dist; bi = i; pos = hit.pos; n = hit.n; oncube = true; cidx = c; sel = 32 + 8 * c + hit.code; }
                    }
                }
            }
            link_bi = bi;
            if (!oncube) { n = mk(0.f, 0.f, 1.f); sel = 0; }
            sel += 64 * (bi + 1);
            dist = bestd;
            link_on_cube = oncube;
            slot_cube[2] = cidx;
            link_nj = bi < 2 ? 3 : bi + 2;                       // proxies 0, 1 on link_3 (3 joints); 2 on link_4; 3 on link_5; 4 on link_6
            invw_link = bi < 2 ? INVW_TRAN_L3 : (bi == 2 ? INVW_TRAN_L4 : (bi == 3 ? INVW_TRAN_L5 : INVW_TRAN_L6));
        }
        T.act = dist < 0.f;
        if (may_cube) coupled = coupled || (T.act && (s < 2 || oncube) && real_lane);
        if constexpr (CPL == CPL_FAST) {   // more lanes with a finger or a gripper-body proxy on their cube than are solved one by one: this substep belongs to the other copy
            if (may_cube && __popcll(__ballot(coupled)) > P.coop_max) {
#pragma unroll
                for (int c = 0; c < NC; c++)
#pragma unroll
                    for (int k = 0; k < 4; k++) S.cq[c][k] = cq_in[c][k];
                return false;
            }
        }
        if (P.diag) {
            if (may_cube && (s < 2 || oncube)) sel += (n.y < 0.5f && n.y > -0.5f) ? 0 : ((NEWTON && s < 2) ? 2 : 16);   // branch of make_frame (pad boxes: sel = cube + 2 branch + 4 code)
            diag_choice(DG, T.act, 12 + s, sel);
        }
        slot_any[s] = __any(T.act) != 0;
#pragma unroll
        for (int k = 0; k < NRW; k++) { T.f[k] = 0.f; T.aref[k] = 0.f; T.inv[k] = 0.f; }
        T.Rn = 1.f; T.n = n; T.t1 = mk(0.f, 1.f, 0.f); T.t2 = mk(-1.f, 0.f, 0.f); T.rc = mk(0.f, 0.f, 0.f);
        if (slot_any[s]) {  // wave-uniform: skip the whole row set-up when no lane of the wave touches
            if (s == 4) {
                // floor contact of proxy link_bi: its horizontal position is needed only now (masked sum: a select chain over
                // five computed points would be turned into a scratch array by the compiler)
                const float qx[5] = {LPX0x, LPX1x, LPX2x, LPX3x, LPX4x}, qy[5] = {LPX0y, LPX1y, LPX2y, LPX3y, LPX4y}, qz[5] = {LPX0z, LPX1z, LPX2z, LPX3z, LPX4z};
                const int ql[5] = {2, 2, 3, 4, 5};
                f2v cb = {0.f, 0.f};
#pragma unroll
                for (int i = 0; i < 5; i++) {
                    const float m = link_bi == i ? 1.f : 0.f;
                    const int L = ql[i];
                    const f2v ci = F.p[L].xy + f2v{qx[i], qx[i]} * F.X[L].xy + f2v{qy[i], qy[i]} * F.Y[L].xy + f2v{qz[i], qz[i]} * F.Z[L].xy;
                    cb = f2v{m, m} * ci + cb;
                }
                if (!oncube) pos = mk(cb.x, cb.y, 0.5f * dist);
            }
            if (may_cube) make_frame(n, T.t1, T.t2);   // (for n = +z this is the floor frame t1 = +y, t2 = -x)
            // joints that move the contact point: the finger spheres sit on link_5 / link_6, the proxies on link_3..link_6
            auto joint_on = [&](int j) -> bool {
                if (s < 4) return j < (sp == 0 ? 5 : 6);
                return j < link_nj;
            };
            f3 cube_p = mk(0.f, 0.f, 0.f), cube_v = mk(0.f, 0.f, 0.f), cube_w = mk(0.f, 0.f, 0.f);
            if (may_cube) {
                if (NC == 2 && cidx == 1) { cube_p = S.cp[NC - 1]; cube_v = S.cv[NC - 1]; cube_w = cww[NC - 1]; }
                else { cube_p = S.cp[0]; cube_v = S.cv[0]; cube_w = cww[0]; }
                T.rc = pos - cube_p;
            }
            float imp, Kc, Bc;
            if (s < 2) { imp = impedance(dist, D0_FC, DW_FC, 1.0f / W_FC); Kc = K_FC; Bc = B_FC; }
            else if (s < 4) { imp = impedance(dist, D0_FF, DW_FF, 1.0f / W_FF); Kc = K_FF; Bc = B_FF; }
            else { imp = impedance(dist, D0_DEF, DW_DEF, 1.0f / W_DEF); Kc = K_DEF; Bc = B_DEF; }   // link geoms: default solref / solimp
            float Rn = fmaxf((1.f - imp) * rcp(imp) * (invw_link + ((may_cube && oncube) ? minv : 0.f)), 1e-15f);
            float Rf = Rn * P.inv_impratio;
            float Rt = Rf * (s < 2 ? P.rt_fc : (s < 4 ? RT_FF : P.rt_cube));
            T.Rn = Rn;
            // squared friction coefficients of this slot's rows (finger<->cube pair: max rule; finger geoms mu 1.5 / torsional 0.005; a link proxy on the
            // floor mu 1, on a cube the cube's) and the metric of the block step (soc_step)
            const float m2_tan = s < 2 ? P.mu_fc2 : (s < 4 ? MU_FINGER * MU_FINGER : (oncube ? P.mu_c2 : 1.f));
            const float m2_tors = s < 2 ? P.mu_fct2 : (s < 4 ? MU_TORS * MU_TORS : P.mu_ct2);
            float Ln = 1.f, Lt = 0.f;
            // point Jacobian columns of the link at the contact point
            f3 jc[6];
#pragma unroll
            for (int j = 0; j < 6; j++) {
                const bool lit = s < 4 ? j < (sp == 0 ? 5 : 6) : true;   // columns that can be non-zero at all
                jc[j] = lit ? cross(z[j], pos - F.p[j]) : mk(0.f, 0.f, 0.f);
                if (s == 4 && j >= 3 && !joint_on(j)) jc[j] = mk(0.f, 0.f, 0.f);
            }
            // NEWTON: the three angular rows of a finger contact (torsion, two rolling rows) are combinations of the finger BODY's angular rows
            // B_a = L^-1 (z_j . e_a)_j, a = x, y, z -- kept once per finger (LDS rows 12 + 3 sp + a) instead of three rows per slot: 22 g rows, 33 KiB per wave,
            // four waves per CU (28 rows were 42 KiB: three).  d . B is also what the warm start and the relative angular velocity need.
            float Bb[3][6];
            f3 wbody = mk(0.f, 0.f, 0.f), tau = mk(0.f, 0.f, 0.f);
            if (NEWTON && s < 4) {
                if (!body_done[sp]) {
#pragma unroll
                    for (int a = 0; a < 3; a++) {
#pragma unroll
                        for (int j = 0; j < 6; j++) Bb[a][j] = joint_on(j) ? (a == 0 ? z[j].x : (a == 1 ? z[j].y : z[j].z)) : 0.f;
                        fsub(CL, Bb[a]);
#pragma unroll
                        for (int k = 0; k < 3; k++)
                            *reinterpret_cast<float2v *>(&lds[(12 + 3 * sp + a) * LDS_ROW + k * 128 + lane * 2]) = float2v{Bb[a][2 * k], Bb[a][2 * k + 1]};
                    }
                    body_done[sp] = true;
                } else {
#pragma unroll
                    for (int a = 0; a < 3; a++)
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            const float2v gp = *reinterpret_cast<const float2v *>(&lds[(12 + 3 * sp + a) * LDS_ROW + k * 128 + lane * 2]);
                            Bb[a][2 * k] = gp.x; Bb[a][2 * k + 1] = gp.y;
                        }
                }
#pragma unroll
                for (int j = 0; j < 6; j++) { if (joint_on(j)) wbody = axpy(S.qd[j], z[j], wbody); }
            }
#pragma unroll
            for (int r = 0; r < arm_rows_of<ROLL, NEWTON>(s); r++) {
                const bool body_row = NEWTON && s < 4 && r >= 3;   // (literal after unrolling)
                f3 d = r == 0 ? T.n : (r == 1 ? T.t1 : (r == 2 ? T.t2 : T.n));   // row 3: rotation about n (torsion)
                if constexpr (ROLL) { if (r >= 4) d = r == 4 ? T.t1 : T.t2; }     // rows 4, 5: rotation about t1, t2 (rolling)
                float g[6];
                float vel = 0.f;
                if (body_row) {
#pragma unroll
                    for (int j = 0; j < 6; j++) g[j] = 0.f;
                    vel = dot(d, wbody);
                } else {
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        g[j] = r < 3 ? dot(jc[j], d) : (joint_on(j) ? dot(z[j], d) : 0.f);
                        vel = fmaf(g[j], S.qd[j], vel);
                    }
                }
                float diagc = 0.f;
                if (may_cube) {
                    float velc;
                    if (r < 3) {
                        f3 rxd = cross(T.rc, d);
                        velc = dot(d, cube_v) + dot(rxd, cube_w);
                        diagc = minv + iinv * dot(rxd, rxd);
                    } else {
                        velc = dot(d, cube_w);
                        diagc = iinv;
                    }
                    if (s == 4) { velc = oncube ? velc : 0.f; diagc = oncube ? diagc : 0.f; }
                    vel -= velc;
                }
                if (!body_row) fsub(CL, g);
                float gg = 0.f;
#pragma unroll
                for (int j = 0; j < 6; j++) gg = fmaf(g[j], g[j], gg);
                // (stored as the float2 pairs the sweeps load: a scalar store read back through a float2 lvalue would be a strict-aliasing
                //  violation, and the compiler did move such loads above the stores in one kernel variant)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    if (body_row) continue;
                    const float2v gp = {g[2 * k], g[2 * k + 1]};
                    if (NC == 2 && !BIG && s == 4) *reinterpret_cast<float2v *>(&P.scratch[((size_t)(r * 3 + k) * P.n + env) * 2]) = gp;
                    else if (NC == 2 && !BIG && r >= 4) *reinterpret_cast<float2v *>(&P.scratch[((size_t)((4 + 2 * s + (r - 4)) * 3 + k) * P.n + env) * 2]) = gp;
                    else *reinterpret_cast<float2v *>(&lds[(arm_row0_of<ROLL, NC, BIG, NEWTON>(s) + r) * LDS_ROW + k * 128 + lane * 2]) = gp;
                }
                float Rr = r == 0 ? Rn : (r == 3 ? Rt : Rf);
                if (ROLL && r > 3) Rr = Rf * (s < 2 ? P.rr_fc : RR_FF);   // (rolling rows of a finger on the floor: NEWTON only)
                T.aref[r] = -Bc * vel - (r == 0 ? Kc * imp * dist : 0.f);
                // warm start: previous substep's force of this slot (zero if it was inactive), applied to the accelerations
                const bool row_on = T.act && (s != 4 || r < 3 || oncube);   // a link proxy on the floor has no torsion row (condim 3)
                {
                    const float arr = gg + diagc + Rr;
                    const float m2r = r == 0 ? 1.f : (r < 3 ? m2_tan : (r == 3 ? m2_tors : P.mu_fcr2));
                    const float KF = 2.f;
                    if (r == 0) Ln = KF * arr;
                    else Lt = fmaf(row_on ? KF * m2r : 0.f, arr, Lt);
                }
                const float fw = row_on ? W.arm[s][r] : 0.f;
                T.f[r] = fw;
                if (body_row) tau = axpy(fw, d, tau);
                else {
#pragma unroll
                    for (int j = 0; j < 6; j++) y[j] = fmaf(g[j], fw, y[j]);
                }
                if (may_cube) {
                    const float fc = (s == 4 && !oncube) ? 0.f : fw;
                    f3 dl = r < 3 ? (-minv * fc) * d : mk(0.f, 0.f, 0.f);
                    f3 da = r < 3 ? (-iinv * fc) * cross(T.rc, d) : (-iinv * fc) * d;
                    if (NC == 2 && cidx == 1) { ca[NC - 1] = ca[NC - 1] + dl; cal[NC - 1] = cal[NC - 1] + da; }
                    else { ca[0] = ca[0] + dl; cal[0] = cal[0] + da; }
                }
            }
            if (NEWTON && s < 4) {   // the warm-start torque of the three angular rows, through the body rows
#pragma unroll
                for (int j = 0; j < 6; j++) y[j] = fmaf(Bb[0][j], tau.x, fmaf(Bb[1][j], tau.y, fmaf(Bb[2][j], tau.z, y[j])));
            }
            {   // k[] of soc_step: iLn, mu_tan^2 iLt, w, mu_tors^2 iLt (, mu_roll^2 iLt)
                const float iLn = T.act ? rcp(Ln) : 0.f, iLt = T.act ? rcp(Lt) : 0.f, iLs = iLt;
                T.inv[0] = iLn; T.inv[1] = m2_tan * iLt; T.inv[2] = Ln * rcp(Ln + Lt); T.inv[3] = (s != 4 || oncube) ? m2_tors * iLs : 0.f;   // (a link proxy on the floor has no torsion row: condim 3)
                if constexpr (ROLL) { T.inv[4] = P.mu_fcr2 * iLs; T.inv[5] = 0.f; }
            }
        }
    }
    }

    // ---- joint limits (rare): rows +-e_j ---------------------------------------------------------
    float flim[6];
    bool lim_act[6];
    unsigned lim_wave = 0u;   // bit j: joint j is beyond a limit in SOME lane (wave-uniform).  Under random actions that is joint 0 in ~40 % of the waves
                              // and the others almost never; a joint no lane needs contributes exact zeros, so its rows are skipped.
#pragma unroll
    for (int j = 0; j < 6; j++) {
        lim_act[j] = (S.q[j] < JLO[j]) || (S.q[j] > JHI[j]);
        if (P.diag) diag_choice(DG, lim_act[j], 18 + j, S.q[j] < JLO[j] ? 0 : 1);
        flim[j] = lim_act[j] ? W.lim[j] : 0.f;
        lim_wave |= __any(lim_act[j]) ? (1u << j) : 0u;
    }
    const bool wave_lim = lim_wave != 0u;
    if (wave_lim) {  // apply the warm-start limit forces
#pragma unroll
        for (int j = 0; j < 6; j++) {
            if (!((lim_wave >> j) & 1u)) continue;
            float g[6];
#pragma unroll
            for (int k = 0; k < 6; k++) g[k] = k == j ? (S.q[j] < JLO[j] ? 1.f : -1.f) : 0.f;
            fsub(CL, g);
#pragma unroll
            for (int k = 0; k < 6; k++) y[k] = fmaf(g[k], flim[j], y[k]);
        }
    }
    const bool wave_arm = slot_any[0] || slot_any[1] || slot_any[2] || slot_any[3] || slot_any[4];

    // ---- projected Gauss-Seidel on the dual, matrix-free, warm-started.  Fixed sweep count (pgs_iters > 0), or ADAPT
    //      (pgs_iters < 0): sweep until the largest force change of a sweep is <= pgs_tol (1 + largest |force|) in EVERY lane
    //      of the wave, at most 50 sweeps ----
    const int max_it = NEWTON ? 0 : (ADAPT ? 50 : P.pgs_iters);
    int sweeps_done = 0;
    bool coupled_any = false;   // (NEWTON, one cube) some lane of the wave had the arm on its cube in this substep
    if constexpr (NEWTON) {
        // ---- Newton on the primal (lcr_newton.h; oracle: newton_product).  The set-up above has already put the carried forces' accelerations into y / ca / cal:
        //      x0 = a0 + M^-1 J'f, MuJoCo's qacc_warmstart.  The bodies are solved per connected component of the wave's coupling graph: while no lane has a finger
        //      or a gripper-body proxy on the cube, arm and cube are independent 6-dimensional problems ----
        const int row0[NAS] = {arm_row0_of<ROLL, NC, BIG, NEWTON>(0), arm_row0_of<ROLL, NC, BIG, NEWTON>(1), arm_row0_of<ROLL, NC, BIG, NEWTON>(2),
                               arm_row0_of<ROLL, NC, BIG, NEWTON>(3), arm_row0_of<ROLL, NC, BIG, NEWTON>(4)};
        const NewtonParams NP = newton_params(P);
        FloorSlot no_walls[4];
        const float no_wsg[2] = {1.f, 1.f};
        NewtonCtx<NC, NRW, false, NCC> C{NP, lds, lane, row0, AS, slot_any, link_on_cube, slot_cube, FS, no_walls, no_wsg, false,
                                         ccl, cc_act, cc_any, ccn, cct1, cct2, S.cp, lim_act, lim_wave, S.q, S.qd, CL, flim, y0s};
        const long long tn0 = P.diag == 2 ? clock64() : 0;   // (profiling aid: cycles of the solves, of the coupled ones, iterations -- tools/newton_phases.py)
        bool prof_coupled = false;
        unsigned prof_coop = 0u;   // (one cube) cycles of the cooperative solves of this substep, number of envs solved that way
        int prof_patients = 0;
        if constexpr (NC == 1) {
            const bool arm_on_cube = slot_any[0] || slot_any[1] || (__any(AS[4].act && link_on_cube) != 0);
            prof_coupled = arm_on_cube;
            const unsigned long long cmask = __ballot(coupled);
            coupled_any = __popcll(cmask) > P.coop_max;   // (CPL_SLOW: stay in this copy while more lanes are coupled than the cooperative solve takes)
            if constexpr (CPL == CPL_SLOW) sweeps_done = newton_solve<NC, NRW, false, NCC, 3>(C, y, ca, cal);   // (uncoupled lanes: the same optimum, block-diagonal Hessian)
            else if (CPL == CPL_BOTH && arm_on_cube) sweeps_done = newton_solve<NC, NRW, false, NCC, 3>(C, y, ca, cal);
            else {
                // the lanes whose arm touches their cube sit out the two small solves and are then solved one by one by the whole wave (lcr_newton_coop.h)
                C.enable = (CPL == CPL_BOTH || !coupled) ? 7 : 0;
                const int ia = newton_solve<NC, NRW, false, NCC, 1>(C, y, ca, cal);
                const int ic = newton_solve<NC, NRW, false, NCC, 2>(C, y, ca, cal);
                sweeps_done = max(ia, ic);
                if constexpr (CPL == CPL_FAST) {
                    float *stage = lds + NEWTON_G_ROWS * LDS_ROW;
                    const long long tc0 = P.diag == 2 ? clock64() : 0;
                    prof_patients = __popcll(cmask);
#if LCR_COOP_ROWS > 0
                    for (unsigned long long m = cmask; m != 0ull;) {   // LCR_COOP_ROWS patients per pass, one per 16-lane row (lcr_newton_coop.h: coop_solve_rows)
                        unsigned long long pm = 0ull;
                        for (int k = 0; k < LCR_COOP_ROWS && m != 0ull; k++) { pm |= m & (0ull - m); m &= m - 1ull; }
                        coop_solve_rows<NC, NRW, NCC>(C, stage, lane, pm, 0ull, y, ca, cal, sweeps_done);
                    }
#else
                    for (unsigned long long m = cmask; m != 0ull; m &= m - 1ull) {
                        const int L = __builtin_ctzll(m);
                        const int ip = coop_solve<NC, NRW, NCC>(C, stage, lane, L, y, ca, cal);
                        sweeps_done = lane == L ? ip : sweeps_done;
                    }
#endif
                    if (P.diag == 2) prof_coop = (unsigned)(clock64() - tc0);
                }
            }
        } else {
            // edges of the wave's coupling graph: arm <-> cube c where some lane has a finger sphere or a gripper-body proxy on that cube, cube 0 <-> cube 1 where some
            // lane's cubes touch
            bool on0 = false, on1 = false;
#pragma unroll
            for (int s = 0; s < 2; s++) { on0 = on0 || (AS[s].act && slot_cube[s] == 0); on1 = on1 || (AS[s].act && slot_cube[s] == 1); }
            on0 = on0 || (AS[4].act && link_on_cube && slot_cube[2] == 0); on1 = on1 || (AS[4].act && link_on_cube && slot_cube[2] == 1);
            const bool eA0 = __any(on0) != 0, eA1 = __any(on1) != 0, e01 = cc_any;
            int i1 = 0, i2 = 0, i3 = 0;
            const unsigned long long cmask = __ballot(coupled);
            coupled_any = __popcll(cmask) > P.coop_max;
            if constexpr (CPL == CPL_FAST) {
                // the bodies of an env that touch sit out the three small solves and are then solved by the whole wave (lcr_newton_coop.h), one env after the other: the
                // arm and ONE cube (12 unknowns; the other cube keeps its small solve) where the arm on that cube is the env's only coupling -- nine patients of ten under
                // random actions --, all three bodies (18 unknowns) otherwise
                const bool pair0 = on0 && !on1 && !(cc_any && cc_count > 0), pair1 = on1 && !on0 && !(cc_any && cc_count > 0);
                C.enable = !coupled ? 7 : (pair0 ? 4 : (pair1 ? 2 : 0));
                i1 = newton_solve<NC, NRW, false, NCC, 1>(C, y, ca, cal); i2 = newton_solve<NC, NRW, false, NCC, 2>(C, y, ca, cal);
                i3 = newton_solve<NC, NRW, false, NCC, 4>(C, y, ca, cal);
                float *stage = lds + NEWTON_G_ROWS * LDS_ROW + 8 * CCR * 64;
                const long long tc0 = P.diag == 2 ? clock64() : 0;
                prof_patients = __popcll(cmask);
                const unsigned long long m0 = __ballot(pair0), m1 = __ballot(pair1);
#ifdef LCR_STACK_ONE_PER_PASS
                // (A/B: one patient at a time with all 64 lanes -- Stack, 32 768 envs: 6.81 ms against 6.03 with four per pass)
                for (unsigned long long m = m0 | m1; m != 0ull; m &= m - 1ull) {
                    const int L = __builtin_ctzll(m);
                    const int ip = coop_solve<NC, NRW, NCC, 1>(C, stage, lane, L, y, ca, cal, (int)(m1 >> L & 1ull));
                    i1 = lane == L ? max(ip, i1) : i1;
                }
#else
                for (unsigned long long m = m0 | m1; m != 0ull;) {   // arm + one cube: four patients per pass, one per 16-lane row (coop_solve_rows)
                    unsigned long long pm = 0ull;
                    for (int k = 0; k < 4 && m != 0ull; k++) { pm |= m & (0ull - m); m &= m - 1ull; }
                    int ip = 0;
                    coop_solve_rows<NC, NRW, NCC>(C, stage, lane, pm, m1, y, ca, cal, ip);
                    i1 = max(ip, i1);
                }
#endif
                for (unsigned long long m = cmask & ~(m0 | m1); m != 0ull; m &= m - 1ull) {   // all three bodies: one patient at a time
                    const int L = __builtin_ctzll(m);
                    const int ip = coop_solve<NC, NRW, NCC, 2>(C, stage, lane, L, y, ca, cal);
                    i1 = lane == L ? max(ip, i1) : i1;
                }
                if (P.diag == 2) prof_coop = (unsigned)(clock64() - tc0);
            } else
            if ((eA0 && eA1) || (e01 && (eA0 || eA1))) i1 = newton_solve<NC, NRW, false, NCC, 7>(C, y, ca, cal);
            else if (eA0) { i1 = newton_solve<NC, NRW, false, NCC, 3>(C, y, ca, cal); i2 = newton_solve<NC, NRW, false, NCC, 4>(C, y, ca, cal); }
            else if (eA1) { i1 = newton_solve<NC, NRW, false, NCC, 5>(C, y, ca, cal); i2 = newton_solve<NC, NRW, false, NCC, 2>(C, y, ca, cal); }
            else if (e01) { i1 = newton_solve<NC, NRW, false, NCC, 1>(C, y, ca, cal); i2 = newton_solve<NC, NRW, false, NCC, 6>(C, y, ca, cal); }
            else {
                i1 = newton_solve<NC, NRW, false, NCC, 1>(C, y, ca, cal); i2 = newton_solve<NC, NRW, false, NCC, 2>(C, y, ca, cal);
                i3 = newton_solve<NC, NRW, false, NCC, 4>(C, y, ca, cal);
            }
            sweeps_done = max(i1, max(i2, i3));
            prof_coupled = eA0 || eA1 || e01;
        }
        if (P.diag == 2) {
            const unsigned dt = (unsigned)(clock64() - tn0);
            // mask: cycles of the solves; count: of the coupled ones (the cooperative solves, or the whole solve of a substep in the coupled SIMT copy);
            // choice: iterations + 65536 envs solved cooperatively + 2^24 substeps in the coupled SIMT copy
            DGtot.mask += dt; DGtot.count += CPL == CPL_SLOW ? dt : prof_coop;
            DGtot.choice += (CPL == CPL_SLOW ? (1u << 24) : 0u) + 65536u * (unsigned)prof_patients + (unsigned)C.wave_its;
            (void)prof_coupled;
        }
    }
    for (int it = 0; it < max_it; it++) {
        float chg = 0.f, fmx = 0.f;   // ADAPT: largest |force change| and |force| of this sweep
        auto track = [&](float d0, float d1, float d2, float d3, float f0, float f1, float f2, float f3_) {
            if (ADAPT) {
                chg = fmaxf(fmaxf(chg, fmaxf(fabsf(d0), fabsf(d1))), fmaxf(fabsf(d2), fabsf(d3)));
                fmx = fmaxf(fmaxf(fmx, fmaxf(fabsf(f0), fabsf(f1))), fmaxf(fabsf(f2), fabsf(f3_)));
            }
        };
        // ---- one arm-coupled slot (finger spheres, arm-link proxies): a block step on its rows ----
        // The rows of a sweep form two groups that are swept as if concurrently (what the two-wave kernels do with two waves; oracle: orc_params.jacobi) --
        //   A: joint limits, finger<->floor (slots 2, 3), arm-link proxies (slot 4)          B: floor<->cube, cube<->cube, finger<->cube (slots 0, 1; two cubes: group A)
        // Gauss-Seidel inside a group; a group sees the other group's effect on the shared unknowns (the arm acceleration y through slots 0, 1; the cube
        // accelerations through slot 4) as of the START of the sweep.  In program order group A runs first: y then carries A's changes, yB keeps the sweep-start
        // value for slots 0, 1 (whose changes go to both copies), and slot 4's change of the cube accelerations is held back in dcaA until the end of the sweep.
        float yB[6];
        f3 dcaA[NC], dcalA[NC];
#pragma unroll
        for (int k = 0; k < 6; k++) yB[k] = y[k];
#pragma unroll
        for (int c = 0; c < NC; c++) { dcaA[c] = mk(0.f, 0.f, 0.f); dcalA[c] = mk(0.f, 0.f, 0.f); }
        auto arm_slot = [&](auto s_tag) {
            constexpr int s = decltype(s_tag)::value;
            if (!wave_arm || !slot_any[s]) return;
            ArmSlot<NRW> &T = AS[s];
            const bool may_cube = s < 2 || s == 4;
            const bool oncube = s < 2 || (s == 4 && link_on_cube);
            constexpr bool roll = ROLL;
            const int nrow = (roll && s < 2) ? 6 : 4;
            const float Rf = T.Rn * P.inv_impratio;
            const float Rt = Rf * (s < 2 ? P.rt_fc : (s < 4 ? RT_FF : P.rt_cube));
            // the six arm components travel as three float2 pairs: dot products and updates become v_pk_mul/v_pk_fma
            float2v g[NRW][3];
#pragma unroll
            for (int r = 0; r < nrow; r++)
#pragma unroll
                for (int k = 0; k < 3; k++)
                    g[r][k] = (NC == 2 && !BIG && s == 4) ? *reinterpret_cast<const float2v *>(&P.scratch[((size_t)(r * 3 + k) * P.n + env) * 2])
                              : (NC == 2 && !BIG && r >= 4) ? *reinterpret_cast<const float2v *>(&P.scratch[((size_t)((4 + 2 * s + (r - 4)) * 3 + k) * P.n + env) * 2])
                                                  : *reinterpret_cast<const float2v *>(&lds[(as_row0<ROLL, NC, BIG>(s) + r) * LDS_ROW + k * 128 + lane * 2]);
            // one cube: the finger<->cube slots (s < 2) belong to group B -- they work from y as of the start of the sweep (yB) and their changes go to BOTH copies;
            // group A (s >= 2; with two cubes every arm slot: "all rows that touch the arm" -- Stack's group B has two cubes' floor rows and the cube<->cube rows already) works on y
            constexpr bool inB = s < 2 && NC == 1;
            float2v yp[3] = {{inB ? yB[0] : y[0], inB ? yB[1] : y[1]}, {inB ? yB[2] : y[2], inB ? yB[3] : y[3]}, {inB ? yB[4] : y[4], inB ? yB[5] : y[5]}};
            float2v yq[3] = {{y[0], y[1]}, {y[2], y[3]}, {y[4], y[5]}};
            float arefv[NRW], invv[NRW], f_in[NRW];
#pragma unroll
            for (int r = 0; r < NRW; r++) { arefv[r] = T.aref[r]; invv[r] = T.inv[r]; f_in[r] = T.f[r]; }
            // pick the cube this slot talks to (wave-divergent only for Stack)
            f3 a_lin = mk(0.f, 0.f, 0.f), a_ang = mk(0.f, 0.f, 0.f);
            const bool second = may_cube && NC == 2 && slot_cube[s == 4 ? 2 : (s & 1)] == 1;
            if (may_cube) { a_lin = second ? ca[NC - 1] : ca[0]; a_ang = second ? cal[NC - 1] : cal[0]; }
            if (may_cube && !inB) {   // group A sees the cube accelerations as of the start of the sweep plus its OWN changes so far (Gauss-Seidel inside the group)
                a_lin = a_lin + (second ? dcaA[NC - 1] : dcaA[0]); a_ang = a_ang + (second ? dcalA[NC - 1] : dcalA[0]);
            }
            // cube-side inverse inertia of this lane's contact (zero when the proxy slot touches the floor: the cube terms vanish)
            const float minv_e = (s == 4 && !oncube) ? 0.f : minv, iinv_e = (s == 4 && !oncube) ? 0.f : iinv;
            // the cube's share of the gradient rows: v_r = d_r . (acceleration of the cube's contact point), wn / w1 / w2 = (n, t1, t2) . (angular acceleration)
            float vq[3] = {0.f, 0.f, 0.f}, wn = 0.f, w1 = 0.f, w2 = 0.f;
            if (may_cube) {
                const f3 Ac = a_lin + cross(a_ang, T.rc);
                vq[0] = dot(T.n, Ac); vq[1] = dot(T.t1, Ac); vq[2] = dot(T.t2, Ac);
                wn = dot(T.n, a_ang);
                if (nrow == 6) { w1 = dot(T.t1, a_ang); w2 = dot(T.t2, a_ang); }
                if (s == 4) {   // floor lanes: no cube share in the rows
#pragma unroll
                    for (int i = 0; i < 3; i++) vq[i] = oncube ? vq[i] : 0.f;
                    wn = oncube ? wn : 0.f;
                }
            }
            // gradient rows of the block from the SAME forces (no serial dependence inside the block), one projected-gradient step (soc_step), then y follows
            float u[NRW], fcur[NRW], nf[NRW];
#pragma unroll
            for (int r = 0; r < NRW; r++) {
                fcur[r] = T.f[r];
                u[r] = 0.f;
                if (r < nrow) {
                    const float2v acc = g[r][0] * yp[0] + g[r][1] * yp[1] + g[r][2] * yp[2];
                    const float gy = acc.x + acc.y;
                    float jc_a = may_cube ? (r < 3 ? -vq[r] : -wn) : 0.f;
                    float Rr = r == 0 ? T.Rn : (r == 3 ? Rt : Rf);
                    if (ROLL && r > 3) { jc_a = r == 4 ? -w1 : -w2; Rr = Rf * P.rr_fc; }
                    u[r] = gy + jc_a - arefv[r] + Rr * fcur[r];
                }
            }
            {   // (finger geoms: mu 1.5 / torsional 0.005; finger<->cube pair: max rule; a link proxy on the floor: mu 1, on a cube: the cube's friction)
                const float imu2 = s < 4 ? 1.f / (MU_FINGER * MU_FINGER) : (oncube ? P.inv_mu_c2 : 1.f);
                const float imt2 = s < 2 ? P.inv_mu_fct2 : (s < 4 ? 1.f / (MU_TORS * MU_TORS) : P.inv_mu_ct2);
                soc_step<NRW>(fcur, u, invv, imu2, imt2, P.inv_mu_fcr2, nrow, nf);
            }
#pragma unroll
            for (int r = 0; r < NRW; r++) {
                if (r < nrow) {
                    const float dlt = nf[r] - fcur[r];
                    T.f[r] = nf[r];
                    const float2v d2 = {dlt, dlt};
#pragma unroll
                    for (int k = 0; k < 3; k++) { yp[k] = g[r][k] * d2 + yp[k]; if (inB) yq[k] = g[r][k] * d2 + yq[k]; }
                }
            }
            // (converged mode: the net force change of this sweep)
            track(T.f[0] - f_in[0], T.f[1] - f_in[1], T.f[2] - f_in[2], T.f[3] - f_in[3], T.f[0], T.f[1], T.f[2], T.f[3]);
            if constexpr (ROLL) { if (nrow == 6) track(T.f[4] - f_in[4], T.f[5] - f_in[5], 0.f, 0.f, T.f[4], T.f[5], 0.f, 0.f); }
            f3 dl_lin = mk(0.f, 0.f, 0.f), dl_ang = mk(0.f, 0.f, 0.f);  // change of the cube acceleration by this slot
            if (may_cube) {
                const float e0 = T.f[0] - f_in[0], e1 = T.f[1] - f_in[1], e2 = T.f[2] - f_in[2], e3 = T.f[3] - f_in[3];
                const f3 Fd = axpy(e0, T.n, axpy(e1, T.t1, e2 * T.t2));   // force change on the arm; the cube gets -Fd at rc
                dl_lin = (-minv_e) * Fd;
                f3 Td = axpy(e3, T.n, cross(T.rc, Fd));
                if constexpr (ROLL) { if (nrow == 6) Td = axpy(T.f[4] - f_in[4], T.t1, axpy(T.f[5] - f_in[5], T.t2, Td)); }
                dl_ang = (-iinv_e) * Td;
            }
            if (inB) {
                yB[0] = yp[0].x; yB[1] = yp[0].y; yB[2] = yp[1].x; yB[3] = yp[1].y; yB[4] = yp[2].x; yB[5] = yp[2].y;
                y[0] = yq[0].x; y[1] = yq[0].y; y[2] = yq[1].x; y[3] = yq[1].y; y[4] = yq[2].x; y[5] = yq[2].y;
            } else {
                y[0] = yp[0].x; y[1] = yp[0].y; y[2] = yp[1].x; y[3] = yp[1].y; y[4] = yp[2].x; y[5] = yp[2].y;
            }
            if (may_cube) {
                // group B's slots change the cube accelerations the group is sweeping; group A's read them as of the start of the sweep (they run
                // before group B's rows) and their change is held back until the end of the sweep
                f3 (&tca)[NC] = inB ? ca : dcaA;
                f3 (&tcal)[NC] = inB ? cal : dcalA;
                if (NC == 2) {
                    if (second) { tca[NC - 1] = tca[NC - 1] + dl_lin; tcal[NC - 1] = tcal[NC - 1] + dl_ang; }
                    else { tca[0] = tca[0] + dl_lin; tcal[0] = tcal[0] + dl_ang; }
                } else { tca[0] = tca[0] + dl_lin; tcal[0] = tcal[0] + dl_ang; }
            }
        };
        if (wave_lim) {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                if (!((lim_wave >> j) & 1u)) continue;
                const bool lower = S.q[j] < JLO[j];
                const float sg = lower ? 1.f : -1.f;
                const float pos = lower ? S.q[j] - JLO[j] : JHI[j] - S.q[j];
                float imp = impedance(pos, D0_DEF, DW_DEF, 1.0f / W_DEF);
                float Rl = fmaxf((1.f - imp) * rcp(imp) * INVW_DOF[j], 1e-15f);
                float aref = -B_DEF * sg * S.qd[j] - K_DEF * imp * pos;
                float g[6];
#pragma unroll
                for (int k = 0; k < 6; k++) g[k] = k == j ? sg : 0.f;
                fsub(CL, g);
                float gg = 0.f, gy = 0.f;
#pragma unroll
                for (int k = 0; k < 6; k++) { gg = fmaf(g[k], g[k], gg); gy = fmaf(g[k], y[k], gy); }
                float res = gy - aref + Rl * flim[j];
                float nf = fmaxf(flim[j] - res * rcp(gg + Rl), 0.f);
                float dl = lim_act[j] ? nf - flim[j] : 0.f;
                flim[j] += dl;
                track(dl, 0.f, 0.f, 0.f, flim[j], 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < 6; k++) y[k] = fmaf(g[k], dl, y[k]);
            }
        }
        // group A: (two cubes: finger<->cube,) finger<->floor, arm-link proxies (after the joint limits above)
        if constexpr (NC == 2) { arm_slot(std::integral_constant<int, 0>{}); arm_slot(std::integral_constant<int, 1>{}); }
        arm_slot(std::integral_constant<int, 2>{});
        arm_slot(std::integral_constant<int, 3>{});
        arm_slot(std::integral_constant<int, 4>{});
        // floor <-> cube
#pragma unroll
        for (int c = 0; c < NC; c++) {
#pragma unroll
            for (int s = 0; s < 4; s++) {
                FloorSlot &T = FS[c][s];
                const f3 r = T.r;
                const float Rf = T.Rn * P.inv_impratio;
                const float Rt = Rf * P.rt_cube;
                // rows J_r = [d_r ; c_r], d = (z, y, -x, 0), c = (r x d) resp. z for the torsion row: the gradient rows u_r against the CURRENT accelerations
                float aref0 = T.aref[0], aref1 = T.aref[1], aref2 = T.aref[2], aref3 = T.aref[3];
                float inv0 = T.inv[0], inv1 = T.inv[1], inv2 = T.inv[2], inv3 = T.inv[3];
                if (c == 0 && NC == 1 && !ROLL) {
                    const float4v *pk = reinterpret_cast<const float4v *>(lds + LDS_G_FLOATS) + (size_t)(s * 2) * 64 + lane;
                    const float4v a4 = pk[0], i4 = pk[64];
                    aref0 = a4.x; aref1 = a4.y; aref2 = a4.z; aref3 = a4.w;
                    inv0 = i4.x; inv1 = i4.y; inv2 = i4.z; inv3 = i4.w;
                }
                const float u0 = ca[c].z + r.y * cal[c].x - r.x * cal[c].y - aref0 + T.Rn * T.f[0];
                const float u1 = ca[c].y - r.z * cal[c].x + r.x * cal[c].z - aref1 + Rf * T.f[1];
                const float u2 = -ca[c].x - r.z * cal[c].y + r.y * cal[c].z - aref2 + Rf * T.f[2];
                const float u3 = cal[c].z - aref3 + Rt * T.f[3];
                // one projected-gradient step of the whole block (soc_step): the four gradient rows above are taken from the same accelerations
                const float uu[4] = {u0, u1, u2, u3}, kk[4] = {inv0, inv1, inv2, inv3};
                float nf[4];
                soc_step<4>(T.f, uu, kk, P.inv_mu_c2, P.inv_mu_ct2, 0.f, 4, nf);
                const float d0 = nf[0] - T.f[0], d1 = nf[1] - T.f[1], d2 = nf[2] - T.f[2], d3 = nf[3] - T.f[3];   // (inactive slot: f = 0, k = 0 -> every delta is 0)
                T.f[0] = nf[0]; T.f[1] = nf[1]; T.f[2] = nf[2]; T.f[3] = nf[3];
                track(d0, d1, d2, d3, T.f[0], T.f[1], T.f[2], T.f[3]);   // (converged mode: net change of the sweep, after the cone projection)
                // a += M^-1 J^T delta
                ca[c].z = fmaf(minv, d0, ca[c].z);
                ca[c].y = fmaf(minv, d1, ca[c].y);
                ca[c].x = fmaf(-minv, d2, ca[c].x);
                cal[c].x = fmaf(iinv, r.y * d0 - r.z * d1, cal[c].x);
                cal[c].y = fmaf(iinv, -r.x * d0 - r.z * d2, cal[c].y);
                cal[c].z = fmaf(iinv, fmaf(r.x, d1, fmaf(r.y, d2, d3)), cal[c].z);
            }
        }
        // cube <-> cube (Stack): block form of the four rows of each contact.  With an orthonormal frame the couplings between
        // the rows need only the projections of the two lever arms on the frame: (r x d_i).(r x d_j) = -(r.d_i)(r.d_j), i != j.
        if constexpr (NC == 2) {
            if (cc_any) {
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    const f3 pos = mk(ccl[(size_t)(s * CCR + 0) * CS], ccl[(size_t)(s * CCR + 1) * CS], ccl[(size_t)(s * CCR + 2) * CS]);
                    const f3 r0 = pos - S.cp[0], r1 = pos - S.cp[1];
                    const float Rn = ccl[(size_t)(s * CCR + CCRN) * CS];
                    const float Rf = Rn * P.inv_impratio;
                    const float Rt = Rf * P.rt_cube;
                    float f[4], aref[4], inv[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        f[r] = ccl[(size_t)(s * CCR + 3 + r) * CS];
                        aref[r] = ccl[(size_t)(s * CCR + 7 + r) * CS];
                        inv[r] = ccl[(size_t)(s * CCR + 11 + r) * CS];
                    }
                    // relative acceleration of the contact point (cube 1 minus cube 0) and relative angular acceleration
                    const f3 A = (ca[1] + cross(cal[1], r1)) - (ca[0] + cross(cal[0], r0));
                    const f3 Wr = cal[1] - cal[0];
                    const float u0 = dot(ccn, A) - aref[0] + Rn * f[0];
                    const float u1 = dot(cct1, A) - aref[1] + Rf * f[1];
                    const float u2 = dot(cct2, A) - aref[2] + Rf * f[2];
                    const float u3 = dot(ccn, Wr) - aref[3] + Rt * f[3];
                    const float uu[4] = {u0, u1, u2, u3};
                    float nf[4];
                    soc_step<4>(f, uu, inv, P.inv_mu_c2, P.inv_mu_ct2, 0.f, 4, nf);   // one projected-gradient step of the block (lcr_step_common.h)
                    const float d0 = nf[0] - f[0], e1 = nf[1] - f[1], e2 = nf[2] - f[2], e3 = nf[3] - f[3];
                    track(d0, e1, e2, e3, nf[0], nf[1], nf[2], nf[3]);
                    ccl[(size_t)(s * CCR + 3) * CS] = nf[0]; ccl[(size_t)(s * CCR + 4) * CS] = nf[1];
                    ccl[(size_t)(s * CCR + 5) * CS] = nf[2]; ccl[(size_t)(s * CCR + 6) * CS] = nf[3];
                    // a += M^-1 J^T delta: the force change F acts at the contact point on cube 1 and, negated, on cube 0
                    const f3 Fd = axpy(d0, ccn, axpy(e1, cct1, e2 * cct2));
                    const f3 T1 = axpy(e3, ccn, cross(r1, Fd)), T0 = axpy(e3, ccn, cross(r0, Fd));
                    ca[1] = axpy(minv, Fd, ca[1]); ca[0] = axpy(-minv, Fd, ca[0]);
                    cal[1] = axpy(iinv, T1, cal[1]); cal[0] = axpy(-iinv, T0, cal[0]);
                }
            }
        }
        // group B, second part: finger<->cube
        if constexpr (NC == 1) { arm_slot(std::integral_constant<int, 0>{}); arm_slot(std::integral_constant<int, 1>{}); }
#pragma unroll
        for (int c = 0; c < NC; c++) { ca[c] = ca[c] + dcaA[c]; cal[c] = cal[c] + dcalA[c]; }   // group A's share of the cube accelerations (slot 4)
        sweeps_done = it + 1;
        if (ADAPT) {
            if (__all(chg <= P.pgs_tol * (1.f + fmx))) break;
        }
    }

    // ---- keep the forces for the next substep's warm start (inactive slots hold zero) ----------------
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int k = 0; k < 4; k++) W.floor[c][s][k] = FS[c][s].f[k];
#pragma unroll
    for (int s = 0; s < NAS; s++)
#pragma unroll
        for (int k = 0; k < NRW; k++) W.arm[s][k] = AS[s].f[k];
#pragma unroll
    for (int s = 0; s < NCC; s++) W.cc_prev[s] = cc_act[s];
    if (P.diag == 1 || (P.diag == 2 && !NEWTON)) {   // wave-uniform (diagnostics = 2 on the Newton kernels: the same fields carry cycle counts instead, see the solve above)
        unsigned m = 0u;
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int s = 0; s < 4; s++) m |= FS[c][s].act ? (1u << (4 * c + s)) : 0u;
#pragma unroll
        for (int s = 0; s < NCC; s++) {
            if (NC == 2) m |= cc_act[s] ? (1u << (s < 4 ? 8 + s : 24 + (s - 4))) : 0u;
        }
        m |= AS[0].act ? (1u << 12) : 0u; m |= AS[1].act ? (1u << 13) : 0u;
        m |= AS[2].act ? (1u << 14) : 0u; m |= AS[3].act ? (1u << 15) : 0u;
        m |= AS[4].act ? (1u << 16) : 0u;
#pragma unroll
        for (int j = 0; j < 6; j++) m |= lim_act[j] ? (1u << (18 + j)) : 0u;
        DGtot.mask |= m;
        DGtot.count += (unsigned)__popc(m);
        DGtot.sweeps = max(DGtot.sweeps, (unsigned)(m ? sweeps_done : 0));
        DGtot.choice += DG.choice * (unsigned)(2 * sub_index + 1);   // odd weight: the same choice in another substep hashes differently
    }
#pragma unroll
    for (int j = 0; j < 6; j++) W.lim[j] = flim[j];

    // ---- implicitfast: (M + h (damping + kv) I) qacc = qfrc_smooth + J^T f = L y -------------------
    float rhs[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float s = rcp(CL.id[i]) * y[i];  // L_ii y_i
#pragma unroll
        for (int k = 0; k < i; k++) s = fmaf(CL.L[i][k], y[k], s);
        rhs[i] = s;
    }
#pragma unroll
    for (int j = 0; j < 6; j++) Mm[j][j] += H * (DAMPING + KV);
    chol6(Mm, CL);
    fsub(CL, rhs);
    bsub(CL, rhs);
#pragma unroll
    for (int j = 0; j < 6; j++) {
        S.qd[j] = fmaf(H, rhs[j], S.qd[j]);
        S.q[j] = fmaf(H, S.qd[j], S.q[j]);
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        S.cv[c] = axpy(H, ca[c], S.cv[c]);
        // body-frame angular acceleration = R^T alpha_world
        f3 ab = mk(dot(CR[c].X, cal[c]), dot(CR[c].Y, cal[c]), dot(CR[c].Z, cal[c]));
        S.cw[c] = axpy(H, ab, S.cw[c]);
        S.cp[c] = axpy(H, S.cv[c], S.cp[c]);
        // q <- q * exp(h w / 2), normalise  (MuJoCo mju_quatIntegrate)
        f3 w = S.cw[c];
        float wn2 = dot(w, w);
        if (wn2 > 0.f) {
            float iw = rsq(wn2), wn = wn2 * iw;
            float sh, chf;
            sincos_small(0.5f * H * wn, &sh, &chf);
            float s = sh * iw;
            float dq0 = chf, dq1 = w.x * s, dq2 = w.y * s, dq3 = w.z * s;
            float q0 = S.cq[c][0], q1 = S.cq[c][1], q2 = S.cq[c][2], q3 = S.cq[c][3];
            float r0 = q0 * dq0 - q1 * dq1 - q2 * dq2 - q3 * dq3;
            float r1 = q0 * dq1 + q1 * dq0 + q2 * dq3 - q3 * dq2;
            float r2 = q0 * dq2 - q1 * dq3 + q2 * dq0 + q3 * dq1;
            float r3 = q0 * dq3 + q1 * dq2 - q2 * dq1 + q3 * dq0;
            float in = rsq(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
            S.cq[c][0] = r0 * in; S.cq[c][1] = r1 * in; S.cq[c][2] = r2 * in; S.cq[c][3] = r3 * in;
        }
    }
    return CPL == CPL_SLOW ? coupled_any : true;
}

// ------------------------------------------------------------------------------------------------
// the step kernel
// ------------------------------------------------------------------------------------------------
template <int NC, bool EE, bool ADAPT, bool ROLL, bool BIG, bool NEWTON = false>
__global__ __launch_bounds__(64) void lcr_step_kernel(LcrDev P, const float *__restrict__ action) {
    constexpr int CCR = NEWTON ? CC_REC_NEWTON : CC_REC;
    __shared__ float lds[NEWTON ? NEWTON_G_ROWS * LDS_ROW + (NC == 2 ? 8 * CCR * 64 : 0) + coop_floats<NC>() : LdsSize<NC, false, ROLL, BIG>::value];   // (NEWTON: 4 x 6 + 4 g rows = 42 KiB per wave, + eight cube<->cube records of 12 floats = 66 KiB)
    constexpr int NCC = NEWTON ? 8 : 4;
    constexpr int CCB = NEWTON ? NEWTON_G_ROWS : cc_base_rows<NC, BIG, ROLL>();
    const int lane = threadIdx.x;
    const int e_raw = blockIdx.x * 64 + lane;
    const bool valid = e_raw < P.n;
    const int e = valid ? e_raw : P.n - 1;  // tail lanes shadow the last env, their stores are masked
    const int N = P.n;

    const long long t_begin = P.diag == 2 ? clock64() : 0;
    EnvState<NC> S;
    load_state<NC>(P, e, S);
    f3 target = mk(0.f, 0.f, 0.f);
    if (P.has_target) target = mk(P.target[e], P.target[N + e], P.target[2 * N + e]);
    int elapsed = P.elapsed[e];

    // ---- apply_action (reach_cube_env.py:223-273) ------------------------------------------------
    float act[6];
#pragma unroll
    for (int i = 0; i < 6; i++) act[i] = i < P.k ? clampf(action[(size_t)i * N + e], -1.f, 1.f) : 0.f;  // np.clip reach:234
    float ctrl[6];
    f3 lag_ee = mk(0.f, 0.f, 0.f);
    int ik_iters = 0;   // executed iterations of the reference's IK loop (diagnostics)
    if (EE) {
        f3 eel = mk(P.ee_lag[e], P.ee_lag[N + e], P.ee_lag[2 * N + e]);
        f3 tgt = mk(eel.x + act[0] * 0.05f, eel.y + act[1] * 0.05f, fmaxf(eel.z + act[2] * 0.05f, 0.f));  // reach:241-242
        // inverse_kinematics (reach:148-221): fixed 10 iterations with a per-lane "frozen" mask instead of break
        float qk[6], qstate[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { qk[j] = S.q[j]; qstate[j] = S.q[j]; }
        bool done = false;
        for (int it = 0; it < 10; it++) {
            ik_iters += done ? 0 : 1;
            ArmFrames F;
            arm_frames(qk, F);
            f3 site = site_pos(F);
            f3 err = tgt - site;
            if (!done) {
#pragma unroll
                for (int j = 0; j < 6; j++) qstate[j] = qk[j];  // reach:185 writes the sim state (REF-QUIRK-3)
            }
            done = done || (dot(err, err) < 0.01f * 0.01f);  // reach:193
            // translational site Jacobian, site on link_5 => column 6 is zero (reach:197)
            f3 Jc[5];
#pragma unroll
            for (int j = 0; j < 5; j++) Jc[j] = cross(joint_axis(F, j), site - F.p[j]);
            // (J^T J + 0.15 I) qdot = J^T e  (reach:200-202); the 6th equation is 0.15 qdot_6 = 0
            float A[6][6], b[6];
#pragma unroll
            for (int a = 0; a < 5; a++) {
#pragma unroll
                for (int c2 = 0; c2 <= a; c2++) A[a][c2] = dot(Jc[a], Jc[c2]) + (a == c2 ? 0.15f : 0.f);
                b[a] = dot(Jc[a], err);
            }
#pragma unroll
            for (int c2 = 0; c2 < 5; c2++) A[5][c2] = 0.f;
            A[5][5] = 0.15f; b[5] = 0.f;
            Chol6 C;
            chol6(A, C);
            fsub(C, b);
            bsub(C, b);
            float nn = 0.f;
#pragma unroll
            for (int j = 0; j < 6; j++) nn = fmaf(b[j], b[j], nn);
            float scale = nn > 1.f ? rsq(nn) : 1.f;  // reach:210-212
#pragma unroll
            for (int j = 0; j < 6; j++) {
                float qn = clampf(fmaf(b[j] * scale, 0.5f, qk[j]), JLO[j], JHI[j]);  // reach:215, 141-146
                qk[j] = done ? qk[j] : qn;
            }
        }
#pragma unroll
        for (int j = 0; j < 6; j++) { ctrl[j] = qk[j]; S.q[j] = qstate[j]; }
        if (P.gripper_active) ctrl[5] = clampf(S.q[5] + act[3] * 0.2f, JLO[5], JHI[5]);  // lift:253-257
        else ctrl[5] = 0.f;                                                                  // reach:247
    } else {
        const float TLO[6] = {-3.14159f, -1.5708f, -1.48353f, -1.91986f, -2.96706f, -1.74533f};  // reach:249-250
        const float THI[6] = {3.14159f, 1.22173f, 1.74533f, 1.91986f, 2.96706f, 0.0523599f};
#pragma unroll
        for (int j = 0; j < 5; j++) ctrl[j] = clampf(act[j] + S.q[j], TLO[j], THI[j]);
        float ga = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) ga = (i == P.k - 1) ? act[i] : ga;  // lift:264 action[-1]
        ctrl[5] = P.gripper_active ? clampf(ga + S.q[5], TLO[5], THI[5]) : 0.f;
    }

    // ---- n_substeps x mj_step (reach:276-279) -----------------------------------------------------
    f3 lag_cube[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) lag_cube[c] = S.cp[c];
    // Constraint forces for the warm start of the first substep: carried from the previous control step (P.warm: [LCR_NWARM][N],
    // zero after reset / set_state), as MuJoCo carries mjData.qacc_warmstart from one mj_step to the next -- the reference never
    // resets it between env.step calls.  With LCR_COMPAT_COLD_SOLVE_EACH_STEP (P.warm == nullptr) every control step starts from zero.
    Warm<NC, ROLL ? 6 : 4> W;
    const bool carry = P.warm != nullptr;   // wave-uniform
    auto wld = [&](int idx) -> float { return (carry && valid) ? P.warm[(size_t)idx * N + e] : 0.f; };
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int k = 0; k < 4; k++) W.floor[c][s][k] = wld(WARM_FLOOR + 16 * c + 4 * s + k);
#pragma unroll
    for (int s = 0; s < NAS; s++)
#pragma unroll
        for (int k = 0; k < (ROLL ? 6 : 4); k++) W.arm[s][k] = wld(WARM_ARM + 6 * s + k);
#pragma unroll
    for (int j = 0; j < 6; j++) W.lim[j] = wld(WARM_LIM + j);
#pragma unroll
    for (int s = 0; s < 8; s++) W.cc_prev[s] = false;
    if constexpr (NC == 2) {   // cube<->cube forces live in their LDS records between substeps
        float *ccl = lds + CCB * LDS_ROW + lane;
#pragma unroll
        for (int s = 0; s < NCC; s++) {
            W.cc_prev[s] = wld(s < 4 ? WARM_CCPREV + s : WARM_CCPREV2 + (s - 4)) != 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) ccl[(size_t)(s * CCR + 3 + r) * 64] = wld(s < 4 ? WARM_CC + 4 * s + r : WARM_CC2 + 4 * (s - 4) + r);
        }
    }
    Diag DG = {0u, 0u, 0u, 0u};
    if constexpr (NEWTON) {
        // two copies of the substep (see CPL above): the wave runs the fast one while at most coop_max of its lanes are coupled, else the SIMT one
        bool slow = false;   // wave-uniform
        for (int s = 0; s < P.n_substeps; s++) {
            if (!slow) slow = !substep<NC, ADAPT, ROLL, BIG, NEWTON, CPL_FAST>(P, S, ctrl, lds, lane, e, lag_ee, lag_cube, W, DG, s);
            if (slow) slow = substep<NC, ADAPT, ROLL, BIG, NEWTON, CPL_SLOW>(P, S, ctrl, lds, lane, e, lag_ee, lag_cube, W, DG, s);
        }
    } else {
        for (int s = 0; s < P.n_substeps; s++) substep<NC, ADAPT, ROLL, BIG, NEWTON>(P, S, ctrl, lds, lane, e, lag_ee, lag_cube, W, DG, s);
    }
    if (P.diag && valid) {
        P.active_mask[e] = DG.mask; P.active_count[e] = DG.count; P.max_sweeps[e] = DG.sweeps;
        P.choice[e] = DG.choice + (P.diag == 2 ? 0u : (unsigned)ik_iters * 0x9E3779B1u);   // (diagnostics = 2: the field carries counters, not the decision hash)
        if (P.diag == 2) P.max_sweeps[e] = (unsigned)(clock64() - t_begin);   // profiling aid: cycles of this wave up to here
#pragma unroll
        for (int j = 0; j < 6; j++) P.ctrl_out[(size_t)j * N + e] = ctrl[j];   // data.ctrl as apply_action left it (reach_cube_env.py:273)
    }

    // ---- reward / success / termination (reach:313-348 and per-task deltas), lagged kinematics (P8) ----
    f3 a3, b3;
    float reward;
    bool success, terminated;
    {
        const int task = P.task;
        if (task == 0) { a3 = lag_ee; b3 = lag_cube[0]; }
        else if (task == 4) { a3 = lag_cube[NC - 1]; b3 = mk(lag_cube[0].x, lag_cube[0].y, lag_cube[0].z + 0.03f); }
        else if (task == 1) { a3 = lag_ee; b3 = lag_cube[0]; }
        else { a3 = lag_cube[0]; b3 = target; }
        f3 df = a3 - b3;
        float d = sqrtf(dot(df, df));
        if (task == 1) {  // lift:341-345: (cube_z - height_threshold) + distance, never terminates, info = {}
            reward = (lag_cube[0].z - P.height_thr) + d;
            success = false; terminated = false;
        } else {
            success = d < P.dist_thr;
            terminated = success;
            // sparse: -(d > thr) as float32, i.e. -0.0f inside the threshold (reach:345-346); bit pattern built explicitly
            reward = P.reward_type == 0 ? __uint_as_float(0x80000000u | (d > P.dist_thr ? 0x3f800000u : 0u)) : -d;
        }
    }
    // failure containment (MuJoCo's mj_checkPos/mj_checkVel reset an unstable simulation; the analogue here): a state
    // with NaN/inf/huge entries ends the episode as truncated and the env is re-initialised with zero velocity
    bool diverged = false;
#pragma unroll
    for (int j = 0; j < 6; j++) diverged = diverged || bad_value(S.q[j]) || bad_value(S.qd[j]);
#pragma unroll
    for (int c = 0; c < NC; c++) {
        diverged = diverged || bad_value(S.cp[c].x) || bad_value(S.cp[c].y) || bad_value(S.cp[c].z);
        diverged = diverged || bad_value(S.cv[c].x) || bad_value(S.cv[c].y) || bad_value(S.cv[c].z);
        diverged = diverged || bad_value(S.cw[c].x) || bad_value(S.cw[c].y) || bad_value(S.cw[c].z);
#pragma unroll
        for (int k = 0; k < 4; k++) diverged = diverged || bad_value(S.cq[c][k]);
    }
    if (diverged) { reward = -1.0f; success = false; terminated = false; }
    elapsed += 1;
    const bool truncated = diverged || (P.max_steps > 0 && elapsed >= P.max_steps);  // gymnasium TimeLimit
    const bool do_reset = diverged || (P.auto_reset && (terminated || truncated));
    if (valid) {
        P.reward[e] = reward;
        P.terminated[e] = terminated;
        P.truncated[e] = truncated;
        P.is_success[e] = success;
        P.did_reset[e] = do_reset;
    }
    if (do_reset) {  // SB3 VecEnv semantics: keep the terminal observation, then reset in place
        if (valid) {
            write_obs18<NC>(P, P.term_obs, e, S, target);
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) P.term_quat[(size_t)(4 * c + k) * N + e] = S.cq[c][k];   // completes the terminal pose (frames of recordings)
        }
        Pcg g = load_rng(P, e);
        reset_env<NC>(P, S, g, target, lag_ee, 0);
        if (diverged) {
#pragma unroll
            for (int j = 0; j < 6; j++) S.qd[j] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; c++) { S.cv[c] = mk(0.f, 0.f, 0.f); S.cw[c] = mk(0.f, 0.f, 0.f); }
        }
        if (valid) {
            store_rng(P, e, g);
            if (P.has_target) { P.target[e] = target.x; P.target[N + e] = target.y; P.target[2 * N + e] = target.z; }
        }
        elapsed = 0;
    }
    if (valid) {
        store_state<NC>(P, e, S);
        P.elapsed[e] = elapsed;
        P.ee_lag[e] = lag_ee.x; P.ee_lag[N + e] = lag_ee.y; P.ee_lag[2 * N + e] = lag_ee.z;
        if (P.sim_time) P.sim_time[e] = __dadd_rn(P.sim_time[e], (double)P.n_substeps * 0.002);  // data.time advances in mj_step only
    }
    if (carry && valid) {   // forces for the next control step's first substep; an env that was just reset starts from zero
        auto wst = [&](int idx, float v) { P.warm[(size_t)idx * N + e] = do_reset ? 0.f : v; };
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int k = 0; k < 4; k++) wst(WARM_FLOOR + 16 * c + 4 * s + k, W.floor[c][s][k]);
#pragma unroll
        for (int s = 0; s < NAS; s++)
#pragma unroll
            for (int k = 0; k < (ROLL ? 6 : 4); k++) wst(WARM_ARM + 6 * s + k, W.arm[s][k]);
#pragma unroll
        for (int j = 0; j < 6; j++) wst(WARM_LIM + j, W.lim[j]);
        if constexpr (NC == 2) {
            const float *ccl = lds + CCB * LDS_ROW + lane;
#pragma unroll
            for (int s = 0; s < NCC; s++) {
                wst(s < 4 ? WARM_CCPREV + s : WARM_CCPREV2 + (s - 4), W.cc_prev[s] ? 1.f : 0.f);
#pragma unroll
                for (int r = 0; r < 4; r++) wst(s < 4 ? WARM_CC + 4 * s + r : WARM_CC2 + 4 * (s - 4) + r, W.cc_prev[s] ? ccl[(size_t)(s * CCR + 3 + r) * 64] : 0.f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// explicit reset kernel
// ------------------------------------------------------------------------------------------------
template <int NC>
__global__ __launch_bounds__(256) void lcr_reset_kernel(LcrDev P, const unsigned char *mask, const unsigned long long *seeds,
                                                          int seed_from_base, unsigned long long base_seed) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= P.n) return;
    if (mask && !mask[e]) return;
    const int N = P.n;
    EnvState<NC> S;
    load_state<NC>(P, e, S);
    Pcg g;
    if (seeds) g = pcg_seed(seeds[e]);
    else if (seed_from_base) g = pcg_seed(base_seed + (unsigned long long)(P.env_off + e));
    else g = load_rng(P, e);
    f3 target = mk(0.f, 0.f, 0.f), ee;
    if (P.has_target) target = mk(P.target[e], P.target[N + e], P.target[2 * N + e]);
    reset_env<NC>(P, S, g, target, ee, P.walls ? P.goal[e] : 0);
    store_rng(P, e, g);
    store_state<NC>(P, e, S);
    if (P.has_target) { P.target[e] = target.x; P.target[N + e] = target.y; P.target[2 * N + e] = target.z; }
    P.ee_lag[e] = ee.x; P.ee_lag[N + e] = ee.y; P.ee_lag[2 * N + e] = ee.z;
    P.elapsed[e] = 0;
    if (P.warm)
        for (int i = 0; i < LCR_DEV_NWARM; i++) P.warm[(size_t)i * N + e] = 0.f;   // no constraint forces carried into a new episode
}

#if LCR_HAS_PART(0)
// ------------------------------------------------------------------------------------------------
// synthetic policy: U(-1,1) from Philox4x32-10 keyed (seed, global env id, step)
// ------------------------------------------------------------------------------------------------
DEV void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__global__ __launch_bounds__(256) void lcr_fill_actions_kernel(float *action, int n, int k, long long env_off, unsigned long long seed,
                                                                 unsigned long long step) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const unsigned long long gid = (unsigned long long)(env_off + e);
    for (int blk = 0; blk * 4 < k; blk++) {
        uint32_t c[4] = {(uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)step, (uint32_t)(step >> 32) ^ ((uint32_t)blk << 24)};
        uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
        for (int r = 0; r < 10; r++) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            int comp = blk * 4 + i;
            if (comp < k) action[(size_t)comp * n + e] = (float)(c[i] >> 8) * (2.0f / 16777216.0f) - 1.0f;
        }
    }
}

// measurement support: one dword per lane copy (the step kernel's access pattern) with a known byte count,
// used to calibrate rocprofv3 FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md, HBM section)
__global__ __launch_bounds__(64) void lcr_calib_copy_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i < n) dst[i] = src[i] + 1.0f;
}

#endif   // LCR_HAS_PART(0)

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int check_launch() {
    hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : (int)err;
}

// The step kernel has 24 instantiations.  gym_lowcostrobot_amd/build.py compiles this file three times in parallel, -DLCR_PART=0|2|3
// selecting which launchers -- and hence which instantiations -- a translation unit emits: 0 the one-cube kernels (+ the dispatcher and
// the small kernels), 2 / 3 the two StackTwoCubes variants.  Without the macro (tools/) everything is in one unit.  (PushCubeLoop:
// lcr_kernels_loop.hip.)
// the four solver-mode / contact-row variants of one launcher
#define LCR_DISPATCH_MODES(fn)                                              \
    do {                                                                    \
        if (P.pgs_iters < 0) {   /* converged mode */                       \
            if (P.roll) fn<true, true>(P, action_dev, ee_mode, st);         \
            else fn<true, false>(P, action_dev, ee_mode, st);               \
        } else {                                                            \
            if (P.roll) fn<false, true>(P, action_dev, ee_mode, st);   /* six-row finger<->cube contacts */ \
            else fn<false, false>(P, action_dev, ee_mode, st);              \
        }                                                                   \
    } while (0)

int lcr_launch_step_stack(const LcrDev &P, const float *action_dev, int ee_mode, hipStream_t st);
int lcr_launch_step_stack_big(const LcrDev &P, const float *action_dev, int ee_mode, hipStream_t st);

#if LCR_HAS_PART(2)
template <bool ADAPT, bool ROLL>
static void launch_stack_t(const LcrDev &P, const float *action_dev, int ee_mode, hipStream_t st) {
    const int blocks = (P.n + 63) / 64;
    if (!ee_mode) hipLaunchKernelGGL((lcr_step_kernel<2, false, ADAPT, ROLL, false>), dim3(blocks), dim3(64), 0, st, P, action_dev);
    else hipLaunchKernelGGL((lcr_step_kernel<2, true, ADAPT, ROLL, false>), dim3(blocks), dim3(64), 0, st, P, action_dev);
}
int lcr_launch_step_stack(const LcrDev &P, const float *action_dev, int ee_mode, hipStream_t st) {
    LCR_DISPATCH_MODES(launch_stack_t);
    return check_launch();
}
#endif
#if LCR_HAS_PART(3)
template <bool ADAPT, bool ROLL>
static void launch_stack_big_t(const LcrDev &P, const float *action_dev, int ee_mode, hipStream_t st) {   // shard of at most three waves per CU: every g row in LDS
    const int blocks = (P.n + 63) / 64;
    if (!ee_mode) hipLaunchKernelGGL((lcr_step_kernel<2, false, ADAPT, ROLL, true>), dim3(blocks), dim3(64), 0, st, P, action_dev);
    else hipLaunchKernelGGL((lcr_step_kernel<2, true, ADAPT, ROLL, true>), dim3(blocks), dim3(64), 0, st, P, action_dev);
}
int lcr_launch_step_stack_big(const LcrDev &P, const float *action_dev, int ee_mode, hipStream_t st) {
    LCR_DISPATCH_MODES(launch_stack_big_t);
    return check_launch();
}
#endif

#if LCR_HAS_PART(4)
// the faithful preset (lcr_config.solver = LCR_SOLVER_NEWTON): six-row finger contacts everywhere, Newton on the primal
int lcr_launch_step_newton(const LcrDev &P, const float *action_dev, int ee_mode, void *stream) {
    const int blocks = (P.n + 63) / 64;
    const hipStream_t st = (hipStream_t)stream;
    if (!ee_mode) hipLaunchKernelGGL((lcr_step_kernel<1, false, false, true, false, true>), dim3(blocks), dim3(64), 0, st, P, action_dev);
    else hipLaunchKernelGGL((lcr_step_kernel<1, true, false, true, false, true>), dim3(blocks), dim3(64), 0, st, P, action_dev);
    return check_launch();
}
#endif

#if LCR_HAS_PART(5)
int lcr_launch_step_newton_stack(const LcrDev &P, const float *action_dev, int ee_mode, void *stream) {
    const int blocks = (P.n + 63) / 64;
    const hipStream_t st = (hipStream_t)stream;
    if (!ee_mode) hipLaunchKernelGGL((lcr_step_kernel<2, false, false, true, true, true>), dim3(blocks), dim3(64), 0, st, P, action_dev);
    else hipLaunchKernelGGL((lcr_step_kernel<2, true, false, true, true, true>), dim3(blocks), dim3(64), 0, st, P, action_dev);
    return check_launch();
}
#endif

#if LCR_HAS_PART(0)
template <bool ADAPT, bool ROLL>
static void launch_one_cube_t(const LcrDev &P, const float *action_dev, int ee_mode, hipStream_t st) {
    const int blocks = (P.n + 63) / 64;
    if (!ee_mode) hipLaunchKernelGGL((lcr_step_kernel<1, false, ADAPT, ROLL, false>), dim3(blocks), dim3(64), 0, st, P, action_dev);
    else hipLaunchKernelGGL((lcr_step_kernel<1, true, ADAPT, ROLL, false>), dim3(blocks), dim3(64), 0, st, P, action_dev);
}
int lcr_launch_step(const LcrDev &P, const float *action_dev, int ee_mode, void *stream) {
    const hipStream_t st = (hipStream_t)stream;
    if (P.walls) return lcr_launch_step_loop(P, action_dev, ee_mode, stream);   // PushCubeLoop: its own unit and solver (lcr_kernels_loop.hip)
    if (P.newton) return P.task == 4 ? lcr_launch_step_newton_stack(P, action_dev, ee_mode, stream) : lcr_launch_step_newton(P, action_dev, ee_mode, stream);
    if (P.coop && P.pgs_iters >= 0 && P.diag != 2) {   // two cooperating waves per 64 envs (no converged mode, no per-wave cycle read-back)
        if (P.task == 4) return P.cc8 ? lcr_launch_step2_stack_cc8(P, action_dev, ee_mode, P.coop, stream) : lcr_launch_step2_stack(P, action_dev, ee_mode, P.coop, stream);
        return lcr_launch_step2_one_cube(P, action_dev, ee_mode, P.coop, stream);
    }
    if (P.task == 4) return P.big_lds ? lcr_launch_step_stack_big(P, action_dev, ee_mode, st) : lcr_launch_step_stack(P, action_dev, ee_mode, st);
    LCR_DISPATCH_MODES(launch_one_cube_t);
    return check_launch();
}

int lcr_launch_reset(const LcrDev &P, const unsigned char *mask_dev, const unsigned long long *seeds_dev, int seed_from_base,
                     unsigned long long base_seed, void *stream) {
    hipStream_t st = (hipStream_t)stream;
    const int blocks = (P.n + 255) / 256;
    if (P.task == 4) hipLaunchKernelGGL((lcr_reset_kernel<2>), dim3(blocks), dim3(256), 0, st, P, mask_dev, seeds_dev, seed_from_base, base_seed);
    else hipLaunchKernelGGL((lcr_reset_kernel<1>), dim3(blocks), dim3(256), 0, st, P, mask_dev, seeds_dev, seed_from_base, base_seed);
    return check_launch();
}

int lcr_launch_fill_actions(float *action_dev, int n, int k, long long env_off, unsigned long long seed, unsigned long long step,
                            void *stream) {
    hipLaunchKernelGGL(lcr_fill_actions_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, action_dev, n, k, env_off, seed, step);
    return check_launch();
}

int lcr_launch_calib_copy(const float *src, float *dst, size_t n, void *stream) {
    hipLaunchKernelGGL(lcr_calib_copy_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, src, dst, n);
    return check_launch();
}
#endif   // LCR_HAS_PART(0)
