// lcr_kernels2.hip -- the step kernel as TWO COOPERATING WAVES per 64 environments (gfx950).
//
// Why: with one wave per 64 envs (lcr_kernels.hip) a 65 536-env batch puts exactly one wave on each of the 1024 SIMDs, and a lone wave
// issues one instruction per ~5.2 cycles although a SIMD can issue one per ~2 (tools/ubench/valu_issue.hip): the
// launch lasts as long as the serial instruction chain of its slowest wave.  A shard of 32 768 envs (BASELINE configs 4 and 5) even
// leaves half the SIMDs without a wave.  Here a workgroup is two waves that share the 64 envs -- lane l of both waves is env l:
//
//   wave A ("arm")   forward kinematics, joint-space inertia + Cholesky factor L (-> wave B), y = L^-1 tau, the contact rows that touch
//                    only the arm (finger spheres <-> floor, arm-link proxies: slots 2-4; joint limits) and their sweeps, qacc = G^-T (V y),
//                    integration of the arm; the action head (incl. the IK loop of ee mode) and the fused tail (reward, termination,
//                    TimeLimit, auto-reset, write-back);
//   wave B ("cubes") its own forward kinematics, RNE bias + actuation + damping = tau (-> wave A), with ONE cube the finger spheres <-> cube slots 0, 1
//                    (it has the cube state, the kinematics and L; with TWO cubes they are wave A's: this wave then carries two cubes' floor rows and the
//                    cube <-> cube rows), floor <-> cube / cube <-> cube contacts, their rows and
//                    sweeps, the implicit-damping factor G ((M + hD) = G G^T) and V = G^-1 L (-> wave A), integration of the cubes and of its copy of the arm state.
//
// The two row groups touch disjoint unknowns (the arm's scaled acceleration y vs the cube accelerations ca / cal) unless a finger sphere
// or a gripper-body proxy touches a cube, so their sweeps are INDEPENDENT chains in almost every (workgroup, substep) pair and run concurrently.
// When some lane of the 64 couples them (wave-uniform flags exchanged at barrier 1: c01 = a row of wave B touches the arm -- a finger sphere on
// the cube, one-cube tasks; cube4 = a row of wave A touches a cube -- a proxy on a cube, with two cubes also a finger sphere on a cube) the groups
// STILL sweep concurrently (round 4): each works from the shared unknowns as of the start of the sweep plus its own changes, and the changes are
// merged at the end of every sweep (block Jacobi between the groups, Gauss-Seidel inside; the oracle's sweep is defined the same way, orc_params.jacobi;
// DESIGN.md section 4 D1).  Barriers per substep (uncoupled: five; coupled: one more before the first sweep and two per sweep):
//
//   A: FK, M, L -> LDS          -X-  y0 = L^-1 tau; pose <- LDS  -X2-  rows of slots 2-4, limits | cube4  -B1-  sweeps: limits, 2-4      | qacc = G^-T V y -> LDS  -E-  integrate arm
//   B: FK, tau -> LDS           -X-  L <- LDS                    -X2-  slots 0, 1; V, G -> LDS; cube rows | c01 -B1-  sweeps: floor, cc, 0-1    | integrate cubes, pose -> LDS  -E-  integrate arm copy
//
// Every LDS hand-over and the barrier that orders it is listed in DESIGN.md section 3.1b; the rule when changing this file: a place may be
// rewritten only after a barrier that its reader has also passed AFTER reading (tests/test_gpu_parity.py::test_kernel_families_agree_and_are_race_free
// caught the one violation there was -- it shows up as run-to-run differences).
//
// At 32 768 envs per GPU the 1024 waves occupy all 1024 SIMDs (one each, up to 512 registers per lane: variant OCC = 1, what lcr_create
// dispatches for such shards); at 65 536 envs two waves share a SIMD (<= 256 registers per lane, variant OCC = 2: same source, same bits;
// since round 4 faster than the one-wave kernels for every one-cube task -- DESIGN.md section 5; PushCubeLoop: lcr_kernels_loop.hip).
//
// Reference map: identical to lcr_kernels.hip (apply_action reach_cube_env.py:223-273, 20 x mj_step :276-279, reward / termination
// :313-348 and the per-task files, reset :297-311); the arithmetic of every block is the one of lcr_kernels.hip, regrouped by owner.
#include "lcr_step_common.h"

#ifndef LCR_PART
#define LCR_PART (-1)
#endif
#define LCR_HAS_PART(k) (LCR_PART == -1 || LCR_PART == (k))

namespace {

// ---- LDS layout of a workgroup: float index = field * 64 + lane (bank = lane mod 32: conflict-free) ----
//  [0, GR * LDS_ROW)            g rows of the arm-coupled slots: slots 0, 1 written and read by wave B (two cubes: wave A), slots 2-4 by wave A
//  [.., + CC records)           Stack: cube<->cube contact records (wave B only; four, or eight with CC8)
//  POSE: NC * 13 fields         cube pose and velocity at the top of a substep: cp3 cq4 cv3 cw3 (B -> A; written before barrier E, read before X2)
//  ACC : NC * 6 fields          y in coupled sweeps (A <-> B), qacc (A -> B at barrier E); after the last substep: B's diagnostics words
//  FLAG: 1 field                [0] c01 (B -> A), [1] cube4 (A -> B) at barrier 1; after the last substep: do_reset per lane (A -> B)
//  PARK: 16 fields              aref / inv of the finger<->floor slots 2, 3 (wave A only; read once per sweep as 16-B vectors)
//  hand-overs that reuse these areas in phases where they are idle: ctrl + post-IK q (A -> B, once) and tau (B -> A at barrier X) in the
//  rows of slot 0; the Cholesky factor L (A -> B at X, read before X2) in the rows of slots 3, 4; V and G (B -> A at barrier 1) in the rows of
//  slots 0, 1 when those are unused (a finger on a cube: WM0, or a global scratch record where LDS is full -- struct comment); the warm-start share of
//  slots 0, 1 in y (B -> A) and of slot 4 in the cube accelerations (A -> B) in POSE[0..5] / POSE[6..] at barrier 1; the cube accelerations
//  of cube4 sweeps in POSE
template <int NC, bool ROLL, bool CC8 = false, bool WMLDS = false> struct Lds2 {
    static constexpr int GR = ROLL ? 24 : 20;
    static constexpr int G0 = 0;
    static constexpr int CC0 = GR * LDS_ROW;
    static constexpr int POSE0 = CC0 + (NC == 2 ? (CC8 ? 2 : 1) * LDS_CC_FLOATS : 0);   // (CC8: eight cube<->cube contact records)
    static constexpr int ACC0 = POSE0 + NC * 13 * 64;
    static constexpr int FLAG0 = ACC0 + NC * 6 * 64;
    static constexpr int PARK0 = FLAG0 + 64;                 // aref[4] | inv[4] of the finger<->floor slots 2, 3 as 16-B vectors: [slot][aref|inv][lane][4]
    static constexpr bool HAS_PARK = !CC8;                   // (eight cube<->cube records: no room, the constants stay in wave A's registers --
                                                             //  that variant runs one wave per SIMD: 2 x 79.7 KiB per CU)
    // V = G^-1 L and G ((M + hD) = G G^T) in a substep with a finger on a cube (the rows of slots 0, 1 are in use then): the one-wave-per-SIMD build has LDS to spare
    // (two workgroups per CU) and keeps a 42-field place for them; the two-waves-per-SIMD build and CC8 go through a global scratch record instead
    static constexpr int WM0 = PARK0 + (HAS_PARK ? 2 * 2 * 64 * 4 : 0);
    static constexpr int TOTAL = WM0 + (WMLDS ? 42 * 64 : 0);
    // hand-over of the Cholesky factor (21 + 6 floats per lane, wave B -> wave A at barrier X): aliases the g rows of slots 3 and 4,
    // which wave A writes only after it has read the factor
    static constexpr int LFAC0 = G0 + (ROLL ? 16 : 12) * LDS_ROW;
};
// an arm-coupled slot between set-up and sweeps; its forces live in the carried-force registers Wf[slot][row]
template <int NRW>
struct ArmSlot2 {
    f3 n, t1, t2, rc;
    float aref[NRW], inv[NRW];
    float Rn;
    bool act;
};
template <bool ROLL> constexpr int as2_row0(int s) { return ROLL ? (s < 2 ? 6 * s : 12 + 4 * (s - 2)) : 4 * s; }

DEV void wg_barrier() { __syncthreads(); }   // s_waitcnt + s_barrier: LDS (and global) writes before it are visible to the partner wave after it

// smooth joint forces: recursive Newton-Euler bias (zero joint acceleration, base accelerating at -g), passive damping and the position
// actuators (ctrlrange == joint range via inheritrange; joint-level force clamp).  The inertia tensors are applied in the link frames
// (R Ic R^T v as three small products): no world inertias are formed.
DEV void rne_tau(const ArmFrames &F, const f3 (&z)[6], const float (&q)[6], const float (&qd)[6], const float (&ctrl)[6], float (&tau)[6]) {
    using namespace lcrm;
    f3 com[6];
    com[0] = local_point(F, 0, C1x, C1y, C1z); com[1] = local_point(F, 1, C2x, C2y, C2z); com[2] = local_point(F, 2, C3x, C3y, C3z);
    com[3] = local_point(F, 3, C4x, C4y, C4z); com[4] = local_point(F, 4, C5x, C5y, C5z); com[5] = local_point(F, 5, C6x, C6y, C6z);
    const float mass[6] = {M1, M2, M3, M4, M5, M6};
    const float IC[6][6] = {{I1_xx, I1_xy, I1_xz, I1_yy, I1_yz, I1_zz}, {I2_xx, I2_xy, I2_xz, I2_yy, I2_yz, I2_zz}, {I3_xx, I3_xy, I3_xz, I3_yy, I3_yz, I3_zz},
                            {I4_xx, I4_xy, I4_xz, I4_yy, I4_yz, I4_zz}, {I5_xx, I5_xy, I5_xz, I5_yy, I5_yz, I5_zz}, {I6_xx, I6_xy, I6_xz, I6_yy, I6_yz, I6_zz}};
    auto inertia_apply = [&](int i, f3 v) -> f3 {
        const f3 l = mk(dot(F.X[i], v), dot(F.Y[i], v), dot(F.Z[i], v));
        const float ax = fmaf(IC[i][0], l.x, fmaf(IC[i][1], l.y, IC[i][2] * l.z));
        const float ay = fmaf(IC[i][1], l.x, fmaf(IC[i][3], l.y, IC[i][4] * l.z));
        const float az = fmaf(IC[i][2], l.x, fmaf(IC[i][4], l.y, IC[i][5] * l.z));
        return axpy(ax, F.X[i], axpy(ay, F.Y[i], az * F.Z[i]));
    };
    f3 w = mk(0.f, 0.f, 0.f), wd = mk(0.f, 0.f, 0.f), a = mk(0.f, 0.f, GRAV), pprev = mk(0.f, 0.f, 0.f);
    f3 Fi[6], Ni[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        f3 r = F.p[i] - pprev;
        a = a + cross(wd, r) + wxwxr(w, r, dot(w, w));
        f3 zq = qd[i] * z[i];
        wd = wd + cross(w, zq);
        w = w + zq;
        f3 rc = com[i] - F.p[i];
        f3 ac = a + cross(wd, rc) + wxwxr(w, rc, dot(w, w));
        Fi[i] = mass[i] * ac;
        Ni[i] = inertia_apply(i, wd) + cross(w, inertia_apply(i, w));
        pprev = F.p[i];
    }
    f3 f = mk(0.f, 0.f, 0.f), n = mk(0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        f3 nn = Ni[i] + cross(com[i] - F.p[i], Fi[i]);
        if (i < 5) nn = nn + n + cross(F.p[i + 1] - F.p[i], f);
        f = f + Fi[i];
        n = nn;
        const float bias = dot(z[i], n);
        const float c = clampf(ctrl[i], JLO[i], JHI[i]);
        const float fa = clampf(fmaf(KP, c - q[i], -KV * qd[i]), -FRC, FRC);
        tau[i] = fa - DAMPING * qd[i] - bias;
    }
}
// who computes tau: the cube wave.  (Measured for StackTwoCubes, where the cube wave is the longer chain: tau on the arm wave instead
// gives 0.637 against 0.628 ms at 32 768 envs -- the slowest workgroups there are bounded by the cube wave's sweeps, not by its prologue.)
template <int NC> constexpr bool rne_on_arm() { return false; }

// ================================================================================================
// wave A: the arm
// ================================================================================================
template <int NC, bool EE, bool ROLL, bool CC8, bool GW>
DEV void arm_program(const LcrDev &P, const float *__restrict__ action, float *lds, const int lane, const int e, const bool valid) {
    using LL = Lds2<NC, ROLL, CC8, !GW>;
    using namespace lcrm;
    constexpr int NRW = ROLL ? 6 : 4;
    constexpr bool RNE_ON_ARM = rne_on_arm<NC>();
    const int N = P.n;
    float q[6], qd[6];
#pragma unroll
    for (int j = 0; j < 6; j++) { q[j] = P.qpos[j * N + e]; qd[j] = P.qvel[j * N + e]; }
    // ---- apply_action (reach_cube_env.py:223-273) ------------------------------------------------
    float act[6];
#pragma unroll
    for (int i = 0; i < 6; i++) act[i] = i < P.k ? clampf(action[(size_t)i * N + e], -1.f, 1.f) : 0.f;  // np.clip reach:234
    float ctrl[6];
    f3 lag_ee = mk(0.f, 0.f, 0.f);
    int ik_iters = 0;
    if (EE) {
        f3 eel = mk(P.ee_lag[e], P.ee_lag[N + e], P.ee_lag[2 * N + e]);
        f3 tgt = mk(eel.x + act[0] * 0.05f, eel.y + act[1] * 0.05f, fmaxf(eel.z + act[2] * 0.05f, 0.f));  // reach:241-242
        // inverse_kinematics (reach:148-221): fixed 10 iterations with a per-lane "frozen" mask instead of break
        float qk[6], qstate[6];
#pragma unroll
        for (int j = 0; j < 6; j++) { qk[j] = q[j]; qstate[j] = q[j]; }
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
            f3 Jc[5];
#pragma unroll
            for (int j = 0; j < 5; j++) Jc[j] = cross(joint_axis(F, j), site - F.p[j]);
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
        for (int j = 0; j < 6; j++) { ctrl[j] = qk[j]; q[j] = qstate[j]; }
        if (P.gripper_active) ctrl[5] = clampf(q[5] + act[3] * 0.2f, JLO[5], JHI[5]);  // lift:253-257
        else ctrl[5] = 0.f;                                                              // reach:247
    } else {
        const float TLO[6] = {-3.14159f, -1.5708f, -1.48353f, -1.91986f, -2.96706f, -1.74533f};  // reach:249-250
        const float THI[6] = {3.14159f, 1.22173f, 1.74533f, 1.91986f, 2.96706f, 0.0523599f};
#pragma unroll
        for (int j = 0; j < 5; j++) ctrl[j] = clampf(act[j] + q[j], TLO[j], THI[j]);
        float ga = 0.f;
#pragma unroll
        for (int i = 0; i < 6; i++) ga = (i == P.k - 1) ? act[i] : ga;  // lift:264 action[-1]
        ctrl[5] = P.gripper_active ? clampf(ga + q[5], TLO[5], THI[5]) : 0.f;
    }

    if (P.diag && P.diag != 3 && valid) {
#pragma unroll
        for (int j = 0; j < 6; j++) P.ctrl_out[(size_t)j * N + e] = ctrl[j];   // data.ctrl as apply_action left it (reach_cube_env.py:273)
    }
    // carried constraint forces of the arm-coupled slots and the joint limits (LcrDev::warm, see lcr_step_common.h WARM_*)
    const bool carry = P.warm != nullptr;
    auto wld = [&](int idx) -> float { return (carry && valid) ? P.warm[(size_t)idx * N + e] : 0.f; };
    // (the same registers hold a slot's forces during the sweeps: Wf[s][r] is row r of slot s, Wlim[j] the limit force of joint j)
    // (this wave owns slots 2-4: finger<->floor and the arm-link proxies; the finger<->cube slots 0, 1 belong to wave B when there is one cube and to this wave
    //  when there are two: StackTwoCubes' cube wave already carries two cubes' floor rows and the cube<->cube rows -- DESIGN.md section 4 D1 "sweep order")
    constexpr int S0 = NC == 2 ? 0 : 2;   // first arm slot of this wave
    float Wf[NAS][NRW], Wlim[6];
#pragma unroll
    for (int s = S0; s < NAS; s++)
#pragma unroll
        for (int k = 0; k < NRW; k++) Wf[s][k] = wld(WARM_ARM + 6 * s + k);
#pragma unroll
    for (int j = 0; j < 6; j++) Wlim[j] = wld(WARM_LIM + j);

    Diag DGtot = {0u, 0u, 0u, 0u};
    f3 lag_cube[NC];
    float *xpose = lds + LL::POSE0 + lane, *xacc = lds + LL::ACC0 + lane;
    int *xflag = reinterpret_cast<int *>(lds + LL::FLAG0);
    // cube state as wave B last published it (pose + velocity at the top of the substep / after the last one)
    f3 cp[NC], cv[NC], cw[NC];
    float cq[NC][4];
    auto read_pose = [&]() {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const float *pp = xpose + (size_t)c * 13 * 64;
            cp[c] = mk(pp[0], pp[64], pp[128]);
#pragma unroll
            for (int k = 0; k < 4; k++) cq[c][k] = pp[(3 + k) * 64];
            cv[c] = mk(pp[7 * 64], pp[8 * 64], pp[9 * 64]);
            cw[c] = mk(pp[10 * 64], pp[11 * 64], pp[12 * 64]);
        }
    };
    {   // wave B owns the actuation: it needs data.ctrl and, after the IK loop's overwrite of qpos (REF-QUIRK-3), the arm configuration
        float *ph = lds + LL::G0 + lane;   // (g rows of slot 0: not written again before barriers X, X2 of the first substep)
#pragma unroll
        for (int j = 0; j < 6; j++) { ph[j * 64] = ctrl[j]; ph[(6 + j) * 64] = q[j]; }
    }
    wg_barrier();   // E0: wave B has published the initial cube pose
    __builtin_amdgcn_s_setprio(2);   // where two waves share a SIMD the arm wave is the longer chain: it wins the issue arbitration
    bool hot = false;                 // this workgroup has had a coupled substep in this step

    // profiling aid (lcr_config.diagnostics = 3): cycles of this wave in total / waiting at barriers / before barrier 1, coupled substeps
    const bool prof = P.diag == 3;
    long long pf_t0 = prof ? clock64() : 0, pf_wait = 0, pf_pre = 0, pf_mark = 0, pf_wx = 0, pf_we = 0;
    unsigned pf_coupled = 0;
    const float minv = P.cube_minv, iinv = P.cube_iinv;
    for (int sub = 0; sub < P.n_substeps; sub++) {
        Diag DG = {0u, 0u, 0u, 0u};
        if (prof) pf_mark = clock64();
        // ---- position stage --------------------------------------------------------------------------
        ArmFrames F;
        arm_frames(q, F);
        lag_ee = site_pos(F);
        f3 z[6];
        // ---- joint-space inertia: composite rigid bodies referenced to the WORLD ORIGIN + armature; Cholesky factor (-> wave B through LDS) ----
        Chol6 CL;
        {
            f3 v0[6];
#pragma unroll
            for (int j = 0; j < 6; j++) { z[j] = joint_axis(F, j); v0[j] = cross(F.p[j], z[j]); }
            f3 com[6];
            Sym3 Iw[6];
            com[0] = local_point(F, 0, C1x, C1y, C1z); Iw[0] = world_inertia(F.X[0], F.Y[0], F.Z[0], I1_xx, I1_xy, I1_xz, I1_yy, I1_yz, I1_zz);
            com[1] = local_point(F, 1, C2x, C2y, C2z); Iw[1] = world_inertia(F.X[1], F.Y[1], F.Z[1], I2_xx, I2_xy, I2_xz, I2_yy, I2_yz, I2_zz);
            com[2] = local_point(F, 2, C3x, C3y, C3z); Iw[2] = world_inertia(F.X[2], F.Y[2], F.Z[2], I3_xx, I3_xy, I3_xz, I3_yy, I3_yz, I3_zz);
            com[3] = local_point(F, 3, C4x, C4y, C4z); Iw[3] = world_inertia(F.X[3], F.Y[3], F.Z[3], I4_xx, I4_xy, I4_xz, I4_yy, I4_yz, I4_zz);
            com[4] = local_point(F, 4, C5x, C5y, C5z); Iw[4] = world_inertia(F.X[4], F.Y[4], F.Z[4], I5_xx, I5_xy, I5_xz, I5_yy, I5_yz, I5_zz);
            com[5] = local_point(F, 5, C6x, C6y, C6z); Iw[5] = world_inertia(F.X[5], F.Y[5], F.Z[5], I6_xx, I6_xy, I6_xz, I6_yy, I6_yz, I6_zz);
            const float mass[6] = {M1, M2, M3, M4, M5, M6};
            float Mm[6][6];
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
                f3 l = axpy(mc, v0[i], cross(z[i], hc));
                f3 n = symv(Ic, z[i]) + cross(hc, v0[i]);
#pragma unroll
                for (int j = 0; j <= i; j++) Mm[i][j] = dot(z[j], n) + dot(v0[j], l);
                Mm[i][i] += ARMATURE;
            }
            chol6(Mm, CL);
            float *pl = lds + LL::LFAC0 + lane;
            int k = 0;
#pragma unroll
            for (int i = 1; i < 6; i++)
#pragma unroll
                for (int j = 0; j < i; j++) pl[(k++) * 64] = CL.L[i][j];
#pragma unroll
            for (int i = 0; i < 6; i++) pl[(k++) * 64] = CL.id[i];
        }
        float tau_own[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if constexpr (RNE_ON_ARM) rne_tau(F, z, q, qd, ctrl, tau_own);
        if (prof) pf_mark = clock64();
        wg_barrier();   // X: L is in LDS for wave B; wave B's tau (bias + actuation + damping) is in LDS for this wave
        if (prof) { pf_wait += clock64() - pf_mark; pf_wx += clock64() - pf_mark; }
        // y = L^T a  (scaled arm acceleration);  y_smooth = L^-1 tau
        float y[6];
#pragma unroll
        for (int j = 0; j < 6; j++) y[j] = RNE_ON_ARM ? tau_own[j] : lds[LL::G0 + lane + j * 64];   // tau: this wave's (two cubes) or wave B's
        fsub(CL, y);
        read_pose();    // cube pose and velocity at the top of this substep (wave B published it before barrier E of the previous one)
        wg_barrier();   // X2: wave B has read L (its LDS place is reused for contact rows from here on); this wave has read the pose
                        //     (wave B reuses the first fields of the pose area for the finger<->cube slots' warm-start share of y)
        CubeRot CR[NC];
        f3 cww[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            CR[c] = quat_to_cols(cq[c]);
            cww[c] = axpy(cw[c].x, CR[c].X, axpy(cw[c].y, CR[c].Y, cw[c].z * CR[c].Z));
            lag_cube[c] = cp[c];   // P8: body xpos as left behind by this mj_step
        }

        // ---- collision + row set-up of the arm-coupled slots (g rows -> LDS).  The share of the warm-start forces that acts on a
        //      cube is collected in dca / dcal and handed to wave B at barrier 1. ----
        f3 dca[NC], dcal[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) { dca[c] = mk(0.f, 0.f, 0.f); dcal[c] = mk(0.f, 0.f, 0.f); }
        ArmSlot2<NRW> AS[NAS];
        bool slot_any[NAS];
        bool link_on_cube = false, wave_on_cube4 = false;
        int link_nj = 3, link_bi = 0;
        int slot_cube[3] = {0, 0, 0};
        {
        const f3 sph[2] = {local_point(F, 4, SPH0x, SPH0y, SPH0z), local_point(F, 5, SPH1x, SPH1y, SPH1z)};
        const float srad[2] = {SPH0r, SPH1r};
        slot_any[0] = false; slot_any[1] = false;
#pragma unroll
        for (int s = S0; s < NAS; s++) {
            const int sp = s & 1;
            const bool may_cube = s < 2 || s == 4;
            ArmSlot2<NRW> &T = AS[s];
            f3 pos = mk(0.f, 0.f, 0.f), n = mk(0.f, 0.f, 1.f);
            float dist = 1.f;
            int cidx = 0;
            bool oncube = s < 2;
            float invw_link = sp == 0 ? INVW_TRAN_L5 : INVW_TRAN_L6;
            int sel = 0;
            if (s < 2) {
                float bestd = 1e30f;
                bool near_any = false;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const f3 dd = sph[sp] - cp[c];
                    near_any = near_any || dot(dd, dd) < (srad[sp] + 1.7321f * CH) * (srad[sp] + 1.7321f * CH);
                }
                if (__any(near_any))
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const SBHit hit = sphere_box(sph[sp], srad[sp], cp[c], CR[c]);
                    if (hit.dist < bestd) { bestd = hit.dist; cidx = c; n = hit.n; pos = hit.pos; sel = 8 * c + hit.code; }
                }
                dist = bestd;
                slot_cube[sp] = cidx;
            } else if (s < 4) {
                dist = sph[sp].z - srad[sp];
                pos = mk(sph[sp].x, sph[sp].y, 0.5f * dist);
                sel = 0;
            } else if (P.arm_collision) {
                const int plink[5] = {2, 2, 3, 4, 5};
                const float px[5] = {LPX0x, LPX1x, LPX2x, LPX3x, LPX4x}, py[5] = {LPX0y, LPX1y, LPX2y, LPX3y, LPX4y}, pz[5] = {LPX0z, LPX1z, LPX2z, LPX3z, LPX4z};
                const float pr[5] = {LPX0r, LPX1r, LPX2r, LPX3r, LPX4r};
                bool near_any = false;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const f3 dd = F.p[4] - cp[c];
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
                            const SBHit hit = sphere_box(ci, pr[i], cp[c], CR[c]);
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
                link_nj = bi < 2 ? 3 : bi + 2;
                invw_link = bi < 2 ? INVW_TRAN_L3 : (bi == 2 ? INVW_TRAN_L4 : (bi == 3 ? INVW_TRAN_L5 : INVW_TRAN_L6));
            }
            T.act = dist < 0.f;
            if (P.diag) {
                if (may_cube && (s < 2 || oncube)) sel += (n.y < 0.5f && n.y > -0.5f) ? 0 : 16;
                diag_choice(DG, T.act, 12 + s, sel);
            }
            slot_any[s] = __any(T.act) != 0;
            if (s == 4) wave_on_cube4 = __any(T.act && oncube) != 0;
#pragma unroll
            for (int k = 0; k < NRW; k++) { T.aref[k] = 0.f; T.inv[k] = 0.f; }
            if (!slot_any[s]) {   // nobody touches: no force is carried
#pragma unroll
                for (int k = 0; k < NRW; k++) Wf[s][k] = 0.f;
            }
            T.Rn = 1.f; T.n = n; T.t1 = mk(0.f, 1.f, 0.f); T.t2 = mk(-1.f, 0.f, 0.f); T.rc = mk(0.f, 0.f, 0.f);
            if (slot_any[s]) {
                if (s == 4) {
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
                if (may_cube) make_frame(n, T.t1, T.t2);
                auto joint_on = [&](int j) -> bool {
                    if (s < 4) return j < (sp == 0 ? 5 : 6);
                    return j < link_nj;
                };
                f3 cube_p = mk(0.f, 0.f, 0.f), cube_v = mk(0.f, 0.f, 0.f), cube_w = mk(0.f, 0.f, 0.f);
                if (may_cube) {
                    if (NC == 2 && cidx == 1) { cube_p = cp[NC - 1]; cube_v = cv[NC - 1]; cube_w = cww[NC - 1]; }
                    else { cube_p = cp[0]; cube_v = cv[0]; cube_w = cww[0]; }
                    T.rc = pos - cube_p;
                }
                float imp, Kc, Bc;
                if (s < 2) { imp = impedance(dist, D0_FC, DW_FC, 1.0f / W_FC); Kc = K_FC; Bc = B_FC; }
                else if (s < 4) { imp = impedance(dist, D0_FF, DW_FF, 1.0f / W_FF); Kc = K_FF; Bc = B_FF; }
                else { imp = impedance(dist, D0_DEF, DW_DEF, 1.0f / W_DEF); Kc = K_DEF; Bc = B_DEF; }
                float Rn = fmaxf((1.f - imp) * rcp(imp) * (invw_link + ((may_cube && oncube) ? minv : 0.f)), 1e-15f);
                float Rf = Rn * P.inv_impratio;
                float Rt = Rf * (s < 2 ? P.rt_fc : (s < 4 ? RT_FF : P.rt_cube));
                T.Rn = Rn;
                // squared friction coefficients of this slot's rows (finger geoms mu 1.5 / torsional 0.005; a link proxy on the floor mu 1, on a cube the cube's)
                const float m2_tan = s < 2 ? P.mu_fc2 : (s < 4 ? MU_FINGER * MU_FINGER : (oncube ? P.mu_c2 : 1.f));
                const float m2_tors = s < 2 ? P.mu_fct2 : (s < 4 ? MU_TORS * MU_TORS : P.mu_ct2);
                float Ln = 1.f, Lt = 0.f;
                f3 jc[6];
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    const bool lit = s < 4 ? j < (sp == 0 ? 5 : 6) : true;
                    jc[j] = lit ? cross(z[j], pos - F.p[j]) : mk(0.f, 0.f, 0.f);
                    if (s == 4 && j >= 3 && !joint_on(j)) jc[j] = mk(0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int r = 0; r < as_rows<ROLL>(s); r++) {
                    // (slot 4: a link proxy on the FLOOR has three rows, condim 3; the torsion row exists only against a cube.  When no lane of the wave has its
                    //  proxy on a cube -- almost always -- the row is skipped: it would contribute exact zeros)
                    if (s == 4 && r == 3 && !wave_on_cube4) { Wf[s][r] = 0.f; continue; }   // (its carried force restarts from zero, as for every row that is off)
                    f3 d = r == 0 ? T.n : (r == 1 ? T.t1 : (r == 2 ? T.t2 : T.n));
                    if constexpr (ROLL) { if (r >= 4) d = r == 4 ? T.t1 : T.t2; }
                    float g[6];
                    float vel = 0.f;
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        g[j] = r < 3 ? dot(jc[j], d) : (joint_on(j) ? dot(z[j], d) : 0.f);
                        vel = fmaf(g[j], qd[j], vel);
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
                    fsub(CL, g);
                    float gg = 0.f;
#pragma unroll
                    for (int j = 0; j < 6; j++) gg = fmaf(g[j], g[j], gg);
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const float2v gp = {g[2 * k], g[2 * k + 1]};
                        *reinterpret_cast<float2v *>(&lds[LL::G0 + (as2_row0<ROLL>(s) + r) * LDS_ROW + k * 128 + lane * 2]) = gp;
                    }
                    float Rr = r == 0 ? Rn : (r == 3 ? Rt : Rf);
                    if (ROLL && r > 3) Rr = Rf * P.rr_fc;
                    T.aref[r] = -Bc * vel - (r == 0 ? Kc * imp * dist : 0.f);
                    const bool row_on = T.act && (s != 4 || r < 3 || oncube);
                    // metric of the block step (soc_step): Ln = 2 (A + R)_nn, Lt = 2 sum mu_j^2 (A + R)_jj over the friction rows that are on
                    {
                        const float arr = gg + diagc + Rr;
                        const float m2r = r == 0 ? 1.f : (r < 3 ? m2_tan : (r == 3 ? m2_tors : P.mu_fcr2));
                        const float KF = 2.f;
                    if (r == 0) Ln = KF * arr;
                    else Lt = fmaf(row_on ? KF * m2r : 0.f, arr, Lt);
                    }
                    const float fw = row_on ? Wf[s][r] : 0.f;
                    Wf[s][r] = fw;
#pragma unroll
                    for (int j = 0; j < 6; j++) y[j] = fmaf(g[j], fw, y[j]);
                    if (may_cube) {
                        const float fc = (s == 4 && !oncube) ? 0.f : fw;
                        f3 dl = r < 3 ? (-minv * fc) * d : mk(0.f, 0.f, 0.f);
                        f3 da = r < 3 ? (-iinv * fc) * cross(T.rc, d) : (-iinv * fc) * d;
                        if (NC == 2 && cidx == 1) { dca[NC - 1] = dca[NC - 1] + dl; dcal[NC - 1] = dcal[NC - 1] + da; }
                        else { dca[0] = dca[0] + dl; dcal[0] = dcal[0] + da; }
                    }
                }
                {   // k[] of soc_step: iLn, mu_tan^2 iLt, w, mu_tors^2 iLt
                    const float iLn = T.act ? rcp(Ln) : 0.f, iLt = T.act ? rcp(Lt) : 0.f, iLs = iLt;
                    T.inv[0] = iLn; T.inv[1] = m2_tan * iLt; T.inv[2] = Ln * rcp(Ln + Lt); T.inv[3] = (s != 4 || oncube) ? m2_tors * iLs : 0.f;   // (a link proxy on the floor has no torsion row: condim 3)
                    if constexpr (ROLL) { T.inv[4] = P.mu_fcr2 * iLs; T.inv[5] = 0.f; }
                }
                if (LL::HAS_PARK && (s == 2 || s == 3)) {   // park the per-substep constants of the finger<->floor slots (read back once per sweep)
                    float4v *pk = reinterpret_cast<float4v *>(lds + LL::PARK0) + (size_t)((s - 2) * 2) * 64 + lane;
                    pk[0] = float4v{T.aref[0], T.aref[1], T.aref[2], T.aref[3]};
                    pk[64] = float4v{T.inv[0], T.inv[1], T.inv[2], T.inv[3]};
                }
            }
        }
        }
        // ---- joint limits: rows +-e_j.  g_j = L^-1 (sg e_j), regulariser and reference acceleration once per substep ----
        bool lim_act[6];
        unsigned lim_wave = 0u;   // bit j: joint j is beyond a limit in SOME lane (wave-uniform).  Under random actions that is joint 0 in ~40 % of the
                                  // waves and the others almost never; a joint no lane needs contributes exact zeros, so its rows are skipped.
#pragma unroll
        for (int j = 0; j < 6; j++) {
            lim_act[j] = (q[j] < JLO[j]) || (q[j] > JHI[j]);
            if (P.diag) diag_choice(DG, lim_act[j], 18 + j, q[j] < JLO[j] ? 0 : 1);
            Wlim[j] = lim_act[j] ? Wlim[j] : 0.f;
            lim_wave |= __any(lim_act[j]) ? (1u << j) : 0u;
        }
        const bool wave_lim = lim_wave != 0u;
        float glim[6][6], lim_aref[6], lim_R[6], lim_inv[6];
        if (wave_lim) {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                if (!((lim_wave >> j) & 1u)) continue;
                const bool lower = q[j] < JLO[j];
                const float sg = lower ? 1.f : -1.f;
                const float pos = lower ? q[j] - JLO[j] : JHI[j] - q[j];
                const float imp = impedance(pos, D0_DEF, DW_DEF, 1.0f / W_DEF);
                lim_R[j] = fmaxf((1.f - imp) * rcp(imp) * INVW_DOF[j], 1e-15f);
                lim_aref[j] = -B_DEF * sg * qd[j] - K_DEF * imp * pos;
#pragma unroll
                for (int k = 0; k < 6; k++) glim[j][k] = k == j ? sg : 0.f;
                fsub(CL, glim[j]);
                float gg = 0.f;
#pragma unroll
                for (int k = 0; k < 6; k++) { gg = fmaf(glim[j][k], glim[j][k], gg); y[k] = fmaf(glim[j][k], Wlim[j], y[k]); }   // (+ warm-start force)
                lim_inv[j] = rcp(gg + lim_R[j]);
            }
        }
        const bool wave_arm = slot_any[0] || slot_any[1] || slot_any[2] || slot_any[3] || slot_any[4];

        // ---- does a gripper-body proxy of some lane touch a cube?  (wave-uniform; with wave B's "a finger sphere touches a cube" it
        //      decides the sweep schedule of BOTH waves) ----
        const bool cube4 = wave_on_cube4;
        // two cubes: the finger<->cube slots are this wave's too -- towards wave B they behave exactly like a proxy on a cube (this wave changes cube accelerations)
        const bool cubeA = cube4 || (NC == 2 && (slot_any[0] || slot_any[1]));
        if (lane == 0) xflag[1] = cubeA ? 1 : 0;
        if (cubeA) {   // share of the warm-start forces of this wave's slots that acts on a cube
#pragma unroll
            for (int c = 0; c < NC; c++) {
                // (pose area behind wave B's dy01: the ACC area may be overwritten by this wave's y hand-over before wave B has read it)
                float *pa = xpose + (size_t)(6 + c * 6) * 64;
                pa[0] = dca[c].x; pa[64] = dca[c].y; pa[128] = dca[c].z;
                pa[192] = dcal[c].x; pa[256] = dcal[c].y; pa[320] = dcal[c].z;
            }
        }
        if (prof) { const long long t = clock64(); pf_pre += t - pf_mark; pf_mark = t; }
        wg_barrier();   // B1
        if (prof) pf_wait += clock64() - pf_mark;
        const bool c01 = __builtin_amdgcn_readfirstlane(xflag[0]) != 0;   // wave B: a finger sphere touches a cube in some lane
        const bool coupled = c01 || cubeA;
        if (coupled && !hot) { hot = true; __builtin_amdgcn_s_setprio(3); }   // (see wave B)
        if (prof) pf_coupled += coupled ? 1u : 0u;
        if (c01) {   // warm-start forces of the finger<->cube slots act on the arm too: wave B's sum of g_r f_r
#pragma unroll
            for (int j = 0; j < 6; j++) y[j] += xpose[j * 64];
        }

        // ---- sweeps ----
        auto limit_rows = [&]() {
            if (wave_lim) {
#pragma unroll
                for (int j = 0; j < 6; j++) {
                    if (!((lim_wave >> j) & 1u)) continue;
                    float gy = 0.f;
#pragma unroll
                    for (int k = 0; k < 6; k++) gy = fmaf(glim[j][k], y[k], gy);
                    const float res = gy - lim_aref[j] + lim_R[j] * Wlim[j];
                    const float nf = fmaxf(Wlim[j] - res * lim_inv[j], 0.f);
                    const float dl = lim_act[j] ? nf - Wlim[j] : 0.f;
                    Wlim[j] += dl;
#pragma unroll
                    for (int k = 0; k < 6; k++) y[k] = fmaf(glim[j][k], dl, y[k]);
                }
            }
        };
        // one Gauss-Seidel pass over the arm-coupled slots.  CPL = false: no lane touches a cube -- slots 0, 1 are off in every lane
        // and the proxy slot is a floor contact everywhere, so every cube term drops out at compile time.
        f3 ca[NC], cal[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) { ca[c] = mk(0.f, 0.f, 0.f); cal[c] = mk(0.f, 0.f, 0.f); }
        auto arm_rows = [&](auto cpl_tag, auto lo_tag, auto hi_tag) {
            constexpr bool CPL = decltype(cpl_tag)::value;
            constexpr int S_LO = decltype(lo_tag)::value, S_HI = decltype(hi_tag)::value;
            if (!wave_arm) return;
#pragma unroll
            for (int s = S_LO; s < S_HI; s++) {
                if (!slot_any[s]) continue;
                ArmSlot2<NRW> &T = AS[s];
                const bool may_cube = CPL && (s < 2 || s == 4);
                const bool oncube = s < 2 || (s == 4 && link_on_cube);
                const int nrow = (ROLL && s < 2) ? 6 : ((s == 4 && !cube4) ? 3 : 4);
                const float Rf = T.Rn * P.inv_impratio;
                const float Rt = Rf * (s < 2 ? P.rt_fc : (s < 4 ? RT_FF : P.rt_cube));
                float2v g[NRW][3];
#pragma unroll
                for (int r = 0; r < nrow; r++)
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        g[r][k] = *reinterpret_cast<const float2v *>(&lds[LL::G0 + (as2_row0<ROLL>(s) + r) * LDS_ROW + k * 128 + lane * 2]);
                float2v yp[3] = {{y[0], y[1]}, {y[2], y[3]}, {y[4], y[5]}};
                float f_in[NRW];
#pragma unroll
                for (int r = 0; r < NRW; r++) f_in[r] = Wf[s][r];
                // slots 2, 3: reference accelerations and inverse diagonals come back from their LDS parking place
                float arefv[NRW], invv[NRW];
#pragma unroll
                for (int r = 0; r < NRW; r++) { arefv[r] = T.aref[r]; invv[r] = T.inv[r]; }
                if (LL::HAS_PARK && (s == 2 || s == 3)) {
                    const float4v *pk = reinterpret_cast<const float4v *>(lds + LL::PARK0) + (size_t)((s - 2) * 2) * 64 + lane;
                    const float4v a4 = pk[0], i4 = pk[64];
                    arefv[0] = a4.x; arefv[1] = a4.y; arefv[2] = a4.z; arefv[3] = a4.w;
                    invv[0] = i4.x; invv[1] = i4.y; invv[2] = i4.z; invv[3] = i4.w;
                }
                f3 a_lin = mk(0.f, 0.f, 0.f), a_ang = mk(0.f, 0.f, 0.f);
                const bool second = may_cube && NC == 2 && slot_cube[s == 4 ? 2 : (s & 1)] == 1;
                if (may_cube) { a_lin = second ? ca[NC - 1] : ca[0]; a_ang = second ? cal[NC - 1] : cal[0]; }
                const float minv_e = (s == 4 && !oncube) ? 0.f : minv, iinv_e = (s == 4 && !oncube) ? 0.f : iinv;
                float vq[3] = {0.f, 0.f, 0.f}, wn = 0.f, w1 = 0.f, w2 = 0.f;
                if (may_cube) {
                    const f3 Ac = a_lin + cross(a_ang, T.rc);
                    vq[0] = dot(T.n, Ac); vq[1] = dot(T.t1, Ac); vq[2] = dot(T.t2, Ac);
                    wn = dot(T.n, a_ang);
                    if (nrow == 6) { w1 = dot(T.t1, a_ang); w2 = dot(T.t2, a_ang); }
                    if (s == 4) {
#pragma unroll
                        for (int i = 0; i < 3; i++) vq[i] = oncube ? vq[i] : 0.f;
                        wn = oncube ? wn : 0.f;
                    }
                }
                // gradient rows of the block from the SAME forces (no serial dependence inside the block), one projected-gradient step, then the change goes to y
                float u[NRW], fcur[NRW], nf[NRW];
#pragma unroll
                for (int r = 0; r < NRW; r++) {
                    fcur[r] = Wf[s][r];
                    u[r] = 0.f;
                    if (r < nrow) {
                        const float2v acc = g[r][0] * yp[0] + g[r][1] * yp[1] + g[r][2] * yp[2];
                        const float gy = acc.x + acc.y;
                        float jc_a = may_cube ? (r < 3 ? -vq[r] : -wn) : 0.f;
                        float Rr = r == 0 ? T.Rn : (r == 3 ? Rt : Rf);
                        if (ROLL && r > 3) { jc_a = may_cube ? (r == 4 ? -w1 : -w2) : 0.f; Rr = Rf * P.rr_fc; }
                        u[r] = gy + jc_a - arefv[r] + Rr * fcur[r];
                    }
                }
                {
                    const float imu2 = s < 4 ? 1.f / (MU_FINGER * MU_FINGER) : (oncube ? P.inv_mu_c2 : 1.f);
                    const float imt2 = s < 2 ? P.inv_mu_fct2 : (s < 4 ? 1.f / (MU_TORS * MU_TORS) : P.inv_mu_ct2);
                    soc_step<NRW>(fcur, u, invv, imu2, imt2, P.inv_mu_fcr2, nrow, nf);
                }
#pragma unroll
                for (int r = 0; r < NRW; r++) {
                    if (r < nrow) {
                        const float dlt = nf[r] - fcur[r];
                        Wf[s][r] = nf[r];
                        const float2v d2 = {dlt, dlt};
#pragma unroll
                        for (int k = 0; k < 3; k++) yp[k] = g[r][k] * d2 + yp[k];
                    }
                }
                y[0] = yp[0].x; y[1] = yp[0].y; y[2] = yp[1].x; y[3] = yp[1].y; y[4] = yp[2].x; y[5] = yp[2].y;
                if (may_cube) {
                    const float e0 = Wf[s][0] - f_in[0], e1 = Wf[s][1] - f_in[1], e2 = Wf[s][2] - f_in[2], e3 = Wf[s][3] - f_in[3];
                    const f3 Fd = axpy(e0, T.n, axpy(e1, T.t1, e2 * T.t2));
                    const f3 dl_lin = (-minv_e) * Fd;
                    f3 Td = axpy(e3, T.n, cross(T.rc, Fd));
                    if constexpr (ROLL) { if (nrow == 6) Td = axpy(Wf[s][4] - f_in[4], T.t1, axpy(Wf[s][5] - f_in[5], T.t2, Td)); }
                    const f3 dl_ang = (-iinv_e) * Td;
                    if (NC == 2 && second) { ca[NC - 1] = ca[NC - 1] + dl_lin; cal[NC - 1] = cal[NC - 1] + dl_ang; }
                    else { ca[0] = ca[0] + dl_lin; cal[0] = cal[0] + dl_ang; }
                }
            }
        };
        using I2 = std::integral_constant<int, 2>; using I5 = std::integral_constant<int, NAS>;
        using I4 = std::integral_constant<int, 4>; using I0 = std::integral_constant<int, 0>;
        auto bar = [&]() {
            if (prof) pf_mark = clock64();
            wg_barrier();
            if (prof) pf_wait += clock64() - pf_mark;
        };
        if (!coupled) {
            for (int it = 0; it < P.pgs_iters; it++) {
                limit_rows();
                arm_rows(std::false_type{}, I2{}, I5{});
            }
        } else {
            // A finger sphere (c01) or a gripper-body proxy (cube4) of some lane touches a cube: the two waves' row groups now share unknowns -- the arm acceleration y
            // (wave B's slots 0, 1) and / or the cube accelerations (this wave's slot 4).  The groups still sweep CONCURRENTLY: each works from the accelerations as
            // they were at the start of the sweep plus its OWN changes, and the changes are merged at the end of the sweep (block Jacobi between the two groups --
            // for two blocks of a positive definite problem that always converges; Gauss-Seidel inside each group; the oracle's sweep is defined the same way,
            // orc_params.jacobi).  Until round 4 such a sweep was serialised (y and the cube accelerations visited the other wave between its row groups), which made
            // the workgroups with a finger on a cube the slowest of every launch.  Hand-overs, single-buffered between barriers:
            //   ACC[0..5]          YA   A -> B  this wave's y after its rows (c01) -- then, B -> A, CAN: the merged cube accelerations (cube4; wave B writes them after it
            //                                   has read YA, this wave never reads YA)
            //   POSE[0..5]         DYB  B -> A  change of y by slots 0, 1 (c01)
            //   POSE[6..6+6 NC)    DCA  A -> B  change of the cube accelerations by slot 4 (cube4)
            // Merged values are formed by the same expression on both sides (yA + DYB, caB + DCA): the two waves' copies of y stay bit-identical.
            // Before the first sweep the same fields carry the start values the other way round (y: A -> B in POSE[0..5], cube accelerations: B -> A in ACC).
            // Rule that makes this race-free without further barriers: a field is only ever OVERWRITTEN by the wave that last READ it.
            float *xdyb = xpose, *xdca = xpose + (size_t)6 * 64;
            if (c01) {   // (this wave has read dy01 from these fields just above)
#pragma unroll
                for (int j = 0; j < 6; j++) xdyb[j * 64] = y[j];
            }
            bar();   // P0: y at the start of the first sweep -> wave B; wave B's cube accelerations -> this wave
            if (cubeA) {
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const float *pa = xacc + (size_t)c * 6 * 64;
                    ca[c] = mk(pa[0], pa[64], pa[128]); cal[c] = mk(pa[192], pa[256], pa[320]);
                }
            }
            for (int it = 0; it < P.pgs_iters; it++) {
                f3 ca_in[NC], cal_in[NC];
#pragma unroll
                for (int c = 0; c < NC; c++) { ca_in[c] = ca[c]; cal_in[c] = cal[c]; }
                limit_rows();
                if constexpr (NC == 2) arm_rows(std::true_type{}, I0{}, I2{});   // (two cubes: finger<->cube, first of this wave's slots as in the oracle's row order)
                arm_rows(std::false_type{}, I2{}, I4{});
                if (cube4) arm_rows(std::true_type{}, I4{}, I5{}); else arm_rows(std::false_type{}, I4{}, I5{});
                if (c01) {
#pragma unroll
                    for (int j = 0; j < 6; j++) xacc[j * 64] = y[j];
                }
                if (cubeA) {   // what this wave's slots changed
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        float *pa = xdca + (size_t)c * 6 * 64;
                        const f3 dl = ca[c] - ca_in[c], da = cal[c] - cal_in[c];
                        pa[0] = dl.x; pa[64] = dl.y; pa[128] = dl.z; pa[192] = da.x; pa[256] = da.y; pa[320] = da.z;
                    }
                }
                bar();   // W: everything of this sweep is written
                if (c01) {
#pragma unroll
                    for (int j = 0; j < 6; j++) y[j] += xdyb[j * 64];
                }
                bar();   // R: everything is read; with cube4 wave B has left the merged cube accelerations in ACC
                if (cubeA) {
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const float *pa = xacc + (size_t)c * 6 * 64;
                        ca[c] = mk(pa[0], pa[64], pa[128]); cal[c] = mk(pa[192], pa[256], pa[320]);
                    }
                }
            }
        }

        // (the forces stay in Wf / Wlim for the next substep's warm start; slots nobody touched were zeroed at set-up)
        if (P.diag) {
            unsigned m = 0u;
            m |= AS[2].act ? (1u << 14) : 0u; m |= AS[3].act ? (1u << 15) : 0u;   // (bits 12, 13: wave B; two cubes: here)
            if constexpr (NC == 2) { m |= AS[0].act ? (1u << 12) : 0u; m |= AS[1].act ? (1u << 13) : 0u; }
            m |= AS[4].act ? (1u << 16) : 0u;
#pragma unroll
            for (int j = 0; j < 6; j++) m |= lim_act[j] ? (1u << (18 + j)) : 0u;
            DGtot.mask |= m;
            DGtot.count += (unsigned)__popc(m);
            DGtot.choice += DG.choice * (unsigned)(2 * sub + 1);
        }

        // ---- implicitfast: (M + h (damping + kv) I) qacc = qfrc_smooth + J^T f = L y, i.e. qacc = G^-T (V y) with wave B's G (G G^T = M + hD) and V = G^-1 L ----
        float qacc[6];
        // wave B left V = G^-1 L and G ((M + hD) = G G^T) before barrier 1: in the idle LDS rows of slots 0, 1, or (a finger on a cube) in their own LDS
        // place / this lane's global scratch record.  qacc = G^-T (V y).
        auto final_solve = [&](auto glob_tag) {
            constexpr bool GLOB = decltype(glob_tag)::value;
            if (GLOB) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const float *pw = lds + ((c01 || NC == 2) ? LL::WM0 : LL::G0) + lane;   // (two cubes: the rows of slots 0, 1 are this wave's, V and G always travel apart)
            const float *gw = P.scratch + ((size_t)blockIdx.x * 64 + lane) * 48;
            auto get = [&](int f) -> float { return GLOB ? gw[f] : pw[f * 64]; };
#pragma unroll
            for (int i = 0; i < 6; i++) {
                float sacc = 0.f;
#pragma unroll
                for (int k = 0; k <= i; k++) sacc = fmaf(get(i * (i + 1) / 2 + k), y[k], sacc);
                qacc[i] = sacc;
            }
#pragma unroll
            for (int i = 5; i >= 0; i--) {   // back substitution with G^T
                float sacc = qacc[i];
#pragma unroll
                for (int k = i + 1; k < 6; k++) sacc = fmaf(-get(21 + k * (k - 1) / 2 + i), qacc[k], sacc);
                qacc[i] = sacc * get(36 + i);
            }
        };
        if (GW && (c01 || NC == 2)) final_solve(std::true_type{}); else final_solve(std::false_type{});
#pragma unroll
        for (int j = 0; j < 6; j++) xacc[j * 64] = qacc[j];   // wave B integrates its copy of the arm state with the same values
        if (prof) pf_mark = clock64();
        wg_barrier();   // E: wave B has integrated the cubes and published their new pose
        if (prof) { pf_wait += clock64() - pf_mark; pf_we += clock64() - pf_mark; }
#pragma unroll
        for (int j = 0; j < 6; j++) {
            qd[j] = fmaf(H, qacc[j], qd[j]);
            q[j] = fmaf(H, qd[j], q[j]);
        }
    }
    if (prof && valid) {
        P.active_mask[e] = (unsigned)(clock64() - pf_t0); P.active_count[e] = (unsigned)pf_wait;
        P.max_sweeps[e] = (unsigned)pf_pre; P.choice[e] = pf_coupled;
        P.ctrl_out[(size_t)2 * N + e] = (float)pf_wx; P.ctrl_out[(size_t)3 * N + e] = (float)pf_we;
    }

    // ================= fused tail (wave A): reward / termination / TimeLimit / auto-reset / write-back =================
    read_pose();   // final cube state
    // wave B's diagnostics words (written after the last barrier E, before T1)
    wg_barrier();   // T1
    if (P.diag && !prof && valid) {
        const unsigned *xb = reinterpret_cast<const unsigned *>(xacc);
        const unsigned bmask = xb[0], bcount = xb[64], bchoice = xb[128];
        const unsigned mask = DGtot.mask | bmask;
        P.active_mask[e] = mask; P.active_count[e] = DGtot.count + bcount;
        P.max_sweeps[e] = mask ? (unsigned)P.pgs_iters : 0u;
        P.choice[e] = DGtot.choice + bchoice + (unsigned)ik_iters * 0x9E3779B1u;
    }
    // (values only the tail needs are loaded here, not carried through the substep loop)
    f3 target = mk(0.f, 0.f, 0.f);
    if (P.has_target) target = mk(P.target[e], P.target[N + e], P.target[2 * N + e]);
    int elapsed = P.elapsed[e];
    EnvState<NC> S;
#pragma unroll
    for (int j = 0; j < 6; j++) { S.q[j] = q[j]; S.qd[j] = qd[j]; }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        S.cp[c] = cp[c]; S.cv[c] = cv[c]; S.cw[c] = cw[c];
#pragma unroll
        for (int k = 0; k < 4; k++) S.cq[c][k] = cq[c][k];
    }
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
        if (task == 1) {
            reward = (lag_cube[0].z - P.height_thr) + d;
            success = false; terminated = false;
        } else {
            success = d < P.dist_thr;
            terminated = success;
            reward = P.reward_type == 0 ? __uint_as_float(0x80000000u | (d > P.dist_thr ? 0x3f800000u : 0u)) : -d;
        }
    }
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
    const bool truncated = diverged || (P.max_steps > 0 && elapsed >= P.max_steps);
    const bool do_reset = diverged || (P.auto_reset && (terminated || truncated));
    reinterpret_cast<int *>(lds + LL::FLAG0)[lane] = do_reset ? 1 : 0;   // wave B zeroes its carried forces where the env was reset
    wg_barrier();   // T2
    if (valid) {
        P.reward[e] = reward;
        P.terminated[e] = terminated;
        P.truncated[e] = truncated;
        P.is_success[e] = success;
        P.did_reset[e] = do_reset;
    }
    if (do_reset) {
        if (valid) {
            write_obs18<NC>(P, P.term_obs, e, S, target);
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int k = 0; k < 4; k++) P.term_quat[(size_t)(4 * c + k) * N + e] = S.cq[c][k];
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
        if (P.sim_time) P.sim_time[e] = __dadd_rn(P.sim_time[e], (double)P.n_substeps * 0.002);
    }
    if (carry && valid) {
        auto wst = [&](int idx, float v) { P.warm[(size_t)idx * N + e] = do_reset ? 0.f : v; };
#pragma unroll
        for (int s = S0; s < NAS; s++)
#pragma unroll
            for (int k = 0; k < NRW; k++) wst(WARM_ARM + 6 * s + k, Wf[s][k]);
#pragma unroll
        for (int j = 0; j < 6; j++) wst(WARM_LIM + j, Wlim[j]);
    }
}

// ================================================================================================
// wave B: the cubes
// ================================================================================================
template <int NC, bool EE, bool ROLL, bool CC8, bool GW>
DEV void cube_program(const LcrDev &P, float *lds, const int lane, const int e, const bool valid) {
    using LL = Lds2<NC, ROLL, CC8, !GW>;
    // cube<->cube manifold points kept (Stack): 4 = the extremes along the diagonals of the reference face; CC8 (lcr_config.cc_points = 8,
    // as many as MuJoCo's mjc_BoxBox may return): also the extremes along its two axes -- narrows deviation D5
    constexpr int NCC = CC8 ? 8 : 4;
    constexpr bool RNE_ON_ARM = rne_on_arm<NC>();
    using namespace lcrm;
    const int N = P.n;
    // this wave's copy of the arm configuration (for the joint-space inertia): it starts from the configuration wave A leaves behind
    // after apply_action (ee mode: the IK overwrite of qpos, REF-QUIRK-3, arrives through LDS) and is integrated with the same qacc
    float q[6], qd[6];
#pragma unroll
    for (int j = 0; j < 6; j++) { q[j] = P.qpos[j * N + e]; qd[j] = P.qvel[j * N + e]; }
    f3 cp[NC], cv[NC], cw[NC];
    float cq[NC][4];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const float *qp = P.qpos + (size_t)(6 + 7 * c) * N + e;
        const float *qv = P.qvel + (size_t)(6 + 6 * c) * N + e;
        cp[c] = mk(qp[0], qp[N], qp[2 * N]);
#pragma unroll
        for (int k = 0; k < 4; k++) cq[c][k] = qp[(3 + k) * N];
        cv[c] = mk(qv[0], qv[N], qv[2 * N]);
        cw[c] = mk(qv[3 * N], qv[4 * N], qv[5 * N]);
        // quaternions are kept normalised from here on (the integration below normalises; this covers states set from outside)
        const float n2 = cq[c][0] * cq[c][0] + cq[c][1] * cq[c][1] + cq[c][2] * cq[c][2] + cq[c][3] * cq[c][3];
        const float in = rsq(n2);
#pragma unroll
        for (int k = 0; k < 4; k++) cq[c][k] *= in;
    }
    float *xpose = lds + LL::POSE0 + lane, *xacc = lds + LL::ACC0 + lane;
    const int *xflag = reinterpret_cast<const int *>(lds + LL::FLAG0);
    auto publish_pose = [&]() {
#pragma unroll
        for (int c = 0; c < NC; c++) {
            float *pp = xpose + (size_t)c * 13 * 64;
            pp[0] = cp[c].x; pp[64] = cp[c].y; pp[128] = cp[c].z;
#pragma unroll
            for (int k = 0; k < 4; k++) pp[(3 + k) * 64] = cq[c][k];
            pp[7 * 64] = cv[c].x; pp[8 * 64] = cv[c].y; pp[9 * 64] = cv[c].z;
            pp[10 * 64] = cw[c].x; pp[11 * 64] = cw[c].y; pp[12 * 64] = cw[c].z;
        }
    };
    const bool carry = P.warm != nullptr;
    auto wld = [&](int idx) -> float { return (carry && valid) ? P.warm[(size_t)idx * N + e] : 0.f; };
    float Wfloor[NC][4][4];
    bool cc_prev[NCC];
#pragma unroll
    for (int s = 0; s < NCC; s++) cc_prev[s] = false;
#pragma unroll
    for (int c = 0; c < NC; c++)
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int k = 0; k < 4; k++) Wfloor[c][s][k] = wld(WARM_FLOOR + 16 * c + 4 * s + k);
    constexpr int NRW = ROLL ? 6 : 4;
    float Wf01[2][NRW];   // carried forces of the finger<->cube slots 0, 1 (one cube: this wave owns them; two cubes: wave A does)
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
        for (int k = 0; k < NRW; k++) Wf01[s][k] = NC == 1 ? wld(WARM_ARM + 6 * s + k) : 0.f;
    float *ccl = lds + LL::CC0 + lane;   // Stack: record field k of slot s at ccl[(s * CC_REC + k) * 64]
    const size_t CS = 64;
    if constexpr (NC == 2) {
#pragma unroll
        for (int s = 0; s < NCC; s++) {
            cc_prev[s] = wld(s < 4 ? WARM_CCPREV + s : WARM_CCPREV2 + (s - 4)) != 0.f;
#pragma unroll
            for (int r = 0; r < 4; r++) ccl[(size_t)(s * CC_REC + 3 + r) * 64] = wld((s < 4 ? WARM_CC + 4 * s : WARM_CC2 + 4 * (s - 4)) + r);
        }
    }
    publish_pose();
    wg_barrier();   // E0
    float ctrl[6];
    {   // wave A's actuator targets and post-IK arm configuration
        const float *ph = lds + LL::G0 + lane;
#pragma unroll
        for (int j = 0; j < 6; j++) { ctrl[j] = ph[j * 64]; q[j] = ph[(6 + j) * 64]; }
    }

    const bool prof = P.diag == 3;   // profiling aid: cycles of this wave in total / waiting at barriers -> ctrl_out[0], [1]
    // issue priority where two waves share a SIMD (the partner there belongs to another workgroup): this wave is on the critical path while it
    // computes tau (wave A waits for it at barrier X) and the joint acceleration (barrier E); elsewhere it has slack and yields
    long long pf_t0 = prof ? clock64() : 0, pf_wait = 0, pf_mark = 0;
    Diag DGtot = {0u, 0u, 0u, 0u};
    const float minv = P.cube_minv, iinv = P.cube_iinv;
    bool hot = false;   // this workgroup has had a coupled substep in this step
    for (int sub = 0; sub < P.n_substeps; sub++) {
        Diag DG = {0u, 0u, 0u, 0u};
        __builtin_amdgcn_s_setprio(3);
        // ---- forward kinematics (this wave's own: cheaper than moving 72 floats through LDS) and, unless the arm wave keeps it (two cubes:
        //      this wave is the busier one there), the smooth joint forces tau -> wave A ----
        ArmFrames F;
        arm_frames(q, F);
        f3 z[6];
#pragma unroll
        for (int j = 0; j < 6; j++) z[j] = joint_axis(F, j);
        const f3 sph[2] = {local_point(F, 4, SPH0x, SPH0y, SPH0z), local_point(F, 5, SPH1x, SPH1y, SPH1z)};   // finger spheres (slots 0, 1 below)
        if constexpr (!RNE_ON_ARM) {
            float tau[6];
            rne_tau(F, z, q, qd, ctrl, tau);
#pragma unroll
            for (int i = 0; i < 6; i++) lds[LL::G0 + lane + i * 64] = tau[i];   // -> wave A (g rows of slot 0: idle between wave A's last sweep and its next row set-up)
        }
        if (prof) pf_mark = clock64();
        wg_barrier();   // X: tau is in LDS for wave A; wave A's Cholesky factor of the joint-space inertia is in LDS
        if (prof) pf_wait += clock64() - pf_mark;
        if (!hot) __builtin_amdgcn_s_setprio(0);
        Chol6 CL;
        {
            const float *pl = lds + LL::LFAC0 + lane;
            int k = 0;
#pragma unroll
            for (int i = 1; i < 6; i++)
#pragma unroll
                for (int j = 0; j < i; j++) CL.L[i][j] = pl[(k++) * 64];
#pragma unroll
            for (int i = 0; i < 6; i++) CL.id[i] = pl[(k++) * 64];
        }
        wg_barrier();   // X2: the factor has been read -- wave A may overwrite its LDS place with contact rows
        CubeRot CR[NC];
        f3 ca[NC], cal[NC], cww[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) {
            CR[c] = quat_to_cols(cq[c]);
            ca[c] = mk(0.f, 0.f, -GRAV);
            cal[c] = mk(0.f, 0.f, 0.f);
            cww[c] = axpy(cw[c].x, CR[c].X, axpy(cw[c].y, CR[c].Y, cw[c].z * CR[c].Z));
        }
        // ---- finger spheres vs cube(s): slots 0, 1 (one contact per sphere, the deepest cube; rows g = L^-1 J^T -> LDS).  Their rows touch the
        //      arm (through y, which visits this wave once per sweep when a sphere touches) and the cube they sit on. ----
        ArmSlot2<NRW> AS01[2];
        bool s01_any[2];
        int s01_cube[2] = {0, 0};
        float dy01[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // sum of g_r f_r of the warm-start forces: wave A adds it to y
        s01_any[0] = false; s01_any[1] = false;
        if constexpr (NC == 1) {   // (two cubes: wave A owns these slots -- this wave then carries two cubes' floor rows and the cube<->cube rows)
        const float srad[2] = {SPH0r, SPH1r};
#pragma unroll
        for (int sp = 0; sp < 2; sp++) {
            ArmSlot2<NRW> &T = AS01[sp];
            f3 pos = mk(0.f, 0.f, 0.f), n = mk(0.f, 0.f, 1.f);
            int cidx = 0, sel = 0;
            float bestd = 1e30f;
            bool near_any = false;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const f3 dd = sph[sp] - cp[c];
                near_any = near_any || dot(dd, dd) < (srad[sp] + 1.7321f * CH) * (srad[sp] + 1.7321f * CH);
            }
            if (__any(near_any))   // broad phase (wave-uniform): a sphere farther than r + h sqrt(3) from every cube centre cannot touch
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const SBHit hit = sphere_box(sph[sp], srad[sp], cp[c], CR[c]);
                if (hit.dist < bestd) { bestd = hit.dist; cidx = c; n = hit.n; pos = hit.pos; sel = 8 * c + hit.code; }  // deepest cube wins (tie: cube 0)
            }
            const float dist = bestd;
            s01_cube[sp] = cidx;
            T.act = dist < 0.f;
            if (P.diag) {
                sel += (n.y < 0.5f && n.y > -0.5f) ? 0 : 16;   // branch of make_frame
                diag_choice(DG, T.act, 12 + sp, sel);
            }
            s01_any[sp] = __any(T.act) != 0;
#pragma unroll
            for (int k = 0; k < NRW; k++) { T.aref[k] = 0.f; T.inv[k] = 0.f; }
            T.Rn = 1.f; T.n = n; T.t1 = mk(0.f, 1.f, 0.f); T.t2 = mk(-1.f, 0.f, 0.f); T.rc = mk(0.f, 0.f, 0.f);
            if (!s01_any[sp]) {
#pragma unroll
                for (int k = 0; k < NRW; k++) Wf01[sp][k] = 0.f;
            } else {
                make_frame(n, T.t1, T.t2);
                f3 cube_p, cube_v, cube_w;
                if (NC == 2 && cidx == 1) { cube_p = cp[NC - 1]; cube_v = cv[NC - 1]; cube_w = cww[NC - 1]; }
                else { cube_p = cp[0]; cube_v = cv[0]; cube_w = cww[0]; }
                T.rc = pos - cube_p;
                const float imp = impedance(dist, D0_FC, DW_FC, 1.0f / W_FC);
                const float Rn = fmaxf((1.f - imp) * rcp(imp) * ((sp == 0 ? INVW_TRAN_L5 : INVW_TRAN_L6) + minv), 1e-15f);
                const float Rf = Rn * P.inv_impratio;
                const float Rt = Rf * P.rt_fc;
                T.Rn = Rn;
                const int nj = sp == 0 ? 5 : 6;   // joints that move the sphere (link_5 / link_6)
                float Ln01 = 1.f, Lt01 = 0.f;
                f3 jc[6];
#pragma unroll
                for (int j = 0; j < 6; j++) jc[j] = j < nj ? cross(z[j], pos - F.p[j]) : mk(0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < NRW; r++) {
                    f3 d = r == 0 ? T.n : (r == 1 ? T.t1 : (r == 2 ? T.t2 : T.n));   // row 3: rotation about n (torsion)
                    if constexpr (ROLL) { if (r >= 4) d = r == 4 ? T.t1 : T.t2; }     // rows 4, 5: rotation about t1, t2 (rolling)
                    float g[6];
                    float vel = 0.f;
#pragma unroll
                    for (int j = 0; j < 6; j++) {
                        g[j] = r < 3 ? dot(jc[j], d) : (j < nj ? dot(z[j], d) : 0.f);
                        vel = fmaf(g[j], qd[j], vel);
                    }
                    float velc, diagc;
                    if (r < 3) {
                        const f3 rxd = cross(T.rc, d);
                        velc = dot(d, cube_v) + dot(rxd, cube_w);
                        diagc = minv + iinv * dot(rxd, rxd);
                    } else {
                        velc = dot(d, cube_w);
                        diagc = iinv;
                    }
                    vel -= velc;
                    fsub(CL, g);
                    float gg = 0.f;
#pragma unroll
                    for (int j = 0; j < 6; j++) gg = fmaf(g[j], g[j], gg);
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const float2v gp = {g[2 * k], g[2 * k + 1]};
                        *reinterpret_cast<float2v *>(&lds[LL::G0 + (as2_row0<ROLL>(sp) + r) * LDS_ROW + k * 128 + lane * 2]) = gp;
                    }
                    float Rr = r == 0 ? Rn : (r == 3 ? Rt : Rf);
                    if (ROLL && r > 3) Rr = Rf * P.rr_fc;
                    T.aref[r] = -B_FC * vel - (r == 0 ? K_FC * imp * dist : 0.f);
                    {   // metric of the block step (soc_step)
                        const float arr = gg + diagc + Rr;
                        const float m2r = r == 0 ? 1.f : (r < 3 ? P.mu_fc2 : (r == 3 ? P.mu_fct2 : P.mu_fcr2));
                        const float KF = 2.f;
                        if (r == 0) Ln01 = KF * arr;
                        else Lt01 = fmaf(KF * m2r, arr, Lt01);
                    }
                    const float fw = T.act ? Wf01[sp][r] : 0.f;       // warm start: previous substep's force of this slot
                    Wf01[sp][r] = fw;
#pragma unroll
                    for (int j = 0; j < 6; j++) dy01[j] = fmaf(g[j], fw, dy01[j]);
                    const f3 dl = r < 3 ? (-minv * fw) * d : mk(0.f, 0.f, 0.f);
                    const f3 da = r < 3 ? (-iinv * fw) * cross(T.rc, d) : (-iinv * fw) * d;
                    if (NC == 2 && cidx == 1) { ca[NC - 1] = ca[NC - 1] + dl; cal[NC - 1] = cal[NC - 1] + da; }
                    else { ca[0] = ca[0] + dl; cal[0] = cal[0] + da; }
                }
                {   // k[] of soc_step (a slot that is off in this lane: zeros -> its updates are exact zeros)
                    const float iLn = T.act ? rcp(Ln01) : 0.f, iLt = T.act ? rcp(Lt01) : 0.f, iLs = iLt;
                    T.inv[0] = iLn; T.inv[1] = P.mu_fc2 * iLt; T.inv[2] = Ln01 * rcp(Ln01 + Lt01); T.inv[3] = P.mu_fct2 * iLs;
                    if constexpr (ROLL) { T.inv[4] = P.mu_fcr2 * iLs; T.inv[5] = 0.f; }
                }
            }
        }
        }
        const bool c01 = NC == 1 && (s01_any[0] || s01_any[1]);
        if (lane == 0) const_cast<int *>(xflag)[0] = c01 ? 1 : 0;
        if (c01) {   // (wave A has read the pose before barrier X2: its first six fields carry dy01 until the sweeps start)
#pragma unroll
            for (int j = 0; j < 6; j++) xpose[j * 64] = dy01[j];
        }

        // (built here, right after the rows of slots 0, 1 -- the last users of the factor L -- and before the floor / cube<->cube / rail rows: the ~50 per-slot constants of
        //  those rows are then not live across this block, which is where this wave's register demand peaks)
        // implicitfast solve at the end of the substep: (M + h (damping + kv) I) qacc = L y, M = L L^T rebuilt from the factor.  What wave A needs for it is
        // built here and handed over; wave A finishes the solve itself at the end of its sweeps: no round trip, and neither factor outlives this block.
        Chol6 CL2;
        float Lf[6][6];
        {
            float Mm[6][6];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) Lf[i][j] = i == j ? rcp(CL.id[i]) : CL.L[i][j];
#pragma unroll
            for (int i = 0; i < 6; i++)
#pragma unroll
                for (int j = 0; j <= i; j++) {
                    float m = 0.f;
#pragma unroll
                    for (int k = 0; k <= j; k++) m = fmaf(Lf[i][k], Lf[j][k], m);
                    Mm[i][j] = m + (i == j ? H * (DAMPING + KV) : 0.f);
                }
            chol6(Mm, CL2);
        }
        // (M + hD) = G G^T.  Wave A needs qacc = (M + hD)^-1 L y = G^-T (V y) with V = G^-1 L (lower triangular): this wave hands over V (21) and G (15 + 6
        // inverse diagonals) -- six forward substitutions on columns that start with zeros, a quarter of the work of Wm = G^-T V -- and wave A finishes with 21 + 21
        // multiply-adds.  Place (42 fields: V row-major, G's strict lower part row-major, G's inverse diagonal): the idle LDS rows of slots 0, 1; with a finger on a
        // cube those rows are in use and the record goes to its own LDS place (build for one wave per SIMD) or to this lane's global scratch record (168 B; the build for
        // two waves per SIMD has no LDS left, and keeping the factors for a solve after the sweeps would cost this wave registers it does not have at the 256-register cap)
        auto hand_over = [&](auto glob_tag) {
            constexpr bool GLOB = decltype(glob_tag)::value;
            float *pw = lds + ((c01 || NC == 2) ? LL::WM0 : LL::G0) + lane;
            float *gw = P.scratch + ((size_t)blockIdx.x * 64 + lane) * 48;
            auto put = [&](int f, float v) { if (GLOB) gw[f] = v; else pw[f * 64] = v; };
#pragma unroll
            for (int j = 0; j < 6; j++) {   // column j of L (zero above the diagonal) -> column j of V
                float col[6];
#pragma unroll
                for (int i = 0; i < 6; i++) col[i] = i < j ? 0.f : Lf[i][j];
                fsub(CL2, col);
#pragma unroll
                for (int i = j; i < 6; i++) put(i * (i + 1) / 2 + j, col[i]);
            }
#pragma unroll
            for (int i = 1; i < 6; i++)
#pragma unroll
                for (int j = 0; j < i; j++) put(21 + i * (i - 1) / 2 + j, CL2.L[i][j]);
#pragma unroll
            for (int i = 0; i < 6; i++) put(36 + i, CL2.id[i]);
            // (the barrier's own fence does not cover these global stores; WORKGROUP scope is what the hand-over needs -- both waves run on one CU and share its vector L1
            //  (no threadgroup-split mode), so nothing has to be written back or invalidated.  Until late in round 4 this was an agent-scope fence pair: `buffer_wbl2` here and
            //  `buffer_inv` on wave A's side, an L2 write-back and an invalidate per coupled substep and workgroup -- Push / Lift / PickPlace at 65 536 envs 0.337-0.340 -> 0.307-0.312 ms)
            if (GLOB) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        };
        if (GW && (c01 || NC == 2)) hand_over(std::true_type{}); else hand_over(std::false_type{});
        // ---- collision: floor <-> cube (MuJoCo plane-box: penetrating vertices in index order, at most 4) ----
        FloorSlot FS[NC][4];
#pragma unroll
        for (int c = 0; c < NC; c++) {
#pragma unroll
            for (int s = 0; s < 4; s++) { FS[c][s].act = false; FS[c][s].r = mk(0.f, 0.f, 0.f); FS[c][s].Rn = 1.f;
#pragma unroll
                for (int k = 0; k < 4; k++) { FS[c][s].aref[k] = 0.f; FS[c][s].inv[k] = 0.f; } }
            float sdist[4] = {0.f, 0.f, 0.f, 0.f};
            const f3 hx = CH * CR[c].X, hy = CH * CR[c].Y, hz = CH * CR[c].Z;
            f3 rv[8];
            float vd[8];
            bool lower_same = true, upper_none = true;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                f3 a = (i & 1) ? hx : neg(hx), b = (i & 2) ? hy : neg(hy), d = (i & 4) ? hz : neg(hz);
                rv[i] = a + b + d;
                vd[i] = cp[c].z + rv[i].z;
                if (i >= 4) upper_none = upper_none && !(vd[i] < 0.f);
                else if (i > 0) lower_same = lower_same && ((vd[i] < 0.f) == (vd[0] < 0.f));
            }
            if (__all(lower_same && upper_none)) {
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
                        FS[c][s].r.z = take ? rv[i].z - 0.5f * dist : FS[c][s].r.z;
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
                float Rn = fmaxf((1.f - imp) * rcp(imp) * minv, 1e-15f);
                float Rf = Rn * P.inv_impratio;
                float Rt = Rf * P.rt_cube;
                T.Rn = Rn;
                f3 vp = cv[c] + cross(cww[c], T.r);
                T.aref[0] = -B_DEF * vp.z - K_DEF * imp * sdist[s];
                T.aref[1] = -B_DEF * vp.y;
                T.aref[2] = B_DEF * vp.x;
                T.aref[3] = -B_DEF * cww[c].z;
                {   // k[] of soc_step: Ln = 2 (A + R)_nn, Lt = 2 (mu^2 ((A + R)_11 + (A + R)_22) + mu_tors^2 (A + R)_33)
                    const float KF = 2.f;   // (two groups, see soc_step)
                    const float Ln = KF * (minv + iinv * (T.r.x * T.r.x + T.r.y * T.r.y) + Rn);
                    const float a12 = 2.f * minv + iinv * (2.f * T.r.z * T.r.z + T.r.x * T.r.x + T.r.y * T.r.y) + 2.f * Rf;
                    const float Ls = KF * P.mu_ct2 * (iinv + Rt);
                    const float Lt = fmaf(KF * P.mu_c2, a12, Ls);
                    const float iLt = T.act ? rcp(Lt) : 0.f, iLs = iLt;
                    T.inv[0] = T.act ? rcp(Ln) : 0.f; T.inv[1] = P.mu_c2 * iLt; T.inv[2] = Ln * rcp(Ln + Lt); T.inv[3] = P.mu_ct2 * iLs;
                }
                float (&Tf)[4] = Wfloor[c][s];   // the carried forces ARE this slot's forces from here on (zero if the slot is off)
#pragma unroll
                for (int k = 0; k < 4; k++) { Tf[k] = T.act ? Tf[k] : 0.f; }
                const f3 r = T.r;
                ca[c].z = fmaf(minv, Tf[0], ca[c].z);
                ca[c].y = fmaf(minv, Tf[1], ca[c].y);
                ca[c].x = fmaf(-minv, Tf[2], ca[c].x);
                cal[c].x = fmaf(iinv, r.y * Tf[0] - r.z * Tf[1], cal[c].x);
                cal[c].y = fmaf(iinv, -r.x * Tf[0] - r.z * Tf[2], cal[c].y);
                cal[c].z = fmaf(iinv, r.x * Tf[1] + r.y * Tf[2] + Tf[3], cal[c].z);
            }
        }

        // ---- collision: cube <-> cube (Stack); records in LDS (this wave only) ----
        bool cc_act[NCC];
#pragma unroll
        for (int s = 0; s < NCC; s++) cc_act[s] = false;
        bool cc_any = false;
        f3 ccn = mk(0.f, 0.f, 1.f), cct1 = mk(0.f, 1.f, 0.f), cct2 = mk(-1.f, 0.f, 0.f);
        if constexpr (NC == 2) {
            const f3 dc = cp[1] - cp[0];
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
            if (__any(touching)) {
                const bool Ais0 = bax < 3;
                const int k = Ais0 ? bax : bax - 3;
                auto sel3 = [](int i, f3 a, f3 b, f3 c) { return i == 0 ? a : (i == 1 ? b : c); };
                const f3 AX = Ais0 ? CR[0].X : CR[1].X, AY = Ais0 ? CR[0].Y : CR[1].Y, AZ = Ais0 ? CR[0].Z : CR[1].Z;
                const f3 BX = Ais0 ? CR[1].X : CR[0].X, BY = Ais0 ? CR[1].Y : CR[0].Y, BZ = Ais0 ? CR[1].Z : CR[0].Z;
                const f3 cA = Ais0 ? cp[0] : cp[1], cB = Ais0 ? cp[1] : cp[0];
                ccn = bsgn * sel3(k, AX, AY, AZ);
                const f3 m = Ais0 ? ccn : neg(ccn);
                const f3 u = sel3(k, AY, AZ, AX), v = sel3(k, AZ, AX, AY);
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
                auto consider = [&](bool ok, int cand, f3 Pp, float dist, float cu, float cv_) {
                    const float key[8] = {cu + cv_, -cu + cv_, -cu - cv_, cu - cv_, cu, cv_, -cu, -cv_};   // diagonals of the reference face, then its axes
#pragma unroll
                    for (int s = 0; s < NCC; s++) {
                        bool t = ok && (sidx[s] < 0 || key[s] > skey[s]);
                        skey[s] = t ? key[s] : skey[s];
                        sidx[s] = t ? cand : sidx[s];
                        cdist[s] = t ? dist : cdist[s];
                        cpos[s].x = t ? Pp.x : cpos[s].x; cpos[s].y = t ? Pp.y : cpos[s].y; cpos[s].z = t ? Pp.z : cpos[s].z;
                    }
                };
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    bool ok = touching && !(fabsf(Vu[i]) > CH + tol || fabsf(Vv[i]) > CH + tol) && Vd[i] < 0.f;
                    consider(ok, i, axpy(-0.5f * Vd[i], m, V[i]), Vd[i], Vu[i], Vv[i]);
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    f3 a = axpy(CH, m, axpy(CH * SP[j], u, axpy(CH * SQ[j], v, cA)));
                    f3 d = a - fB;
                    float t = -dot(d, nb) * rcp(mnb);
                    bool ok = touching && mnb < -0.5f && !(fabsf(dot(d, pa)) > CH + tol || fabsf(dot(d, qa)) > CH + tol) && t < 0.f;
                    consider(ok, 4 + j, axpy(0.5f * t, m, a), t, SP[j] * CH, SQ[j] * CH);
                }
#pragma unroll
                for (int ed = 0; ed < 4; ed++) {
                    const int e2 = (ed + 1) & 3;
#pragma unroll
                    for (int l = 0; l < 4; l++) {
                        const bool on_u = l < 2;
                        const float sg = (l & 1) ? -1.f : 1.f;
                        const float cP = on_u ? Vu[ed] : Vv[ed], cQ = on_u ? Vu[e2] : Vv[e2];
                        const float oP = on_u ? Vv[ed] : Vu[ed], oQ = on_u ? Vv[e2] : Vu[e2];
                        const float fP = cP - sg * CH, fQ = cQ - sg * CH;
                        const bool cross_ = (fP < 0.f && fQ > 0.f) || (fP > 0.f && fQ < 0.f);
                        const float t = fP * rcp(cross_ ? fP - fQ : 1.f);
                        const float ot = fmaf(t, oQ - oP, oP);
                        const float d = fmaf(t, Vd[e2] - Vd[ed], Vd[ed]);
                        bool ok = touching && cross_ && !(fabsf(ot) > CH + tol) && d < 0.f;
                        f3 X = axpy(t, V[e2] - V[ed], V[ed]);
                        consider(ok, 8 + 4 * ed + l, axpy(-0.5f * d, m, X), d, on_u ? sg * CH : ot, on_u ? ot : sg * CH);
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
            if (cc_any) {
                make_frame(ccn, cct1, cct2);
#pragma unroll
                for (int s = 0; s < NCC; s++) {
                    const f3 r0 = cpos[s] - cp[0], r1 = cpos[s] - cp[1];
                    float imp = impedance(cdist[s], D0_DEF, DW_DEF, 1.0f / W_DEF);
                    float Rn = fmaxf((1.f - imp) * rcp(imp) * (2.f * minv), 1e-15f);
                    float Rf = Rn * P.inv_impratio;
                    float Rt = Rf * P.rt_cube;
                    f3 vrel = (cv[1] + cross(cww[1], r1)) - (cv[0] + cross(cww[0], r0));
                    f3 wrel = cww[1] - cww[0];
                    ccl[(size_t)(s * CC_REC + 0) * CS] = cpos[s].x; ccl[(size_t)(s * CC_REC + 1) * CS] = cpos[s].y; ccl[(size_t)(s * CC_REC + 2) * CS] = cpos[s].z;
                    float ccLn = 1.f, ccLt = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const f3 d = r == 0 ? ccn : (r == 1 ? cct1 : (r == 2 ? cct2 : ccn));
                        float vel = r < 3 ? dot(d, vrel) : dot(d, wrel);
                        float aref = -B_DEF * vel - (r == 0 ? K_DEF * imp * cdist[s] : 0.f);
                        float diag;
                        if (r < 3) { f3 a0 = cross(r0, d), a1 = cross(r1, d); diag = 2.f * minv + iinv * (dot(a0, a0) + dot(a1, a1)); }
                        else diag = 2.f * iinv;
                        float Rr = r == 0 ? Rn : (r == 3 ? Rt : Rf);
                        {
                            float fw = (cc_act[s] && cc_prev[s]) ? ccl[(size_t)(s * CC_REC + 3 + r) * CS] : 0.f;
                            ccl[(size_t)(s * CC_REC + 3 + r) * CS] = fw;
                            if (r < 3) {
                                f3 a0 = cross(r0, d), a1 = cross(r1, d);
                                ca[1] = axpy(minv * fw, d, ca[1]); ca[0] = axpy(-minv * fw, d, ca[0]);
                                cal[1] = axpy(iinv * fw, a1, cal[1]); cal[0] = axpy(-iinv * fw, a0, cal[0]);
                            } else { cal[1] = axpy(iinv * fw, d, cal[1]); cal[0] = axpy(-iinv * fw, d, cal[0]); }
                        }
                        ccl[(size_t)(s * CC_REC + 7 + r) * CS] = aref;
                        if (r == 0) ccLn = 2.f * (diag + Rr); else ccLt = fmaf(2.f * (r == 3 ? P.mu_ct2 : P.mu_c2), diag + Rr, ccLt);
                    }
                    {   // k[] of soc_step in the record's four "inverse diagonal" fields
                        const float iLt = cc_act[s] ? rcp(ccLt) : 0.f;
                        ccl[(size_t)(s * CC_REC + 11) * CS] = cc_act[s] ? rcp(ccLn) : 0.f;
                        ccl[(size_t)(s * CC_REC + 12) * CS] = P.mu_c2 * iLt;
                        ccl[(size_t)(s * CC_REC + 13) * CS] = ccLn * rcp(ccLn + ccLt);
                        ccl[(size_t)(s * CC_REC + 14) * CS] = P.mu_ct2 * iLt;
                    }
                    ccl[(size_t)(s * CC_REC + 15) * CS] = Rn;
                }
            }
        }

        if (prof) pf_mark = clock64();
        wg_barrier();   // B1: wave A has set up its rows and decided whether this substep is coupled
        if (prof) pf_wait += clock64() - pf_mark;
        const bool cube4 = __builtin_amdgcn_readfirstlane(xflag[1]) != 0;   // wave A: a gripper-body proxy touches a cube in some lane
        const bool coupled = c01 || cube4;
        // A workgroup with coupled substeps is the one a launch waits for (its two chains run in series): from its first coupled substep on both its
        // waves take the top issue priority for the rest of the step, so that where two waves share a SIMD the partner fills the gaps instead of halving them.
        if (coupled && !hot) { hot = true; __builtin_amdgcn_s_setprio(3); }
        if (cube4) {   // warm-start forces of the proxy slot act on the cube too
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const float *pa = xpose + (size_t)(6 + c * 6) * 64;
                ca[c] = ca[c] + mk(pa[0], pa[64], pa[128]); cal[c] = cal[c] + mk(pa[192], pa[256], pa[320]);
            }
        }

        // ---- one Gauss-Seidel pass over the rows that touch only the cubes ----
        auto cube_rows = [&]() {
#pragma unroll
            for (int c = 0; c < NC; c++) {
#pragma unroll
                for (int s = 0; s < 4; s++) {
                    FloorSlot &T = FS[c][s];
                    float (&Tf)[4] = Wfloor[c][s];
                    const f3 r = T.r;
                    const float Rf = T.Rn * P.inv_impratio;
                    const float Rt = Rf * P.rt_cube;
                    const float u0 = ca[c].z + r.y * cal[c].x - r.x * cal[c].y - T.aref[0] + T.Rn * Tf[0];
                    const float u1 = ca[c].y - r.z * cal[c].x + r.x * cal[c].z - T.aref[1] + Rf * Tf[1];
                    const float u2 = -ca[c].x - r.z * cal[c].y + r.y * cal[c].z - T.aref[2] + Rf * Tf[2];
                    const float u3 = cal[c].z - T.aref[3] + Rt * Tf[3];
                    const float uu[4] = {u0, u1, u2, u3};
                    float nf[4];
                    soc_step<4>(Tf, uu, T.inv, P.inv_mu_c2, P.inv_mu_ct2, 0.f, 4, nf);
                    const float d0 = nf[0] - Tf[0], d1 = nf[1] - Tf[1], d2 = nf[2] - Tf[2], d3 = nf[3] - Tf[3];
                    Tf[0] = nf[0]; Tf[1] = nf[1]; Tf[2] = nf[2]; Tf[3] = nf[3];
                    ca[c].z = fmaf(minv, d0, ca[c].z);
                    ca[c].y = fmaf(minv, d1, ca[c].y);
                    ca[c].x = fmaf(-minv, d2, ca[c].x);
                    cal[c].x = fmaf(iinv, r.y * d0 - r.z * d1, cal[c].x);
                    cal[c].y = fmaf(iinv, -r.x * d0 - r.z * d2, cal[c].y);
                    cal[c].z = fmaf(iinv, fmaf(r.x, d1, fmaf(r.y, d2, d3)), cal[c].z);
                }
            }
            if constexpr (NC == 2) {
                if (cc_any) {
#pragma unroll
                    for (int s = 0; s < NCC; s++) {
                        const f3 pos = mk(ccl[(size_t)(s * CC_REC + 0) * CS], ccl[(size_t)(s * CC_REC + 1) * CS], ccl[(size_t)(s * CC_REC + 2) * CS]);
                        const f3 r0 = pos - cp[0], r1 = pos - cp[1];
                        const float Rn = ccl[(size_t)(s * CC_REC + 15) * CS];
                        const float Rf = Rn * P.inv_impratio;
                        const float Rt = Rf * P.rt_cube;
                        float f[4], aref[4], inv[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            f[r] = ccl[(size_t)(s * CC_REC + 3 + r) * CS];
                            aref[r] = ccl[(size_t)(s * CC_REC + 7 + r) * CS];
                            inv[r] = ccl[(size_t)(s * CC_REC + 11 + r) * CS];
                        }
                        const f3 A = (ca[1] + cross(cal[1], r1)) - (ca[0] + cross(cal[0], r0));
                        const f3 Wr = cal[1] - cal[0];
                        const float u0 = dot(ccn, A) - aref[0] + Rn * f[0];
                        const float u1 = dot(cct1, A) - aref[1] + Rf * f[1];
                        const float u2 = dot(cct2, A) - aref[2] + Rf * f[2];
                        const float u3 = dot(ccn, Wr) - aref[3] + Rt * f[3];
                        const float uu[4] = {u0, u1, u2, u3};
                        float nf[4];
                        soc_step<4>(f, uu, inv, P.inv_mu_c2, P.inv_mu_ct2, 0.f, 4, nf);
                        const float d0 = nf[0] - f[0], e1 = nf[1] - f[1], e2 = nf[2] - f[2], e3 = nf[3] - f[3];
                        ccl[(size_t)(s * CC_REC + 3) * CS] = nf[0]; ccl[(size_t)(s * CC_REC + 4) * CS] = nf[1];
                        ccl[(size_t)(s * CC_REC + 5) * CS] = nf[2]; ccl[(size_t)(s * CC_REC + 6) * CS] = nf[3];
                        const f3 Fd = axpy(d0, ccn, axpy(e1, cct1, e2 * cct2));
                        const f3 T1 = axpy(e3, ccn, cross(r1, Fd)), T0 = axpy(e3, ccn, cross(r0, Fd));
                        ca[1] = axpy(minv, Fd, ca[1]); ca[0] = axpy(-minv, Fd, ca[0]);
                        cal[1] = axpy(iinv, T1, cal[1]); cal[0] = axpy(-iinv, T0, cal[0]);
                    }
                }
            }
        };
        // ---- one Gauss-Seidel pass over the finger<->cube slots 0, 1: arm side through yv (the arm acceleration on its visit from wave A),
        //      cube side tracked as scalars with closed-form couplings (see lcr_kernels.hip), summed force applied to the cube at the end ----
        float yv[6];
        auto finger_rows = [&]() {
#pragma unroll
            for (int s = 0; s < 2; s++) {
                if (!s01_any[s]) continue;
                ArmSlot2<NRW> &T = AS01[s];
                const float Rf = T.Rn * P.inv_impratio;
                const float Rt = Rf * P.rt_fc;
                float2v g[NRW][3];
#pragma unroll
                for (int r = 0; r < NRW; r++)
#pragma unroll
                    for (int k = 0; k < 3; k++)
                        g[r][k] = *reinterpret_cast<const float2v *>(&lds[LL::G0 + (as2_row0<ROLL>(s) + r) * LDS_ROW + k * 128 + lane * 2]);
                float2v yp[3] = {{yv[0], yv[1]}, {yv[2], yv[3]}, {yv[4], yv[5]}};
                float f_in[NRW];
#pragma unroll
                for (int r = 0; r < NRW; r++) f_in[r] = Wf01[s][r];
                const bool second = NC == 2 && s01_cube[s] == 1;
                const f3 a_lin = second ? ca[NC - 1] : ca[0], a_ang = second ? cal[NC - 1] : cal[0];
                float vq[3], wn, w1 = 0.f, w2 = 0.f;
                {
                    const f3 Ac = a_lin + cross(a_ang, T.rc);
                    vq[0] = dot(T.n, Ac); vq[1] = dot(T.t1, Ac); vq[2] = dot(T.t2, Ac);
                    wn = dot(T.n, a_ang);
                    if (NRW == 6) { w1 = dot(T.t1, a_ang); w2 = dot(T.t2, a_ang); }
                }
                float u[NRW], fcur[NRW], nf[NRW];
#pragma unroll
                for (int r = 0; r < NRW; r++) {
                    fcur[r] = Wf01[s][r];
                    const float2v acc = g[r][0] * yp[0] + g[r][1] * yp[1] + g[r][2] * yp[2];
                    const float gy = acc.x + acc.y;
                    float jc_a = r < 3 ? -vq[r] : -wn;
                    float Rr = r == 0 ? T.Rn : (r == 3 ? Rt : Rf);
                    if (ROLL && r > 3) { jc_a = r == 4 ? -w1 : -w2; Rr = Rf * P.rr_fc; }
                    u[r] = gy + jc_a - T.aref[r] + Rr * fcur[r];
                }
                soc_step<NRW>(fcur, u, T.inv, 1.f / (MU_FINGER * MU_FINGER), P.inv_mu_fct2, P.inv_mu_fcr2, NRW, nf);   // (finger<->cube pair: max rule, the finger's 1.5 is the largest cube friction of any task)
#pragma unroll
                for (int r = 0; r < NRW; r++) {
                    const float dlt = nf[r] - fcur[r];
                    Wf01[s][r] = nf[r];
                    const float2v d2 = {dlt, dlt};
#pragma unroll
                    for (int k = 0; k < 3; k++) yp[k] = g[r][k] * d2 + yp[k];
                }
                yv[0] = yp[0].x; yv[1] = yp[0].y; yv[2] = yp[1].x; yv[3] = yp[1].y; yv[4] = yp[2].x; yv[5] = yp[2].y;
                {
                    const float e0 = Wf01[s][0] - f_in[0], e1 = Wf01[s][1] - f_in[1], e2 = Wf01[s][2] - f_in[2], e3 = Wf01[s][3] - f_in[3];
                    const f3 Fd = axpy(e0, T.n, axpy(e1, T.t1, e2 * T.t2));   // force change on the arm; the cube gets -Fd at rc
                    const f3 dl_lin = (-minv) * Fd;
                    f3 Td = axpy(e3, T.n, cross(T.rc, Fd));
                    if constexpr (ROLL) Td = axpy(Wf01[s][4] - f_in[4], T.t1, axpy(Wf01[s][5] - f_in[5], T.t2, Td));
                    const f3 dl_ang = (-iinv) * Td;
                    if (second) { ca[NC - 1] = ca[NC - 1] + dl_lin; cal[NC - 1] = cal[NC - 1] + dl_ang; }
                    else { ca[0] = ca[0] + dl_lin; cal[0] = cal[0] + dl_ang; }
                }
            }
        };
        auto bar = [&]() {
            if (prof) pf_mark = clock64();
            wg_barrier();
            if (prof) pf_wait += clock64() - pf_mark;
        };
        if (!coupled) {
            for (int it = 0; it < P.pgs_iters; it++) cube_rows();
        } else {
            // coupled substep: the two row groups sweep concurrently and merge their changes at the end of every sweep (protocol: see wave A)
            float *xdyb = xpose, *xdca = xpose + (size_t)6 * 64;
            float ystart[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (cube4) {   // this wave's accelerations after its row set-up (incl. slot 4's warm-start share) -> wave A, which needs them for slot 4 in the first sweep
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    float *pa = xacc + (size_t)c * 6 * 64;
                    pa[0] = ca[c].x; pa[64] = ca[c].y; pa[128] = ca[c].z; pa[192] = cal[c].x; pa[256] = cal[c].y; pa[320] = cal[c].z;
                }
            }
            bar();   // P0
            if (c01) {
#pragma unroll
                for (int j = 0; j < 6; j++) ystart[j] = xdyb[j * 64];
            }
            for (int it = 0; it < P.pgs_iters; it++) {
                cube_rows();
                if (c01) {
#pragma unroll
                    for (int j = 0; j < 6; j++) yv[j] = ystart[j];
                    finger_rows();
#pragma unroll
                    for (int j = 0; j < 6; j++) xdyb[j * 64] = yv[j] - ystart[j];
                }
                bar();   // W
                if (c01) {   // merged y = wave A's y after its rows + this wave's change: the start of the next sweep (the same expression as on wave A)
#pragma unroll
                    for (int j = 0; j < 6; j++) ystart[j] = xacc[j * 64] + xdyb[j * 64];
                }
                if (cube4) {   // merged cube accelerations = this wave's + slot 4's change; wave A reads them back from ACC after R (YA has been read above)
#pragma unroll
                    for (int c = 0; c < NC; c++) {
                        const float *pd = xdca + (size_t)c * 6 * 64;
                        ca[c] = ca[c] + mk(pd[0], pd[64], pd[128]); cal[c] = cal[c] + mk(pd[192], pd[256], pd[320]);
                        float *pa = xacc + (size_t)c * 6 * 64;
                        pa[0] = ca[c].x; pa[64] = ca[c].y; pa[128] = ca[c].z; pa[192] = cal[c].x; pa[256] = cal[c].y; pa[320] = cal[c].z;
                    }
                }
                bar();   // R
            }
        }

        // ---- keep the forces for the next substep's warm start (inactive slots hold zero) ----------------
#pragma unroll
        for (int s = 0; s < NCC; s++) cc_prev[s] = cc_act[s];
        if (P.diag) {
            unsigned m = 0u;
#pragma unroll
            for (int c = 0; c < NC; c++)
#pragma unroll
                for (int s = 0; s < 4; s++) m |= FS[c][s].act ? (1u << (4 * c + s)) : 0u;
#pragma unroll
            for (int s = 0; s < 4; s++) {
                if (NC == 2) m |= cc_act[s] ? (1u << (8 + s)) : 0u;
            }
            if constexpr (CC8) {
#pragma unroll
                for (int s = 4; s < 8; s++) m |= cc_act[s] ? (1u << (24 + (s - 4))) : 0u;
            }
            if constexpr (NC == 1) { m |= AS01[0].act ? (1u << 12) : 0u; m |= AS01[1].act ? (1u << 13) : 0u; }   // (two cubes: wave A's bits)
            DGtot.mask |= m;
            DGtot.count += (unsigned)__popc(m);
            DGtot.choice += DG.choice * (unsigned)(2 * sub + 1);
        }

        // ---- integrate the cubes ----
#pragma unroll
        for (int c = 0; c < NC; c++) {
            cv[c] = axpy(H, ca[c], cv[c]);
            f3 ab = mk(dot(CR[c].X, cal[c]), dot(CR[c].Y, cal[c]), dot(CR[c].Z, cal[c]));
            cw[c] = axpy(H, ab, cw[c]);
            cp[c] = axpy(H, cv[c], cp[c]);
            f3 w = cw[c];
            float wn2 = dot(w, w);
            if (wn2 > 0.f) {
                float iw = rsq(wn2), wn = wn2 * iw;
                float sh, chf;
                sincos_small(0.5f * H * wn, &sh, &chf);
                float s = sh * iw;
                float dq0 = chf, dq1 = w.x * s, dq2 = w.y * s, dq3 = w.z * s;
                float q0 = cq[c][0], q1 = cq[c][1], q2 = cq[c][2], q3 = cq[c][3];
                float r0 = q0 * dq0 - q1 * dq1 - q2 * dq2 - q3 * dq3;
                float r1 = q0 * dq1 + q1 * dq0 + q2 * dq3 - q3 * dq2;
                float r2 = q0 * dq2 - q1 * dq3 + q2 * dq0 + q3 * dq1;
                float r3 = q0 * dq3 + q1 * dq2 - q2 * dq1 + q3 * dq0;
                float in = rsq(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
                cq[c][0] = r0 * in; cq[c][1] = r1 * in; cq[c][2] = r2 * in; cq[c][3] = r3 * in;
            }
        }
        publish_pose();   // (this wave finishes its sweeps first: the integration is off the critical path)
        // (the implicitfast solve is wave A's: it has y; this wave reads qacc after barrier E)
        if (prof) pf_mark = clock64();
        wg_barrier();   // E
        if (prof) pf_wait += clock64() - pf_mark;
#pragma unroll
        for (int j = 0; j < 6; j++) {   // this wave's copy of the arm state follows with the same qacc
            const float qacc = xacc[j * 64];
            qd[j] = fmaf(H, qacc, qd[j]);
            q[j] = fmaf(H, qd[j], q[j]);
        }
    }
    if (prof && valid) { P.ctrl_out[e] = (float)(clock64() - pf_t0); P.ctrl_out[(size_t)N + e] = (float)pf_wait; }
    // diagnostics words of this wave -> wave A (the ACC area is free now)
    if (P.diag && !prof) {
        unsigned *xb = reinterpret_cast<unsigned *>(xacc);
        xb[0] = DGtot.mask; xb[64] = DGtot.count; xb[128] = DGtot.choice;
    }
    wg_barrier();   // T1
    wg_barrier();   // T2: wave A has decided which envs are reset
    const bool do_reset = reinterpret_cast<const int *>(lds + LL::FLAG0)[lane] != 0;
    if (carry && valid) {
        auto wst = [&](int idx, float v) { P.warm[(size_t)idx * N + e] = do_reset ? 0.f : v; };
#pragma unroll
        for (int c = 0; c < NC; c++)
#pragma unroll
            for (int s = 0; s < 4; s++)
#pragma unroll
                for (int k = 0; k < 4; k++) wst(WARM_FLOOR + 16 * c + 4 * s + k, Wfloor[c][s][k]);
        if constexpr (NC == 1) {
#pragma unroll
        for (int s = 0; s < 2; s++)
#pragma unroll
            for (int k = 0; k < NRW; k++) wst(WARM_ARM + 6 * s + k, Wf01[s][k]);
        }
        if constexpr (NC == 2) {
#pragma unroll
            for (int s = 0; s < NCC; s++) {
                wst(s < 4 ? WARM_CCPREV + s : WARM_CCPREV2 + (s - 4), cc_prev[s] ? 1.f : 0.f);
#pragma unroll
                for (int r = 0; r < 4; r++) wst((s < 4 ? WARM_CC + 4 * s : WARM_CC2 + 4 * (s - 4)) + r, cc_prev[s] ? ccl[(size_t)(s * CC_REC + 3 + r) * 64] : 0.f);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// the kernel: workgroup = 2 waves x 64 lanes over 64 envs.  OCC = waves per SIMD the register budget must allow
// (1: up to 512 registers per lane -- shards that put at most one wave on a SIMD; 2: <= 256)
// ------------------------------------------------------------------------------------------------
template <int NC, bool EE, bool ROLL, int OCC, bool CC8>
__global__ __launch_bounds__(128, OCC) void lcr_step2_kernel(LcrDev P, const float *__restrict__ action) {
    constexpr bool GW = OCC == 2 || CC8;   // where V and G travel in substeps with a finger on a cube: global scratch record (true) or their own LDS place
    __shared__ float lds[Lds2<NC, ROLL, CC8, !GW>::TOTAL];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int e_raw = blockIdx.x * 64 + lane;
    const bool valid = e_raw < P.n;
    const int e = valid ? e_raw : P.n - 1;   // tail lanes shadow the last env, their stores are masked
#if defined(LCR2_ONLY_ARM)        // (register-budget study only: one role compiled alone; such a kernel must not be launched)
    arm_program<NC, EE, ROLL, CC8, GW>(P, action, lds, lane, e, valid);
#elif defined(LCR2_ONLY_CUBE)
    cube_program<NC, EE, ROLL, CC8, GW>(P, lds, lane, e, valid);
#else
    if (wave == 0) arm_program<NC, EE, ROLL, CC8, GW>(P, action, lds, lane, e, valid);
    else cube_program<NC, EE, ROLL, CC8, GW>(P, lds, lane, e, valid);
#endif
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// launchers.  build.py compiles this file once per LCR_PART (10 / 14: one cube, built for one / two waves per SIMD; 12: StackTwoCubes, 13: StackTwoCubes with the eight-point manifold)
// ------------------------------------------------------------------------------------------------
static int check_launch2() {
    hipError_t err = hipGetLastError();
    return err == hipSuccess ? 0 : (int)err;
}

template <int NC, bool CC8, int OCC>
static int launch2_build(const LcrDev &P, const float *action_dev, int ee_mode, hipStream_t st) {
    const int blocks = (P.n + 63) / 64;
#define LCR2_GO(EE, ROLL) hipLaunchKernelGGL((lcr_step2_kernel<NC, EE, ROLL, OCC, CC8>), dim3(blocks), dim3(128), 0, st, P, action_dev)
    if (ee_mode) { if (P.roll) LCR2_GO(true, true); else LCR2_GO(true, false); }
    else { if (P.roll) LCR2_GO(false, true); else LCR2_GO(false, false); }
#undef LCR2_GO
    return check_launch2();
}
template <int NC, bool CC8>
static int launch2_family(const LcrDev &P, const float *action_dev, int ee_mode, int occ, hipStream_t st) {
    return occ >= 2 ? launch2_build<NC, CC8, 2>(P, action_dev, ee_mode, st) : launch2_build<NC, CC8, 1>(P, action_dev, ee_mode, st);
}

// The one-cube kernels' two builds are separate units (10: one wave per SIMD, 14: two) because they want different instruction-scheduling flags
// (gym_lowcostrobot_amd/build.py, measured: DESIGN.md section 5) -- same source, same bits.
int lcr_launch_step2_one_cube_occ2(const LcrDev &P, const float *action_dev, int ee_mode, void *stream);
#if LCR_HAS_PART(10)
int lcr_launch_step2_one_cube(const LcrDev &P, const float *action_dev, int ee_mode, int occ, void *stream) {
    if (occ >= 2) return lcr_launch_step2_one_cube_occ2(P, action_dev, ee_mode, stream);
    return launch2_build<1, false, 1>(P, action_dev, ee_mode, (hipStream_t)stream);
}
#endif
#if LCR_HAS_PART(14)
int lcr_launch_step2_one_cube_occ2(const LcrDev &P, const float *action_dev, int ee_mode, void *stream) {
    return launch2_build<1, false, 2>(P, action_dev, ee_mode, (hipStream_t)stream);
}
#endif
#if LCR_HAS_PART(12)
int lcr_launch_step2_stack(const LcrDev &P, const float *action_dev, int ee_mode, int occ, void *stream) {
    return launch2_family<2, false>(P, action_dev, ee_mode, occ, (hipStream_t)stream);
}
#endif
#if LCR_HAS_PART(13)
int lcr_launch_step2_stack_cc8(const LcrDev &P, const float *action_dev, int ee_mode, int occ, void *stream) {   // eight-point cube<->cube manifold
    return launch2_family<2, true>(P, action_dev, ee_mode, 1, (hipStream_t)stream);   // (74-80 KiB of LDS per workgroup: one wave per SIMD at most)
}
#endif
