// lcr_step_common.h -- pieces shared by the two step-kernel families: the one-wave-per-64-envs kernels (lcr_kernels.hip) and the
// two-cooperating-waves kernels (lcr_kernels2.hip): model constants, soft-contact helpers, 6x6 Cholesky, numpy-compatible PCG64,
// the per-env state record, contact-slot records, the layout of the carried-force block, sphere-box narrow phase, diagnostics,
// reset and observation write-out.  Everything is in an anonymous namespace (each translation unit gets its own copy).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lcr_arm.h"
#include "lcr_device.h"
#include "lcr_model_gen.h"

using namespace lcrdev;

namespace {

// ------------------------------------------------------------------------------------------------
// constants (follower.xml:3,7,8 ; scene xmls)
// ------------------------------------------------------------------------------------------------
constexpr float H = 0.002f;
constexpr float ARMATURE = 0.1f;
constexpr float DAMPING = 1.0f;
constexpr float KP = 1000.0f;
constexpr float KV = 10.0f;
constexpr float FRC = 10.0f;
constexpr float GRAV = 9.81f;
constexpr float CH = lcrm::CUBE_HALF;  // cube half size
constexpr float JLO[6] = {lcrm::JNT_LO[0], lcrm::JNT_LO[1], lcrm::JNT_LO[2], lcrm::JNT_LO[3], lcrm::JNT_LO[4], lcrm::JNT_LO[5]};
constexpr float JHI[6] = {lcrm::JNT_HI[0], lcrm::JNT_HI[1], lcrm::JNT_HI[2], lcrm::JNT_HI[3], lcrm::JNT_HI[4], lcrm::JNT_HI[5]};
// soft-constraint parameters (MuJoCo defaults solref=(0.02,1), solimp=(0.9,0.95,0.001,0.5,2); follower.xml:15 fingers)
// K = 1/(dmax^2 tc^2), B = 2/(dmax tc)
constexpr float K_DEF = 1.0f / (0.95f * 0.95f * 0.02f * 0.02f), B_DEF = 2.0f / (0.95f * 0.02f);
constexpr float D0_DEF = 0.9f, DW_DEF = 0.95f, W_DEF = 0.001f;
constexpr float K_FC = 1.0f / (0.975f * 0.975f * 0.02f * 0.02f), B_FC = 2.0f / (0.975f * 0.02f);  // finger-cube (mixed)
constexpr float D0_FC = 0.4575f, DW_FC = 0.975f, W_FC = 0.0185f;
constexpr float K_FF = 1.0f / (0.9999f * 0.9999f * 0.02f * 0.02f), B_FF = 2.0f / (0.9999f * 0.02f);  // finger-floor
constexpr float D0_FF = 0.015f, DW_FF = 0.9999f, W_FF = 0.036f;
constexpr float MU_FINGER = 1.5f, MU_TORS = 0.005f;  // finger geom (follower.xml:15); the cube geom's friction is per task (LcrDev)
constexpr float RT_FF = (MU_FINGER * MU_FINGER) / (MU_TORS * MU_TORS);
// PushCubeLoop rails (push_cube_loop.xml:44-47): inner faces of the four wall boxes and their top
constexpr float WALL_X = 0.115f, WALL_Y0 = 0.10f, WALL_Y1 = 0.17f, WALL_TOP = 0.012f;
constexpr float WALL_THICK = 0.02f;   // the rail boxes are 2 x 0.01 thick (push_cube_loop.xml:45-48): their outer faces lie WALL_THICK beyond the inner ones
constexpr float INVW_DOF[6] = {lcrm::INVW_DOF1, lcrm::INVW_DOF2, lcrm::INVW_DOF3, lcrm::INVW_DOF4, lcrm::INVW_DOF5, lcrm::INVW_DOF6};

// MuJoCo impedance curve, power 2, midpoint 0.5 (see oracle kbi()): y = 2x^2 for x <= 0.5, 1 - 2(1-x)^2 above.  Branch-free:
// with t = min(x, 1-x) both halves are 0.5 -/+ (0.5 - 2t^2), the sign being that of x - 0.5 (a divergent if/else costs two
// exec-mask round trips per call, and there are about ten calls per substep).
DEV float impedance(float dist, float d0, float dw, float inv_width) {
    const float x = fminf(fabsf(dist) * inv_width, 1.0f);
    const float t = fminf(x, 1.0f - x);
    const float u = fmaf(-2.0f * t, t, 0.5f);                 // >= 0
    const float y = 0.5f + copysignf(u, x - 0.5f);
    return fmaf(y, dw - d0, d0);
}

// ------------------------------------------------------------------------------------------------
// The contact solve of a sweep (round 4; oracle: pgs_block_pg, orc_params.cone = 3).  Each contact block -- normal row, friction rows -- takes ONE
// PROJECTED-GRADIENT STEP in the variables y = (f_n, f_j / mu_j), in which MuJoCo's elliptic cone (follower.xml:3 cone="elliptic") is the second-order cone:
//     y <- P_K^D( y - D^-1 g~ ),   D = diag(Ln, Lt, .., Lt),  Ln = 2 (A + R)_nn,  Lt = 2 sum_j mu_j^2 (A + R)_jj
// (D majorises the scaled block, so a step never increases the dual objective), with the closed-form D-projection onto the cone: for v = y - D^-1 g~, N = |v_t|:
//     N <= v_n: v;   else  y_n = max(0, w v_n + (1 - w) N),  y_t = v_t y_n / N,   w = Ln / (Ln + Lt).
// Fixed points are exactly the optima of MuJoCo's convex constraint problem.  (Rounds 1-3 updated the rows one by one and then scaled the friction rows radially
// onto the cone with the normal already clamped: its fixed points are not the optima -- a contact whose optimum needs a friction-supported normal force, mu > 1: the
// sliding fingers, stayed at zero force; tools/kkt_distance.py measures p90 6e-3 rad per control step against the exact optimum there, 2e-4 with this step.)
// All rows of a block are evaluated from the same forces: no serial dependence inside a block.  In force units, select-free:
//     f_n' = f_n - u_n iLn,  f_j' = f_j - (mu_j^2 iLt) u_j,  N = |(f_j' / mu_j)|,  f_n = max(f_n', w f_n' + (1 - w) N, 0),  f_j = f_j' min(1, f_n / N)
// k[0] = iLn, k[1] = mu_tan^2 iLt, k[2] = w, k[3] = mu_tors^2 iLt (, k[4] = mu_roll^2 iLt); a block that is off has k = 0 and f = 0: its updates are exact zeros.
// PushCubeLoop keeps the row-wise sweeps (lcr_kernels_loop.hip): its cube has torsional and rolling coefficients of 1.5 m (push_cube_loop.xml:31) whose scaled curvature
// mu^2 / I would set Lt, i.e. the step of every friction row of the block.
template <int NR>
DEV void soc_step(const float (&f)[NR], const float (&u)[NR], const float (&k)[NR], float im_tan2, float im_tors2, float im_roll2, int nrow, float (&nf)[NR]) {
    float fp[NR];
    fp[0] = fmaf(-u[0], k[0], f[0]);
    fp[1] = fmaf(-u[1], k[1], f[1]);
    fp[2] = fmaf(-u[2], k[1], f[2]);
    float s2 = (fp[1] * fp[1] + fp[2] * fp[2]) * im_tan2, s2s = 0.f;
    if (NR > 3) { fp[3] = nrow > 3 ? fmaf(-u[3], k[3], f[3]) : 0.f; s2s = fp[3] * fp[3] * im_tors2; }
    if constexpr (NR > 4) {
        fp[4] = nrow > 4 ? fmaf(-u[4], k[4], f[4]) : 0.f;
        fp[5] = nrow > 4 ? fmaf(-u[5], k[4], f[5]) : 0.f;
        s2s = fmaf(fp[4] * fp[4] + fp[5] * fp[5], im_roll2, s2s);
    }
    s2 += s2s;
    const float rs = rsq(fmaxf(s2, 1e-30f)), N = s2 * rs;
    const float a = fmaf(k[2], fp[0] - N, N);
    const float y0 = fmaxf(fmaxf(fp[0], a), 0.f);
    const float sc = fminf(y0 * rs, 1.f);
    nf[0] = y0;
#pragma unroll
    for (int r = 1; r < NR; r++) nf[r] = fp[r] * sc;
}

// contact frame from unit normal (MuJoCo mju_makeFrame): t1, t2
// (D7) PushCubeLoop rails act as the inner faces of the four wall boxes on the cube vertices below the wall top -- as long as the cube CENTRE is inside the
// outer rectangle of the rails (inner faces + the boxes' thickness).  A cube that was knocked over a rail lies outside the pen untouched, as next to the
// reference's wall boxes, instead of being "deep inside" a half-space (which ejected it at up to 1 200 m/s: 0.5 % of the env-states of a random-policy run
// had the cube out there).  The penetration a rail can see is thereby bounded by thickness + half a cube diagonal.
// (D7, round 5: the rails as BOXES for the arm.)  The surface of the world a point p with a margin r (a sphere's radius; 0 for a pad vertex) is deepest inside: the floor
// (depth r - p.z, normal +z, code 0) or one of the four rail boxes (push_cube_loop.xml:45-48: left / right |x| in [0.115, 0.135], y in [0.08, 0.19]; bottom / top
// |x| < 0.125, y in [0.08, 0.10] / [0.17, 0.19]; top at z = 0.012), inflated by r -- inside one, the face it is shallowest below: the top (code 1 + 5 b) or a side
// (+x, -x, +y, -y: 2 .. 5 + 5 b).  A finger that comes in sideways at floor height is stopped by the side face (rounds 2-4 knew the top faces only and lifted it).
// oracle: world_surface
constexpr float RAIL_CLAMP = 0.012f;   // largest penetration a rail's inner face reports for the CUBE (a cube re-entering the outer rectangle is not shot in)
struct WorldHit { float depth; f3 n; int code; };
template <bool WALLS>
DEV WorldHit world_surface(f3 p, float r) {
    WorldHit h;
    h.depth = r - p.z; h.n = mk(0.f, 0.f, 1.f); h.code = 0;
    if constexpr (WALLS) {
        const float rcx[4] = {-(WALL_X + 0.5f * WALL_THICK), WALL_X + 0.5f * WALL_THICK, 0.f, 0.f};
        const float rcy[4] = {0.5f * (WALL_Y0 + WALL_Y1), 0.5f * (WALL_Y0 + WALL_Y1), WALL_Y0 - 0.5f * WALL_THICK, WALL_Y1 + 0.5f * WALL_THICK};
        const float rhx[4] = {0.5f * WALL_THICK, 0.5f * WALL_THICK, WALL_X + 0.5f * WALL_THICK, WALL_X + 0.5f * WALL_THICK};
        const float rhy[4] = {0.5f * (WALL_Y1 - WALL_Y0) + WALL_THICK, 0.5f * (WALL_Y1 - WALL_Y0) + WALL_THICK, 0.5f * WALL_THICK, 0.5f * WALL_THICK};
        const float ez = WALL_TOP + r - p.z;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const float dx = p.x - rcx[b], dy = p.y - rcy[b];
            const float ex = rhx[b] + r - fabsf(dx), ey = rhy[b] + r - fabsf(dy);
            const bool in = ex > 0.f && ey > 0.f && ez > 0.f;
            float d = ez; int f = 0;
            if (ex < d) { d = ex; f = dx < 0.f ? 2 : 1; }
            if (ey < d) { d = ey; f = dy < 0.f ? 4 : 3; }
            const bool take = in && d > h.depth;
            h.depth = take ? d : h.depth;
            h.code = take ? 1 + 5 * b + f : h.code;
            h.n = take ? mk(f == 1 ? 1.f : (f == 2 ? -1.f : 0.f), f == 3 ? 1.f : (f == 4 ? -1.f : 0.f), f == 0 ? 1.f : 0.f) : h.n;
        }
    }
    return h;
}
DEV bool cube_in_pen(f3 c) {
    return fabsf(c.x) < WALL_X + WALL_THICK && c.y > WALL_Y0 - WALL_THICK && c.y < WALL_Y1 + WALL_THICK;
}
DEV void make_frame(f3 n, f3 &t1, f3 &t2) {
    f3 y = (n.y < 0.5f && n.y > -0.5f) ? mk(0.f, 1.f, 0.f) : mk(0.f, 0.f, 1.f);
    float d = dot(n, y);
    y = axpy(-d, n, y);
    float il = rsq(dot(y, y));
    t1 = il * y;
    t2 = cross(n, t1);
}

struct Sym3 { float xx, xy, xz, yy, yz, zz; };
DEV f3 symv(const Sym3 &S, f3 v) {  // columns of a symmetric matrix: (xx,xy | xz) v.x + (xy,yy | yz) v.y + (xz,yz | zz) v.z
    const f2v xy = f2v{S.xx, S.xy} * f2v{v.x, v.x} + f2v{S.xy, S.yy} * f2v{v.y, v.y} + f2v{S.xz, S.yz} * f2v{v.z, v.z};
    return mk2(xy, fmaf(S.xz, v.x, fmaf(S.yz, v.y, S.zz * v.z)));
}
// w x (w x r) = w (w.r) - r |w|^2   (ww = |w|^2 is shared by the two uses per link)
DEV f3 wxwxr(f3 w, f3 r, float ww) { return axpy(dot(w, r), w, (-ww) * r); }
// world inertia about the link com: R Ic R^T with R = [X Y Z]
DEV Sym3 world_inertia(f3 X, f3 Y, f3 Z, float ixx, float ixy, float ixz, float iyy, float iyz, float izz) {
    f3 Tx = axpy(ixx, X, axpy(ixy, Y, ixz * Z));
    f3 Ty = axpy(ixy, X, axpy(iyy, Y, iyz * Z));
    f3 Tz = axpy(ixz, X, axpy(iyz, Y, izz * Z));
    Sym3 S;
    S.xx = fmaf(Tx.x, X.x, fmaf(Ty.x, Y.x, Tz.x * Z.x));
    S.xy = fmaf(Tx.x, X.y, fmaf(Ty.x, Y.y, Tz.x * Z.y));
    S.xz = fmaf(Tx.x, X.z, fmaf(Ty.x, Y.z, Tz.x * Z.z));
    S.yy = fmaf(Tx.y, X.y, fmaf(Ty.y, Y.y, Tz.y * Z.y));
    S.yz = fmaf(Tx.y, X.z, fmaf(Ty.y, Y.z, Tz.y * Z.z));
    S.zz = fmaf(Tx.z, X.z, fmaf(Ty.z, Y.z, Tz.z * Z.z));
    return S;
}

// ------------------------------------------------------------------------------------------------
// 6x6 dense helpers (fully unrolled; everything lives in VGPRs)
// ------------------------------------------------------------------------------------------------
struct Chol6 {
    float L[6][6];  // strictly-lower part used
    float id[6];    // 1 / L_ii
};
DEV void chol6(const float (&A)[6][6], Chol6 &C) {  // A symmetric, lower part read
#pragma unroll
    for (int j = 0; j < 6; j++) {
        float d = A[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) d = fmaf(-C.L[j][k], C.L[j][k], d);
        float id = rsq(d);
        C.id[j] = id;
#pragma unroll
        for (int i = j + 1; i < 6; i++) {
            float s = A[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) s = fmaf(-C.L[i][k], C.L[j][k], s);
            C.L[i][j] = s * id;
        }
    }
}
DEV void fsub(const Chol6 &C, float (&x)[6]) {  // x <- L^-1 x
#pragma unroll
    for (int i = 0; i < 6; i++) {
        float s = x[i];
#pragma unroll
        for (int k = 0; k < i; k++) s = fmaf(-C.L[i][k], x[k], s);
        x[i] = s * C.id[i];
    }
}
DEV void bsub(const Chol6 &C, float (&x)[6]) {  // x <- L^-T x
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        float s = x[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) s = fmaf(-C.L[k][i], x[k], s);
        x[i] = s * C.id[i];
    }
}

// ------------------------------------------------------------------------------------------------
// numpy-compatible PCG64 (Generator(PCG64(SeedSequence(seed)))) -- reset sampling must reproduce
// self.np_random.uniform(low, high) of reach_cube_env.py:302 bit for bit
// ------------------------------------------------------------------------------------------------
typedef unsigned __int128 u128;
DEV u128 pcg_mult() { return (((u128)0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL; }
struct Pcg { u128 st, inc; };
DEV double pcg_double(Pcg &g) {
    g.st = g.st * pcg_mult() + g.inc;
    unsigned long long hi = (unsigned long long)(g.st >> 64), lo = (unsigned long long)g.st, x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    unsigned long long o = (x >> rot) | (x << ((64 - rot) & 63));
    return __dmul_rn((double)(o >> 11), 1.0 / 9007199254740992.0);
}
DEV uint32_t ss_hashmix(uint32_t v, uint32_t &hc) { v ^= hc; hc *= 0x931e8875u; v *= hc; v ^= v >> 16; return v; }
DEV uint32_t ss_mix(uint32_t x, uint32_t y) { uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y; r ^= r >> 16; return r; }
__attribute__((unused)) DEV Pcg pcg_seed(unsigned long long seed) {  // SeedSequence(seed).generate_state(4, uint64) -> pcg64 srandom
    uint32_t ent0 = (uint32_t)seed, ent1 = (uint32_t)(seed >> 32);
    uint32_t pool[4], hc = 0x43b0d7e5u;
    pool[0] = ss_hashmix(ent0, hc);
    pool[1] = ss_hashmix(ent1, hc);  // a zero high word hashes exactly like the implicit zero padding
    pool[2] = ss_hashmix(0u, hc);
    pool[3] = ss_hashmix(0u, hc);
#pragma unroll
    for (int s = 0; s < 4; s++)
#pragma unroll
        for (int d = 0; d < 4; d++)
            if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], hc));
    uint32_t w[8], hb = 0x8b51f9ddu;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint32_t v = pool[i & 3];
        v ^= hb; hb *= 0x58f38dedu; v *= hb; v ^= v >> 16; w[i] = v;
    }
    unsigned long long s0 = (unsigned long long)w[0] | ((unsigned long long)w[1] << 32);
    unsigned long long s1 = (unsigned long long)w[2] | ((unsigned long long)w[3] << 32);
    unsigned long long s2 = (unsigned long long)w[4] | ((unsigned long long)w[5] << 32);
    unsigned long long s3 = (unsigned long long)w[6] | ((unsigned long long)w[7] << 32);
    u128 initstate = ((u128)s0 << 64) | s1, initseq = ((u128)s2 << 64) | s3;
    Pcg g;
    g.inc = (initseq << 1) | 1;
    g.st = 0;
    g.st = g.st * pcg_mult() + g.inc;
    g.st += initstate;
    g.st = g.st * pcg_mult() + g.inc;
    return g;
}

// ------------------------------------------------------------------------------------------------
// per-env state held in registers
// ------------------------------------------------------------------------------------------------
template <int NC>
struct EnvState {
    float q[6], qd[6];
    f3 cp[NC];       // cube positions
    float cq[NC][4]; // cube quaternions (w,x,y,z)
    f3 cv[NC];       // cube linear velocity (world)
    f3 cw[NC];       // cube angular velocity (BODY frame, MuJoCo free-joint convention)
};

template <int NC>
DEV void load_state(const LcrDev &P, int e, EnvState<NC> &S) {
    const int N = P.n;
#pragma unroll
    for (int j = 0; j < 6; j++) { S.q[j] = P.qpos[j * N + e]; S.qd[j] = P.qvel[j * N + e]; }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        const float *qp = P.qpos + (size_t)(6 + 7 * c) * N + e;
        const float *qv = P.qvel + (size_t)(6 + 6 * c) * N + e;
        S.cp[c] = mk(qp[0], qp[N], qp[2 * N]);
#pragma unroll
        for (int k = 0; k < 4; k++) S.cq[c][k] = qp[(3 + k) * N];
        S.cv[c] = mk(qv[0], qv[N], qv[2 * N]);
        S.cw[c] = mk(qv[3 * N], qv[4 * N], qv[5 * N]);
    }
}
template <int NC>
DEV void store_state(const LcrDev &P, int e, const EnvState<NC> &S) {
    const int N = P.n;
#pragma unroll
    for (int j = 0; j < 6; j++) { P.qpos[j * N + e] = S.q[j]; P.qvel[j * N + e] = S.qd[j]; }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        float *qp = P.qpos + (size_t)(6 + 7 * c) * N + e;
        float *qv = P.qvel + (size_t)(6 + 6 * c) * N + e;
        qp[0] = S.cp[c].x; qp[N] = S.cp[c].y; qp[2 * N] = S.cp[c].z;
#pragma unroll
        for (int k = 0; k < 4; k++) qp[(3 + k) * N] = S.cq[c][k];
        qv[0] = S.cv[c].x; qv[N] = S.cv[c].y; qv[2 * N] = S.cv[c].z;
        qv[3 * N] = S.cw[c].x; qv[4 * N] = S.cw[c].y; qv[5 * N] = S.cw[c].z;
    }
}


// ------------------------------------------------------------------------------------------------
// arm-coupled contact slot: finger sphere vs cube (HASCUBE) or vs floor
// ------------------------------------------------------------------------------------------------
typedef float float2v __attribute__((ext_vector_type(2)));

template <int NRW>
struct ArmSlot {
    f3 n, t1, t2, rc;  // frame and contact point relative to the cube centre (cube slots only)
    float f[NRW], aref[NRW], inv[NRW];   // rows: normal, two tangents, torsion (NRW = 6: + two rolling rows, finger<->cube slots of the ROLL kernels)
    float Rn;
    bool act;
};

// constraint forces carried from one substep to the next -- and, through LcrDev::warm, from one control step to the next (zero after
// a reset or lcr_set_state; zero at every control step with LCR_COMPAT_COLD_SOLVE_EACH_STEP): warm start of the PGS sweeps, as MuJoCo
// warm-starts its solver from mjData.qacc_warmstart.  Cube<->cube forces (Stack) persist in their LDS records.
// arm-coupled contact slots: 0,1 finger sphere 0/1 vs cube; 2,3 finger sphere 0/1 vs floor; 4 the arm-link proxies (D3): one
// contact, vs floor or (gripper body) cube per lane, 4 rows (the torsion row only on a cube: link<->floor is condim 3)
constexpr int NAS = 5;
// ROLL kernels (lcr_config.finger_cube_condim = 6): the finger<->cube slots 0, 1 carry MuJoCo's two rolling-friction rows as well
// (follower.xml:15 condim="6"; deviation D4 is then limited to the finger<->floor contacts)
template <bool ROLL> constexpr int as_rows(int s) { return (ROLL && s < 2) ? 6 : 4; }
// first LDS row of a slot (Stack keeps four LDS rows per slot in every variant: its rolling rows live in the global scratch)
// (BIG: the Stack variant for shards of at most three waves per CU, which keeps every g row in LDS)
template <bool ROLL, int NC, bool BIG> constexpr int as_row0(int s) { return (ROLL && (NC == 1 || BIG)) ? (s < 2 ? 6 * s : 12 + 4 * (s - 2)) : 4 * s; }
constexpr int AS_TOTAL_ROWS = 20;
// layout of LcrDev::warm ([LCR_NWARM][N] floats): the Warm fields of one env between two control steps
constexpr int WARM_FLOOR = 0, WARM_ARM = 32, WARM_LIM = 62, WARM_WALL = 68, WARM_CC = 84, WARM_CCPREV = 100, WARM_CC2 = 104, WARM_CCPREV2 = 120;   // (.._CC2: cube<->cube slots 4-7 of the eight-point manifold)   // LCR_DEV_NWARM = 124 (lcr_device.h) = LCR_NWARM (include/lcr.h, where the layout is part of the ABI)
template <int NC, int NRW>
struct Warm {
    float floor[NC][4][4];
    float arm[NAS][NRW];
    float lim[6];
    float wall[4][4];
    bool cc_prev[8];   // (slots 4-7: the eight-point manifold of the Newton kernels)
};

constexpr int LDS_ROW = 6 * 64;            // one g row for 64 lanes
constexpr int LDS_G_FLOATS = AS_TOTAL_ROWS * LDS_ROW;  // 20 rows -> 30 KiB per wave (4 waves per CU: 120 of 160 KiB)
// Stack only: cube<->cube contact records, 4 slots x 16 floats per lane: pos3 f4 aref4 inv4 Rn (LDS, see LdsSize)
constexpr int CC_REC = 16;
constexpr int CC_REC_NEWTON = 12, CC_RN_NEWTON = 11;   // the Newton kernels keep no inverse-diagonal fields (11-14 of the sweep kernels' record): pos 0-2, force 3-6, aref 7-10, Rn 11 -- 8 KiB less LDS per StackTwoCubes wave
// The per-substep constants of the floor<->cube slots of cube 0 (aref[4], inv[4]: written once per substep, read once per PGS
// sweep) are parked in LDS as 16-B vectors instead of occupying 32 registers across the whole solver loop:
// [slot 0..3][aref|inv][lane][4] = 8 KiB per wave (30 + 8 = 38 of the 40 KiB a wave may use at four waves per CU).
constexpr int LDS_PARK_FLOATS = 4 * 2 * 64 * 4;
// Stack (two cubes): the 40 KiB hold the 16 g rows of the four finger slots (24 KiB) and the four cube<->cube contact records
// (16 floats per lane each, 16 KiB: they are read and written by every sweep of the waves that determine the launch time);
// the four g rows of its arm-link proxy slot live in a coalesced global scratch array instead ([12][N] float2, L1/L2 resident).
constexpr int LDS_CC_FLOATS = 4 * CC_REC * 64;
// ROLL kernels: one cube: 24 g rows (36 KiB), no parking; Stack: the four rolling rows join the proxy slot's rows in the global scratch
// BIG (Stack, <= 3 waves per CU, i.e. <= 49 152 envs on an MI355X; the per-GPU shard of BASELINE config 5 is 32 768): all 20 / 24 g rows and
// the cube<->cube records in LDS (46 / 52 KiB per wave), nothing in the global scratch
template <int NC, bool BIG, bool ROLL> constexpr int cc_base_rows() { return (NC == 2 && BIG) ? (ROLL ? 24 : 20) : 16; }
template <int NC, bool WALLS, bool ROLL, bool BIG> struct LdsSize {
    static constexpr int value = NC == 2 ? cc_base_rows<NC, BIG, ROLL>() * LDS_ROW + LDS_CC_FLOATS : (ROLL ? 24 * LDS_ROW : LDS_G_FLOATS + LDS_PARK_FLOATS);
};
typedef float float4v __attribute__((ext_vector_type(4)));

// sphere (centre, radius) vs cube box: signed distance, world normal (box -> sphere) and contact point midway between the surfaces
struct SBHit { float dist; f3 n, pos; int code; };   // code: 7 centre outside the box, else 2 * face axis + (negative side)
DEV SBHit sphere_box(f3 centre, float rad, f3 cp, const CubeRot &R) {
    f3 d = centre - cp;
    f3 l = mk(dot(R.X, d), dot(R.Y, d), dot(R.Z, d));
    f3 qv = mk(clampf(l.x, -CH, CH), clampf(l.y, -CH, CH), clampf(l.z, -CH, CH));
    bool outside = (l.x != qv.x) || (l.y != qv.y) || (l.z != qv.z);
    f3 df = l - qv;
    float dn2 = dot(df, df);
    float idn = rsq(fmaxf(dn2, 1e-30f));
    float dn = dn2 * idn;
    f3 nl_out = idn * df;
    // centre inside the box: face of least depth (first minimal index)
    float dx = CH - fabsf(l.x), dy = CH - fabsf(l.y), dz = CH - fabsf(l.z);
    int best = 0; float bd = dx;
    if (dy < bd) { bd = dy; best = 1; }
    if (dz < bd) { bd = dz; best = 2; }
    float sg = ((best == 0 ? l.x : (best == 1 ? l.y : l.z)) < 0.f) ? -1.f : 1.f;
    f3 nl_in = mk(best == 0 ? sg : 0.f, best == 1 ? sg : 0.f, best == 2 ? sg : 0.f);
    f3 q_in = mk(best == 0 ? sg * CH : l.x, best == 1 ? sg * CH : l.y, best == 2 ? sg * CH : l.z);
    f3 nl = outside ? nl_out : nl_in;
    f3 ql = outside ? qv : q_in;
    float dd = outside ? dn - rad : -(bd + rad);
    f3 pl = axpy(0.5f * dd, nl, ql);
    SBHit h;
    h.code = outside ? 7 : 2 * best + (sg < 0.f ? 1 : 0);
    h.dist = dd;
    h.n = axpy(nl.x, R.X, axpy(nl.y, R.Y, nl.z * R.Z));
    h.pos = axpy(pl.x, R.X, axpy(pl.y, R.Y, axpy(pl.z, R.Z, cp)));
    return h;
}

// ---- finger pads as boxes (the faithful preset; oracle: collide_plane_pad / collide_box_pad, orc_params.finger_geom = 1): one box per finger fitted to the two outermost
// slabs of the finger's collision hull (lcr_model_gen.h PAD*; follower.xml:15,89,97).  Against the floor the lowest vertex decides, against a cube vertex-in-box tests both
// ways (the pad's vertices in the cube, the cube's in the pad); the vertices of the same face within PAD_BLEND of the deepest one share the contact point. ----
constexpr float PAD_BLEND = 0.0005f;
struct PadBox { f3 c, ax, ay, az; };   // centre and the link's axes scaled by the half extents
DEV f3 pad_vertex(const PadBox &B, int i) { return B.c + ((i & 1) ? B.ax : neg(B.ax)) + ((i & 2) ? B.ay : neg(B.ay)) + ((i & 4) ? B.az : neg(B.az)); }
struct PadFloorHit { float dist; f3 pos, n; int code; };
template <bool WALLS>
DEV PadFloorHit pad_floor(const PadBox &B) {
    if constexpr (!WALLS) {   // the floor only: heights of the eight vertices, nothing else (same result as the general path below; it costs the five tasks without rails
                              // 10 % of their step when they run it -- registers, not arithmetic)
        float z[8];
        f2v xy[8];
        int best = 0;
        float zb = 1e30f;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const f3 v = pad_vertex(B, i);
            z[i] = v.z; xy[i] = v.xy;
            if (z[i] < zb) { zb = z[i]; best = i; }
        }
        float wsum = 0.f;
        f2v q = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const float w = fmaxf(zb - z[i] + PAD_BLEND, 0.f);
            wsum += w; q = f2v{w, w} * xy[i] + q;
        }
        const float iw = rcp(fmaxf(wsum, 1e-30f));
        PadFloorHit h;
        h.dist = zb; h.n = mk(0.f, 0.f, 1.f);
        h.pos = mk(q.x * iw, q.y * iw, 0.5f * zb);
        h.code = best;
        return h;
    }
    float depth[8];
    int code[8];
    f3 v[8], nn[8];
    int best = 0;
    float bd = -1e30f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        v[i] = pad_vertex(B, i);
        const WorldHit w = world_surface<WALLS>(v[i], 0.f);
        depth[i] = w.depth; code[i] = w.code; nn[i] = w.n;
        if (depth[i] > bd) { bd = depth[i]; best = i; }
    }
    int cb = 0;
    f3 nb = mk(0.f, 0.f, 1.f), vb = v[0];
#pragma unroll
    for (int i = 0; i < 8; i++) { cb = best == i ? code[i] : cb; nb = best == i ? nn[i] : nb; vb = best == i ? v[i] : vb; }
    float wsum = 0.f;
    f3 q = mk(0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 8; i++) {   // the vertices on the SAME surface within PAD_BLEND of the deepest one share the contact point
        const float w = (code[i] == cb) ? fmaxf(depth[i] - bd + PAD_BLEND, 0.f) : 0.f;
        wsum += w; q = axpy(w, v[i], q);
    }
    q = rcp(fmaxf(wsum, 1e-30f)) * q;
    PadFloorHit h;
    h.dist = -bd; h.n = nb;
    h.pos = axpy(dot(vb, nb) - dot(q, nb) + 0.5f * bd, nb, q);   // the blended point at the deepest vertex' level along the normal, then midway to the surface
    h.code = best + 8 * cb;
    return h;
}
// point q inside the box (centre c, unit axes X, Y, Z, half extents h)?  depth to the nearest face, that face (2 k + (negative side)), local coordinates
DEV bool point_in_box(f3 q, f3 c, f3 X, f3 Y, f3 Z, f3 h, float &depth, int &face, f3 &l) {
    const f3 d = q - c;
    l = mk(dot(X, d), dot(Y, d), dot(Z, d));
    const float dx = h.x - fabsf(l.x), dy = h.y - fabsf(l.y), dz = h.z - fabsf(l.z);
    int k = 0; float bd = dx;
    if (dy < bd) { bd = dy; k = 1; }
    if (dz < bd) { bd = dz; k = 2; }
    depth = bd;
    face = 2 * k + (((k == 0 ? l.x : (k == 1 ? l.y : l.z)) < 0.f) ? 1 : 0);
    return dx > 0.f && dy > 0.f && dz > 0.f;
}
// pad box (link axes PX, PY, PZ unit, half extents ph, centre pc) vs cube: SBHit as sphere_box, code = (kind * 6 + face) * 8 + vertex
DEV SBHit pad_box(f3 pc, f3 PX, f3 PY, f3 PZ, f3 ph, f3 cp, const CubeRot &R) {
    const f3 hc = mk(CH, CH, CH);
    float depth[16];
    int face[16];
    bool in[16];
    f3 loc[16];
    int best = -1;
    float bd = -1.f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (i < 8) {
            const f3 w = pc + ((i & 1) ? ph.x : -ph.x) * PX + ((i & 2) ? ph.y : -ph.y) * PY + ((i & 4) ? ph.z : -ph.z) * PZ;
            in[i] = point_in_box(w, cp, R.X, R.Y, R.Z, hc, depth[i], face[i], loc[i]);
        } else {
            const int j = i - 8;
            const f3 u = cp + ((j & 1) ? CH : -CH) * R.X + ((j & 2) ? CH : -CH) * R.Y + ((j & 4) ? CH : -CH) * R.Z;
            in[i] = point_in_box(u, pc, PX, PY, PZ, ph, depth[i], face[i], loc[i]);
        }
        if (in[i] && depth[i] > bd) { bd = depth[i]; best = i; }
    }
    SBHit h;
    h.dist = 1.f; h.n = mk(0.f, 0.f, 1.f); h.pos = mk(0.f, 0.f, 0.f); h.code = 0;
    if (best < 0) return h;
    const bool typeB = best >= 8;
    int fb = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) fb = best == i ? face[i] : fb;
    const int k = fb >> 1;
    const float sg = (fb & 1) ? -1.f : 1.f;
    float wsum = 0.f;
    f3 q = mk(0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const bool same = in[i] && (i >= 8) == typeB && face[i] == fb;
        const float w = same ? fmaxf(depth[i] - bd + PAD_BLEND, 0.f) : 0.f;
        wsum += w; q = axpy(w, loc[i], q);
    }
    q = rcp(fmaxf(wsum, 1e-30f)) * q;
    const float dist = -bd;
    const float hk = typeB ? (k == 0 ? ph.x : (k == 1 ? ph.y : ph.z)) : CH;
    const float qk = sg * hk + sg * dist * 0.5f;
    q = mk(k == 0 ? qk : q.x, k == 1 ? qk : q.y, k == 2 ? qk : q.z);
    const f3 nl = mk(k == 0 ? sg : 0.f, k == 1 ? sg : 0.f, k == 2 ? sg : 0.f);
    if (!typeB) {
        h.n = axpy(nl.x, R.X, axpy(nl.y, R.Y, nl.z * R.Z));
        h.pos = axpy(q.x, R.X, axpy(q.y, R.Y, axpy(q.z, R.Z, cp)));
    } else {
        h.n = neg(axpy(nl.x, PX, axpy(nl.y, PY, nl.z * PZ)));
        h.pos = axpy(q.x, PX, axpy(q.y, PY, axpy(q.z, PZ, pc)));
    }
    h.dist = dist;
    h.code = ((typeB ? 6 : 0) + fb) * 8 + (best & 7);
    return h;
}

// per-env diagnostics of one control step (written only when LcrDev.diag): which constraint slots were active (bit = slot id:
// 0-7 floor<->cube, 8-11 cube<->cube / rails, 12-13 finger<->cube, 14-15 finger<->floor, 16 arm-link proxies, 18+j limit j),
// the number of (slot, substep) activations, and the largest PGS sweep count of a substep
struct Diag { unsigned mask, count, sweeps, choice; };
// choice: wrapping sum over substeps s (weight 2s + 1) and active constraints of (slot + 1)(sel + 1) 2654435761, sel = the discrete choice behind
// the contact (vertex index, manifold candidate, box face case, proxy member, limit side) -- same formula as the oracle's
DEV void diag_choice(Diag &DG, bool act, int slot, int sel) { DG.choice += act ? (unsigned)(slot + 1) * (unsigned)(sel + 1) * 2654435761u : 0u; }


// ------------------------------------------------------------------------------------------------
// reset of one env (reach_cube_env.py:297-311 etc.); sampling in fp64 exactly as numpy does
// ------------------------------------------------------------------------------------------------
template <int NC>
DEV void reset_env(const LcrDev &P, EnvState<NC> &S, Pcg &g, f3 &target, f3 &ee_lag, int goal) {
#pragma unroll
    for (int c = 0; c < NC; c++) {
        // low + range*u with separately rounded product and sum (numpy's random_uniform); never contracted into an fma
        double x = __dadd_rn(P.cube_lo[0], __dmul_rn(P.cube_rng[0], pcg_double(g)));
        double yv = __dadd_rn(P.cube_lo[1], __dmul_rn(P.cube_rng[1], pcg_double(g)));
        double zv = __dadd_rn(P.cube_lo[2], __dmul_rn(P.cube_rng[2], pcg_double(g)));
        if (P.walls) {  // push_cube_loop_env.py:304-308: the sample is centred on the current goal region
            x = __dadd_rn(x, goal ? -0.06 : 0.06);
            yv = __dadd_rn(yv, 0.135);
        }
        S.cp[c] = mk((float)x, (float)yv, (float)zv);
        S.cq[c][0] = 1.f; S.cq[c][1] = 0.f; S.cq[c][2] = 0.f; S.cq[c][3] = 0.f;
    }
    if (P.has_target) {
        double x = __dadd_rn(P.tgt_lo[0], __dmul_rn(P.tgt_rng[0], pcg_double(g)));
        double yv = __dadd_rn(P.tgt_lo[1], __dmul_rn(P.tgt_rng[1], pcg_double(g)));
        double zv = __dadd_rn(P.tgt_lo[2], __dmul_rn(P.tgt_rng[2], pcg_double(g)));
        target = mk((float)x, (float)yv, (float)zv);
    }
#pragma unroll
    for (int j = 0; j < 6; j++) S.q[j] = 0.f;
    if (P.compat & 1u) {  // LCR_COMPAT_ZERO_QVEL_ON_RESET; default keeps qvel (the reference has no mj_resetData)
#pragma unroll
        for (int j = 0; j < 6; j++) S.qd[j] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; c++) { S.cv[c] = mk(0.f, 0.f, 0.f); S.cw[c] = mk(0.f, 0.f, 0.f); }
    }
    // mj_forward at q = 0 refreshes site_xpos (reach_cube_env.py:309)
    float q0[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    ArmFrames F;
    arm_frames(q0, F);
    ee_lag = site_pos(F);
}

DEV Pcg load_rng(const LcrDev &P, int e) {
    const size_t N = (size_t)P.n;
    Pcg g;
    g.st = ((u128)P.rng[e] << 64) | P.rng[N + e];
    g.inc = ((u128)P.rng[2 * N + e] << 64) | P.rng[3 * N + e];
    return g;
}
DEV void store_rng(const LcrDev &P, int e, const Pcg &g) {
    const size_t N = (size_t)P.n;
    P.rng[e] = (unsigned long long)(g.st >> 64);
    P.rng[N + e] = (unsigned long long)g.st;
    P.rng[2 * N + e] = (unsigned long long)(g.inc >> 64);
    P.rng[3 * N + e] = (unsigned long long)g.inc;
}

// observation vector of the reference (get_observation reach:281-295, push:291-306, stack:290-305)
template <int NC>
DEV void write_obs18(const LcrDev &P, float *dst, int e, const EnvState<NC> &S, f3 target) {
    const int N = P.n;
#pragma unroll
    for (int j = 0; j < 6; j++) { dst[j * N + e] = S.q[j]; dst[(6 + j) * N + e] = S.qd[j]; }
    dst[12 * N + e] = S.cp[0].x; dst[13 * N + e] = S.cp[0].y; dst[14 * N + e] = S.cp[0].z;
    f3 aux = P.has_target ? target : (NC == 2 ? S.cp[NC - 1] : mk(0.f, 0.f, 0.f));
    dst[15 * N + e] = aux.x; dst[16 * N + e] = aux.y; dst[17 * N + e] = aux.z;
}

}  // namespace
