// lcr_newton.h -- building blocks of the FAITHFUL preset's contact solve: Newton's method on the primal problem, MuJoCo's default solver
// (follower.xml:3 names no solver).  Oracle: newton_product in oracle/lcr_oracle.c (orc_params.solver = 2); decision record: profiles/r05_solver_decision.txt.
//
// The constrained accelerations x minimise the strictly convex, C^1, piecewise quadratic
//     F(x) = 1/2 (x - a0)' M (x - a0) + sum_b s_b(J_b x - aref_b),      s_b(z) = max_{f in K_b} ( -f'z - 1/2 f'R_b f ),
// the constraint forces are f_b = argmax.  Unknowns here: the arm in the coordinates y = L' qacc (M = L L': its metric is the identity) and per cube the linear
// and angular acceleration (metric: mass, isotropic inertia).  One iteration: gradient g and Hessian H = M + J'WJ at x (W: Jacobian of -f w.r.t. the row
// residuals), dx = -H^-1 g by Cholesky, line search on phi'(al) = grad F(x + al dx).dx (derivative only: one gradient pass per evaluation; Illinois variant of
// regula falsi after bracketing), x += al dx.  Rounds 1-4 swept per-contact blocks of the DUAL problem; what those sweeps cannot resolve in any sane number of
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
constexpr float MU_ROLL = 1e-4f;
constexpr float RR_FF = (MU_FINGER * MU_FINGER) / (MU_ROLL * MU_ROLL);   // regulariser scale of the rolling rows: mu_tan^2 / mu_roll^2

}  // namespace
