// lcr_arm.h -- device-side vector helpers and the forward kinematics of the fixed arm tree, shared by the step
// kernels (lcr_kernels.hip) and the image renderer (lcr_render.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "lcr_model_gen.h"

#define DEV __device__ __forceinline__

namespace lcrdev {

typedef float f2v __attribute__((ext_vector_type(2)));
// x and y share a 64-bit register pair so that element-wise vector arithmetic issues as packed v_pk_* instructions
struct f3 {
    union {
        struct { float x, y; };
        f2v xy;
    };
    float z;
};
DEV f3 mk(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
DEV f3 mk2(f2v xy, float z) { f3 r; r.xy = xy; r.z = z; return r; }
DEV f3 operator+(f3 a, f3 b) { return mk2(a.xy + b.xy, a.z + b.z); }
DEV f3 operator-(f3 a, f3 b) { return mk2(a.xy - b.xy, a.z - b.z); }
DEV f3 operator*(float s, f3 a) { return mk2(f2v{s, s} * a.xy, s * a.z); }
DEV f3 neg(f3 a) { return mk2(-a.xy, -a.z); }
DEV float dot(f3 a, f3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
DEV f3 cross(f3 a, f3 b) { return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
DEV f3 axpy(float s, f3 a, f3 b) { return mk2(f2v{s, s} * a.xy + b.xy, fmaf(s, a.z, b.z)); }  // s*a+b
DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
DEV float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
// NaN / inf / |x| >= 2^34 (~1.7e10, MuJoCo's mjMAXVAL is 1e10) by exponent bits: immune to -ffast-math
DEV bool bad_value(float x) { return ((__float_as_uint(x) >> 23) & 0xffu) >= 127u + 34u; }

// sin/cos for |x| <~ 4 (joint angles are range-limited, follower.xml:58-95): quadrant reduction with a two-term pi/2 and
// the classic single-precision minimax polynomials on [-pi/4, pi/4]; max abs error 8.5e-8 on [-3.3, 3.3], ~30
// instructions instead of libm sincosf's ~150 (which carries a large-argument path this kernel can never take).
DEV void sincos_small(float x, float *sp_out, float *cp_out) {
    const float k = rintf(x * 0.636619772f);
    float r = fmaf(k, -1.57079637f, x);
    r = fmaf(k, 4.37113883e-8f, r);
    const float r2 = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515296e-4f, r2, 8.3321609e-3f), r2, -1.6666655e-1f), r2 * r, r);
    const float cp = fmaf(fmaf(fmaf(2.4433157e-5f, r2, -1.3887316e-3f), r2, 4.1666646e-2f), r2 * r2, fmaf(-0.5f, r2, 1.0f));
    const int q = (int)k;
    const float ss = (q & 1) ? cp : sp, cc = (q & 1) ? sp : cp;
    *sp_out = (q & 2) ? -ss : ss;
    *cp_out = ((q + 1) & 2) ? -cc : cc;
}


// ------------------------------------------------------------------------------------------------
// arm kinematics: world frames of link_1..link_6 (base quat of follower.xml:51 folded in)
// ------------------------------------------------------------------------------------------------
struct ArmFrames {
    f3 X[6], Y[6], Z[6], p[6];
};

DEV void arm_frames(const float (&q)[6], ArmFrames &F) {
    using namespace lcrm;
    float s, c;
    // base_link: Rz(-90deg): X0=(0,-1,0) Y0=(1,0,0) Z0=(0,0,1)
    const f3 X0 = mk(0.f, -1.f, 0.f), Y0 = mk(1.f, 0.f, 0.f), Z0 = mk(0.f, 0.f, 1.f);
    // link_1: pos (P1x,0,P1z), axis -z
    F.p[0] = axpy(P1x, X0, P1z * Z0);
    sincos_small(q[0], &s, &c);
    F.X[0] = axpy(c, X0, (-s) * Y0);
    F.Y[0] = axpy(s, X0, c * Y0);
    F.Z[0] = Z0;
    // link_2: pos (0,P2y,P2z), axis +y
    F.p[1] = axpy(P2y, F.Y[0], axpy(P2z, F.Z[0], F.p[0]));
    sincos_small(q[1], &s, &c);
    F.X[1] = axpy(c, F.X[0], (-s) * F.Z[0]);
    F.Z[1] = axpy(s, F.X[0], c * F.Z[0]);
    F.Y[1] = F.Y[0];
    // link_3: axis -y
    F.p[2] = axpy(P3x, F.X[1], axpy(P3y, F.Y[1], axpy(P3z, F.Z[1], F.p[1])));
    sincos_small(q[2], &s, &c);
    F.X[2] = axpy(c, F.X[1], s * F.Z[1]);
    F.Z[2] = axpy(-s, F.X[1], c * F.Z[1]);
    F.Y[2] = F.Y[1];
    // link_4: axis +y
    F.p[3] = axpy(P4x, F.X[2], axpy(P4y, F.Y[2], axpy(P4z, F.Z[2], F.p[2])));
    sincos_small(q[3], &s, &c);
    F.X[3] = axpy(c, F.X[2], (-s) * F.Z[2]);
    F.Z[3] = axpy(s, F.X[2], c * F.Z[2]);
    F.Y[3] = F.Y[2];
    // link_5: pos (P5x,P5y,0), axis +x
    F.p[4] = axpy(P5x, F.X[3], axpy(P5y, F.Y[3], F.p[3]));
    sincos_small(q[4], &s, &c);
    F.Y[4] = axpy(c, F.Y[3], s * F.Z[3]);
    F.Z[4] = axpy(-s, F.Y[3], c * F.Z[3]);
    F.X[4] = F.X[3];
    // link_6: axis -z
    F.p[5] = axpy(P6x, F.X[4], axpy(P6y, F.Y[4], axpy(P6z, F.Z[4], F.p[4])));
    sincos_small(q[5], &s, &c);
    F.X[5] = axpy(c, F.X[4], (-s) * F.Y[4]);
    F.Y[5] = axpy(s, F.X[4], c * F.Y[4]);
    F.Z[5] = F.Z[4];
}
DEV f3 joint_axis(const ArmFrames &F, int j) {  // world joint axes (follower.xml:58,65,72,79,86,95); j is a literal after unrolling
    switch (j) {
    case 0: return neg(F.Z[0]);
    case 1: return F.Y[1];
    case 2: return neg(F.Y[2]);
    case 3: return F.Y[3];
    case 4: return F.X[4];
    default: return neg(F.Z[5]);
    }
}
DEV f3 local_point(const ArmFrames &F, int i, float x, float y, float z) {
    return axpy(x, F.X[i], axpy(y, F.Y[i], axpy(z, F.Z[i], F.p[i])));
}
DEV f3 site_pos(const ArmFrames &F) { return local_point(F, 4, lcrm::SITEx, lcrm::SITEy, lcrm::SITEz); }


struct CubeRot { f3 X, Y, Z; };  // columns of the cube rotation matrix
DEV CubeRot quat_to_cols(const float (&q)[4]) {
    float w = q[0], x = q[1], y = q[2], z = q[3];
    CubeRot R;
    R.X = mk(1.f - 2.f * (y * y + z * z), 2.f * (x * y + w * z), 2.f * (x * z - w * y));
    R.Y = mk(2.f * (x * y - w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z + w * x));
    R.Z = mk(2.f * (x * z + w * y), 2.f * (y * z - w * x), 1.f - 2.f * (x * x + y * y));
    return R;
}

// ------------------------------------------------------------------------------------------------
// floor <-> cube contact slot (cube-only rows; frame n=+z, t1=+y, t2=-x)
// ------------------------------------------------------------------------------------------------
struct FloorSlot {
    f3 r;            // contact point relative to cube centre (world)
    float f[4];      // n, t1, t2, torsion
    float aref[4];
    float inv[4];    // 1 / (A_ii + R_i)
    float Rn;
    bool act;
};

}  // namespace lcrdev
