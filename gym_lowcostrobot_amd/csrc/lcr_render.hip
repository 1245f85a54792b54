// lcr_render.hip -- image observations: a small ray-caster for the two 240x320 observation cameras of every env
// (get_observation, envs/reach_cube_env.py:288-292: renderer.update_scene(camera="camera_front"/"camera_top"); render())
// and for the 640x640 `camera_vizu` frame of render() (envs/reach_cube_env.py:350-355).
//
// It is an APPROXIMATE restatement, not MuJoCo's OpenGL renderer (which cannot run here): pinhole cameras with the
// poses of the scene xmls (reach_cube.xml:29-31) and MuJoCo's default fovy 45 deg; checker floor (texrepeat 5 -> 0.1 m
// squares, reach_cube.xml:14-16), gradient sky, the cube(s) as exact oriented boxes, the arm as 7 capsules between the
// link origins (the 20 STL meshes are not shipped), ambient 0.3 + headlight 0.6 Lambert shading, no shadows.
//
// Mapping: a workgroup owns a quarter of one env's 480 image rows (both frames); a wave renders one row at a time (5 pixels
// per lane), stages its 960 bytes in LDS and writes them with ONE non-temporal 16-B store instruction (60 lanes, 960
// contiguous bytes).  The per-env scene (FK of the arm, cube frames, screen-space bounding boxes of every primitive for
// both cameras) is built once per workgroup in LDS; culling is wave-uniform (per row and 64-pixel group), so most pixels
// cost one ray-plane intersection.
#include <hip/hip_runtime.h>

#include "lcr_arm.h"
#include "lcr_device.h"

using namespace lcrdev;

namespace {

constexpr int NCAP = 7;
constexpr int NBOX = 3;   // cube, second cube (Stack), target marker (Push / PickPlace)
constexpr int NPRIM = NCAP + NBOX;

struct Scene {
    f3 ca[NCAP], cb[NCAP];
    float cr[NCAP];
    f3 bc[NBOX], bX[NBOX], bY[NBOX], bZ[NBOX], bh[NBOX];
    f3 bcol[NBOX];
    float balpha[NBOX];
    int nbox;
    // screen-space bounding boxes per camera (x0, x1, y0, y1), inclusive
    int bb[2][NPRIM][4];
    // capsule silhouettes per camera as 2D swept discs: a (u,v), b-a (du,dv), 1/|b-a|^2, conservative radius^2
    float seg[2][NCAP][6];
};

DEV void project_bbox(const LcrCam &C, int W, int H, const f3 *pts, int npts, float rad, int *bb) {
    int x0 = W, x1 = -1, y0 = H, y1 = -1;
    bool behind = false;
    for (int i = 0; i < npts; i++) {
        f3 d = pts[i] - mk(C.px, C.py, C.pz);
        float xc = dot(d, mk(C.xx, C.xy, C.xz)), yc = dot(d, mk(C.yx, C.yy, C.yz)), zc = -dot(d, mk(C.zx, C.zy, C.zz));
        if (zc < 0.02f) { behind = true; continue; }
        float inv = 1.0f / (zc * C.s);
        float u = 0.5f * W + xc * inv - 0.5f, v = 0.5f * H - yc * inv - 0.5f, rp = rad * inv + 1.5f;
        x0 = min(x0, (int)floorf(u - rp)); x1 = max(x1, (int)ceilf(u + rp));
        y0 = min(y0, (int)floorf(v - rp)); y1 = max(y1, (int)ceilf(v + rp));
    }
    if (behind) { x0 = 0; x1 = W - 1; y0 = 0; y1 = H - 1; }
    bb[0] = x0; bb[1] = x1; bb[2] = y0; bb[3] = y1;
}

DEV void build_scene(const LcrDev &P, int env, Scene &S) {
    const int N = P.n;
    float q[6];
    for (int j = 0; j < 6; j++) q[j] = P.qpos[(size_t)j * N + env];
    ArmFrames F;
    arm_frames(q, F);
    const f3 s0 = local_point(F, 4, lcrm::SPH0x, lcrm::SPH0y, lcrm::SPH0z), s1 = local_point(F, 5, lcrm::SPH1x, lcrm::SPH1y, lcrm::SPH1z);
    // capsules along the kinematic chain (radii: eyeballed link thickness)
    S.ca[0] = mk(0.f, 0.f, 0.f);  S.cb[0] = F.p[0]; S.cr[0] = 0.026f;
    S.ca[1] = F.p[0]; S.cb[1] = F.p[1]; S.cr[1] = 0.022f;
    S.ca[2] = F.p[1]; S.cb[2] = F.p[2]; S.cr[2] = 0.016f;
    S.ca[3] = F.p[2]; S.cb[3] = F.p[3]; S.cr[3] = 0.014f;
    S.ca[4] = F.p[3]; S.cb[4] = F.p[4]; S.cr[4] = 0.013f;
    S.ca[5] = F.p[4]; S.cb[5] = s0;     S.cr[5] = 0.0075f;  // fixed finger
    S.ca[6] = F.p[5]; S.cb[6] = s1;     S.cr[6] = 0.0070f;  // jaw
    const int ncube = P.task == 4 ? 2 : 1;
    int nb = 0;
    for (int c = 0; c < ncube; c++) {
        const float *qp = P.qpos + (size_t)(6 + 7 * c) * N + env;
        float cq[4] = {qp[3 * (size_t)N], qp[4 * (size_t)N], qp[5 * (size_t)N], qp[6 * (size_t)N]};
        CubeRot R = quat_to_cols(cq);
        S.bc[nb] = mk(qp[0], qp[N], qp[2 * (size_t)N]);
        S.bX[nb] = R.X; S.bY[nb] = R.Y; S.bZ[nb] = R.Z;
        S.bh[nb] = mk(0.015f, 0.015f, 0.015f);
        S.bcol[nb] = c == 0 ? mk(0.5f, 0.f, 0.f) : mk(0.f, 0.f, 0.5f);  // reach_cube.xml:26 rgba / stack_two_cubes.xml:34
        S.balpha[nb] = 1.f;
        nb++;
    }
    if (P.has_target) {  // push_cube.xml:35 cylinder r=0.035 h=0.01 / pick_place_cube.xml:35 box 0.015^3, rgba 0 0 1 0.3
        S.bc[nb] = mk(P.target[env], P.target[N + env], P.target[2 * (size_t)N + env]);
        S.bX[nb] = mk(1.f, 0.f, 0.f); S.bY[nb] = mk(0.f, 1.f, 0.f); S.bZ[nb] = mk(0.f, 0.f, 1.f);
        S.bh[nb] = P.task == 2 ? mk(0.035f, 0.035f, 0.01f) : mk(0.015f, 0.015f, 0.015f);
        S.bcol[nb] = mk(0.f, 0.f, 1.f);
        S.balpha[nb] = 0.3f;
        nb++;
    }
    S.nbox = nb;
}

// screen bounding box of ONE primitive (called by one thread per (camera, primitive))
DEV void build_bbox(const LcrCam &C, int W, int H, const Scene &S, int prim, int *bb, float *seg) {
    if (prim < NCAP) {
        f3 pts[2] = {S.ca[prim], S.cb[prim]};
        project_bbox(C, W, H, pts, 2, S.cr[prim], bb);
        // 2D silhouette (conservative): projected end points and the larger projected radius, +20 % for perspective stretch
        float uv[2][2], rp[2];
        bool ok = true;
        for (int i = 0; i < 2; i++) {
            f3 d = pts[i] - mk(C.px, C.py, C.pz);
            float xc = dot(d, mk(C.xx, C.xy, C.xz)), yc = dot(d, mk(C.yx, C.yy, C.yz)), zc = -dot(d, mk(C.zx, C.zy, C.zz));
            if (zc < 0.02f) { ok = false; zc = 0.02f; }
            float inv = 1.0f / (zc * C.s);
            uv[i][0] = 0.5f * W + xc * inv - 0.5f; uv[i][1] = 0.5f * H - yc * inv - 0.5f; rp[i] = S.cr[prim] * inv;
        }
        const float du = uv[1][0] - uv[0][0], dv = uv[1][1] - uv[0][1];
        const float rr = 1.2f * fmaxf(rp[0], rp[1]) + 2.0f;
        seg[0] = uv[0][0]; seg[1] = uv[0][1]; seg[2] = du; seg[3] = dv;
        seg[4] = 1.0f / fmaxf(du * du + dv * dv, 1e-6f);
        seg[5] = ok ? rr * rr : 1e30f;
        return;
    }
    const int k = prim - NCAP;
    if (k >= S.nbox) { bb[0] = W; bb[1] = -1; bb[2] = H; bb[3] = -1; return; }
    f3 pts[8];
    for (int i = 0; i < 8; i++)
        pts[i] = axpy((i & 1) ? S.bh[k].x : -S.bh[k].x, S.bX[k], axpy((i & 2) ? S.bh[k].y : -S.bh[k].y, S.bY[k],
                 axpy((i & 4) ? S.bh[k].z : -S.bh[k].z, S.bZ[k], S.bc[k])));
    project_bbox(C, W, H, pts, 8, 0.f, bb);
}

// rgb in [0,1] -> 0x00BBGGRR with v_cvt_pk_u8_f32 (saturating float->byte conversion and byte insert in one instruction)
DEV unsigned pack_rgb(f3 c) {
    unsigned v = 0u;
    v = __builtin_amdgcn_cvt_pk_u8_f32(c.x * 255.f, 0, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(c.y * 255.f, 1, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(c.z * 255.f, 2, v);
    return v;
}

// background only (no primitive can cover this pixel): checker floor below the horizon, gradient sky above.
// rdu = un-normalised ray direction; the floor normal is +z so the headlight Lambert term is 0.3 - 0.6 rd_z.
DEV unsigned shade_background(f3 ro, f3 rdu) {
    const float inv = rsq(dot(rdu, rdu));
    const float rdz = rdu.z * inv;
    if (rdz < -1e-6f) {
        const float t = -ro.z * rcp(rdu.z);
        const float fx = fmaf(t, rdu.x, ro.x), fy = fmaf(t, rdu.y, ro.y);
        const int cell = ((int)floorf(fx * 10.f) + (int)floorf(fy * 10.f)) & 1;
        const float lam = fminf(fmaf(-0.6f, rdz, 0.3f), 1.f) * 255.f;
        unsigned v = 0u;
        v = __builtin_amdgcn_cvt_pk_u8_f32((cell ? 0.2f : 0.1f) * lam, 0, v);
        v = __builtin_amdgcn_cvt_pk_u8_f32((cell ? 0.3f : 0.2f) * lam, 1, v);
        v = __builtin_amdgcn_cvt_pk_u8_f32((cell ? 0.4f : 0.3f) * lam, 2, v);
        return v;
    }
    const float a = clampf(rdz * 2.f, 0.f, 1.f);
    return pack_rgb(mk(0.15f + a * 0.15f, 0.25f + a * 0.25f, 0.35f + a * 0.35f));
}

// one pixel with primitives: returns linear rgb in [0,1]
DEV f3 shade_pixel(const LcrCam &C, const Scene &S, f3 rdu, unsigned prim_mask) {
    const f3 ro = mk(C.px, C.py, C.pz);
    const f3 rd = rsq(dot(rdu, rdu)) * rdu;
    float tbest = 1e30f;
    f3 nbest = mk(0.f, 0.f, 1.f), col;
    // background: sky gradient above the horizon, checker floor below (builtin checker, 0.1 m squares)
    if (rd.z < -1e-6f) {
        tbest = -ro.z / rd.z;
        const float fx = ro.x + tbest * rd.x, fy = ro.y + tbest * rd.y;
        const int cell = ((int)floorf(fx * 10.f) + (int)floorf(fy * 10.f)) & 1;
        col = cell ? mk(0.2f, 0.3f, 0.4f) : mk(0.1f, 0.2f, 0.3f);
    } else {
        const float a = clampf(rd.z * 2.f, 0.f, 1.f);
        return mk(0.15f + a * 0.15f, 0.25f + a * 0.25f, 0.35f + a * 0.35f);
    }
    // capsules
    for (unsigned m = prim_mask & ((1u << NCAP) - 1u); m; m &= m - 1u) {
        const int k = __builtin_ctz(m);
        const f3 ba = S.cb[k] - S.ca[k], oa = ro - S.ca[k];
        const float r = S.cr[k];
        const float baba = dot(ba, ba), bard = dot(ba, rd), baoa = dot(ba, oa), rdoa = dot(rd, oa), oaoa = dot(oa, oa);
        const float A = baba - bard * bard;
        float B = baba * rdoa - baoa * bard, Cc = baba * oaoa - baoa * baoa - r * r * baba;
        float h = B * B - A * Cc;
        float t = -1.f;
        if (h >= 0.f && A > 1e-12f) {
            t = (-B - sqrtf(h)) / A;
            const float y = baoa + t * bard;
            if (!(y > 0.f && y < baba)) {
                const f3 oc = y <= 0.f ? oa : ro - S.cb[k];
                B = dot(rd, oc); Cc = dot(oc, oc) - r * r; h = B * B - Cc;
                t = h > 0.f ? -B - sqrtf(h) : -1.f;
            }
        }
        if (t > 0.f && t < tbest) {
            tbest = t;
            const f3 pa = axpy(t, rd, ro) - S.ca[k];
            const float hh = clampf(dot(pa, ba) / fmaxf(baba, 1e-12f), 0.f, 1.f);
            nbest = (1.f / r) * (pa - hh * ba);
            col = k >= 5 ? mk(0.75f, 0.75f, 0.75f) : mk(0.8f, 0.8f, 0.8f);
        }
    }
    // boxes (cubes opaque, target marker blended at alpha 0.3 if it is in front of the opaque hit)
    float talpha = 0.f;
    f3 tcol = mk(0.f, 0.f, 0.f);
    for (unsigned m = prim_mask >> NCAP; m; m &= m - 1u) {
        const int k = __builtin_ctz(m);
        if (k >= S.nbox) break;
        const f3 d = ro - S.bc[k];
        const f3 ol = mk(dot(S.bX[k], d), dot(S.bY[k], d), dot(S.bZ[k], d));
        const f3 dl = mk(dot(S.bX[k], rd), dot(S.bY[k], rd), dot(S.bZ[k], rd));
        const f3 inv = mk(1.f / (fabsf(dl.x) > 1e-9f ? dl.x : 1e-9f), 1.f / (fabsf(dl.y) > 1e-9f ? dl.y : 1e-9f), 1.f / (fabsf(dl.z) > 1e-9f ? dl.z : 1e-9f));
        const float tx1 = (-S.bh[k].x - ol.x) * inv.x, tx2 = (S.bh[k].x - ol.x) * inv.x;
        const float ty1 = (-S.bh[k].y - ol.y) * inv.y, ty2 = (S.bh[k].y - ol.y) * inv.y;
        const float tz1 = (-S.bh[k].z - ol.z) * inv.z, tz2 = (S.bh[k].z - ol.z) * inv.z;
        const float tnx = fminf(tx1, tx2), tny = fminf(ty1, ty2), tnz = fminf(tz1, tz2);
        const float tmin = fmaxf(tnx, fmaxf(tny, tnz)), tmax = fminf(fmaxf(tx1, tx2), fminf(fmaxf(ty1, ty2), fmaxf(tz1, tz2)));
        if (tmin <= tmax && tmin > 0.f && tmin < tbest) {
            f3 n = tmin == tnx ? (dl.x > 0.f ? neg(S.bX[k]) : S.bX[k]) : (tmin == tny ? (dl.y > 0.f ? neg(S.bY[k]) : S.bY[k]) : (dl.z > 0.f ? neg(S.bZ[k]) : S.bZ[k]));
            if (S.balpha[k] < 1.f) {
                const float lam = 0.3f + 0.6f * fmaxf(0.f, -dot(n, rd));
                tcol = lam * S.bcol[k]; talpha = S.balpha[k];
            } else { tbest = tmin; nbest = n; col = S.bcol[k]; talpha = 0.f; }
        }
    }
    const float lam = fminf(0.3f + 0.6f * fmaxf(0.f, -dot(nbest, rd)), 1.f);  // ambient + headlight (reach_cube.xml:8)
    f3 out = lam * col;
    if (talpha > 0.f) out = axpy(talpha, tcol, (1.f - talpha) * out);
    return out;
}

__global__ __launch_bounds__(256) void lcr_render_obs_kernel(LcrDev P, LcrCam front, LcrCam top) {
    // A wave renders one image ROW at a time (320 pixels = 5 per lane, pixel = lane + 64 g): the 960 bytes of the row are
    // staged in LDS and leave as 60 contiguous 16-B non-temporal stores (one store instruction per row).  Primitive
    // culling is wave-uniform: a row / 64-pixel group only ray-tests primitives whose screen bounding box overlaps it.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ Scene S;
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][960];
    const int env = blockIdx.x;
    const int W = 320, H = 240;
    if (threadIdx.x == 0) build_scene(P, env, S);
    __syncthreads();
    if (threadIdx.x < 2 * NPRIM) {
        const int cam_id = threadIdx.x / NPRIM, prim = threadIdx.x - cam_id * NPRIM;
        build_bbox(cam_id ? top : front, W, H, S, prim, S.bb[cam_id][prim], prim < NCAP ? S.seg[cam_id][prim] : nullptr);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const size_t img_bytes = (size_t)H * W * 3;
    unsigned char *st = stage[wave];
    // Culling data lives in registers, one primitive per lane (lane k <-> primitive k): the per-row / per-span primitive
    // masks are then single ballots instead of LDS-latency-bound scalar loops.
    const int pk = lane < NPRIM ? lane : 0, ck = lane < NCAP ? lane : 0;
    int bx0[2], bx1[2], by0[2], by1[2];
    float sgp[2][6];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        bx0[c] = S.bb[c][pk][0]; bx1[c] = S.bb[c][pk][1]; by0[c] = S.bb[c][pk][2]; by1[c] = S.bb[c][pk][3];
#pragma unroll
        for (int i = 0; i < 6; i++) sgp[c][i] = S.seg[c][ck][i];
    }
    for (int task = blockIdx.y * 4 + wave; task < 2 * H; task += gridDim.y * 4) {
        const bool is_top = task >= H;
        const int row = is_top ? task - H : task;
        const LcrCam &C = is_top ? top : front;
        const int x0 = is_top ? bx0[1] : bx0[0], x1 = is_top ? bx1[1] : bx1[0], y0 = is_top ? by0[1] : by0[0], y1 = is_top ? by1[1] : by1[0];
        const bool in_row = lane < NPRIM && row >= y0 && row <= y1;
        // per-row constants of the ray: rd(px) = rbase + X * sx(px), camera looks along -Z
        const float sy = -(row + 0.5f - 0.5f * H) * C.s;
        const f3 ro = mk(C.px, C.py, C.pz);
        const f3 rbase = mk(C.yx * sy - C.zx, C.yy * sy - C.zy, C.yz * sy - C.zz);
#pragma unroll
        for (int g = 0; g < 5; g++) {
            unsigned m = (unsigned)__ballot(in_row && x1 >= 64 * g && x0 <= 64 * g + 63);
            const int px = lane + 64 * g;
            // refine: keep a capsule only if some pixel of this 64-pixel span lies inside its 2D silhouette
            for (unsigned r = m & ((1u << NCAP) - 1u); r; r &= r - 1u) {
                const int k = __builtin_ctz(r);
                float sg[6];
#pragma unroll
                for (int i = 0; i < 6; i++)
                    sg[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(is_top ? sgp[1][i] : sgp[0][i]), k));
                const float pu = (float)px - sg[0], pv = (float)row - sg[1];
                const float t = clampf((pu * sg[2] + pv * sg[3]) * sg[4], 0.f, 1.f);
                const float eu = pu - t * sg[2], ev = pv - t * sg[3];
                if (!__any(eu * eu + ev * ev <= sg[5])) m &= ~(1u << k);
            }
            const float sx = (px + 0.5f - 0.5f * W) * C.s;
            const f3 rdu = axpy(sx, mk(C.xx, C.xy, C.xz), rbase);
            const unsigned rgb = m ? pack_rgb(shade_pixel(C, S, rdu, m)) : shade_background(ro, rdu);
            st[3 * px + 0] = (unsigned char)rgb;
            st[3 * px + 1] = (unsigned char)(rgb >> 8);
            st[3 * px + 2] = (unsigned char)(rgb >> 16);
        }
        if (lane < 60) {
            const u32x4 v = *reinterpret_cast<const u32x4 *>(st + 16 * lane);
            u32x4 *dst = reinterpret_cast<u32x4 *>((is_top ? P.img_top : P.img_front) + (size_t)env * img_bytes) + (size_t)row * 60 + lane;
            __builtin_nontemporal_store(v, dst);
        }
    }
}

// one env, arbitrary camera / resolution (render(), 640x640 camera_vizu): one thread per pixel, no culling
__global__ __launch_bounds__(256) void lcr_render_single_kernel(LcrDev P, LcrCam cam, int env, int W, int H, unsigned char *out) {
    __shared__ Scene S;
    if (threadIdx.x == 0) build_scene(P, env, S);
    __syncthreads();
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= W * H) return;
    const int v = pix / W, u = pix - v * W;
    const float sx = (u + 0.5f - 0.5f * W) * cam.s, sy = -(v + 0.5f - 0.5f * H) * cam.s;
    const f3 rdu = mk(cam.xx * sx + cam.yx * sy - cam.zx, cam.xy * sx + cam.yy * sy - cam.zy, cam.xz * sx + cam.yz * sy - cam.zz);
    const unsigned rgb = pack_rgb(shade_pixel(cam, S, rdu, (1u << NPRIM) - 1u));
    out[3 * (size_t)pix + 0] = (unsigned char)rgb;
    out[3 * (size_t)pix + 1] = (unsigned char)(rgb >> 8);
    out[3 * (size_t)pix + 2] = (unsigned char)(rgb >> 16);
}

}  // namespace

int lcr_launch_render_obs(const LcrDev &P, const LcrCam &front, const LcrCam &top, void *stream) {
    if (!P.img_front || !P.img_top) return 0;
    hipLaunchKernelGGL(lcr_render_obs_kernel, dim3(P.n, 2), dim3(256), 0, (hipStream_t)stream, P, front, top);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int lcr_launch_render_single(const LcrDev &P, const LcrCam &cam, int env, int W, int H, unsigned char *out_dev, void *stream) {
    hipLaunchKernelGGL(lcr_render_single_kernel, dim3((W * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, cam, env, W, H, out_dev);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
