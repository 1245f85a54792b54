// lcr_render.hip -- image observations: a small ray-caster for the two 240x320 observation cameras of every env
// (get_observation, envs/reach_cube_env.py:288-292: renderer.update_scene(camera="camera_front"/"camera_top"); render())
// and for the 640x640 `camera_vizu` frame of render() (envs/reach_cube_env.py:350-355).
//
// It is an APPROXIMATE restatement, not MuJoCo's OpenGL renderer (which cannot run here): pinhole cameras with the
// poses of the scene xmls (reach_cube.xml:29-31) and MuJoCo's default fovy 45 deg; checker floor (texrepeat 5 -> 0.1 m
// squares, reach_cube.xml:14-16), gradient sky, the cube(s) as exact oriented boxes, the arm as 7 capsules between the
// link origins (the 20 STL meshes are not shipped), ambient 0.3 + headlight 0.6 Lambert shading, no shadows.
//
// Mapping: a workgroup owns one env's 480 image rows (both frames); a wave handles one row at a time and writes its 960
// bytes with ONE non-temporal 16-B store instruction (60 lanes, 960 contiguous bytes).  The background (floor + sky) does
// not depend on the env: it is rendered once into a cached frame pair and rows no primitive touches are plain copies of
// it.  The per-env scene (FK of the arm, cube frames, screen-space bounding boxes of every primitive for both cameras)
// is built once per workgroup in LDS; culling is wave-uniform (per row and 64-pixel span), only spans that contain a
// primitive are ray-cast.
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
    // capsule silhouettes per camera as 2D swept discs: a (u,v), b-a (du,dv), 1/|b-a|^2, conservative radius
    float seg[2][NCAP][6];
    // per camera ray-test constants (the ray origin is fixed per camera):
    // capsule: ba(3) baba | oa(3) baoa | ob(3) K=baba*oaoa-baoa^2-r^2*baba | oaoa-r^2, obob-r^2, 1/r, 1/baba
    float capc[2][NCAP][16];
    f3 box_ol[2][NBOX];   // box-frame coordinates of the camera position
};

DEV void project_bbox(const LcrCam &C, int W, int H, const f3 *pts, int npts, float rad, int *bb) {
    int x0 = W, x1 = -1, y0 = H, y1 = -1;
    bool behind = false;
    for (int i = 0; i < npts; i++) {
        f3 d = pts[i] - mk(C.px, C.py, C.pz);
        float xc = dot(d, mk(C.xx, C.xy, C.xz)), yc = dot(d, mk(C.yx, C.yy, C.yz)), zc = -dot(d, mk(C.zx, C.zy, C.zz));
        if (zc < 0.02f) { behind = true; continue; }
        float inv = 1.0f / (zc * C.s);
        // a sphere of radius rad projects to an ellipse stretched radially by <= 1 + tan^2(off-axis angle)
        const float stretch = 1.0f + (xc * xc + yc * yc) / (zc * zc);
        float u = 0.5f * W + xc * inv - 0.5f, v = 0.5f * H - yc * inv - 0.5f, rp = rad * inv * stretch + 1.5f;
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
DEV void build_bbox(const LcrCam &C, int W, int H, const Scene &S, int prim, int *bb, float *seg, float *cc, f3 *ol) {
    const f3 ro = mk(C.px, C.py, C.pz);
    if (prim < NCAP) {
        f3 pts[2] = {S.ca[prim], S.cb[prim]};
        project_bbox(C, W, H, pts, 2, S.cr[prim], bb);
        {
            const f3 ba = S.cb[prim] - S.ca[prim], oa = ro - S.ca[prim], ob = ro - S.cb[prim];
            const float r = S.cr[prim], baba = dot(ba, ba), baoa = dot(ba, oa), oaoa = dot(oa, oa);
            cc[0] = ba.x; cc[1] = ba.y; cc[2] = ba.z; cc[3] = baba;
            cc[4] = oa.x; cc[5] = oa.y; cc[6] = oa.z; cc[7] = baoa;
            cc[8] = ob.x; cc[9] = ob.y; cc[10] = ob.z; cc[11] = baba * oaoa - baoa * baoa - r * r * baba;
            cc[12] = oaoa - r * r; cc[13] = dot(ob, ob) - r * r; cc[14] = 1.0f / r; cc[15] = 1.0f / fmaxf(baba, 1e-12f);
        }
        // 2D silhouette (conservative): projected end points and the larger projected (perspective-stretched) radius
        float uv[2][2], rp[2];
        bool ok = true;
        for (int i = 0; i < 2; i++) {
            f3 d = pts[i] - mk(C.px, C.py, C.pz);
            float xc = dot(d, mk(C.xx, C.xy, C.xz)), yc = dot(d, mk(C.yx, C.yy, C.yz)), zc = -dot(d, mk(C.zx, C.zy, C.zz));
            if (zc < 0.02f) { ok = false; zc = 0.02f; }
            float inv = 1.0f / (zc * C.s);
            uv[i][0] = 0.5f * W + xc * inv - 0.5f; uv[i][1] = 0.5f * H - yc * inv - 0.5f;
            rp[i] = S.cr[prim] * inv * (1.0f + (xc * xc + yc * yc) / (zc * zc));   // perspective stretch of the silhouette
        }
        const float du = uv[1][0] - uv[0][0], dv = uv[1][1] - uv[0][1];
        const float rr = fmaxf(rp[0], rp[1]) + 1.5f;
        seg[0] = uv[0][0]; seg[1] = uv[0][1]; seg[2] = du; seg[3] = dv;
        seg[4] = 1.0f / fmaxf(du * du + dv * dv, 1e-6f);
        seg[5] = ok ? rr : 1e15f;
        return;
    }
    const int k = prim - NCAP;
    if (k >= S.nbox) { bb[0] = W; bb[1] = -1; bb[2] = H; bb[3] = -1; return; }
    {
        const f3 d = ro - S.bc[k];
        *ol = mk(dot(S.bX[k], d), dot(S.bY[k], d), dot(S.bZ[k], d));
    }
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
    bool sky = false;
    // background: sky gradient above the horizon, checker floor below (builtin checker, 0.1 m squares)
    if (rd.z < -1e-6f) {
        tbest = -ro.z / rd.z;
        const float fx = ro.x + tbest * rd.x, fy = ro.y + tbest * rd.y;
        const int cell = ((int)floorf(fx * 10.f) + (int)floorf(fy * 10.f)) & 1;
        col = cell ? mk(0.2f, 0.3f, 0.4f) : mk(0.1f, 0.2f, 0.3f);
    } else {
        const float a = clampf(rd.z * 2.f, 0.f, 1.f);
        col = mk(0.15f + a * 0.15f, 0.25f + a * 0.25f, 0.35f + a * 0.35f);
        sky = true;   // drawn unshaded unless a primitive is hit
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
            tbest = t; sky = false;
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
            } else { tbest = tmin; nbest = n; col = S.bcol[k]; talpha = 0.f; sky = false; }
        }
    }
    const float lam = sky ? 1.f : fminf(0.3f + 0.6f * fmaxf(0.f, -dot(nbest, rd)), 1.f);  // ambient + headlight (reach_cube.xml:8)
    f3 out = lam * col;
    if (talpha > 0.f) out = axpy(talpha, tcol, (1.f - talpha) * out);
    return out;
}

// One 64-pixel span of an observation row: ray-cast the primitives of mask `m` (wave-uniform) along the UN-normalised ray
// ro + t d.  The staged row already holds the background, so a lane only reports a colour when its ray hits something.
// Capsule / box constants that depend on the camera position only come precomputed from the Scene; the shading normal is
// evaluated once, for the nearest hit, after the depth loop.  `stpx` = this pixel's 3 staged background bytes (read for
// the translucent target marker only).  Returns true and sets rgb when the pixel has to be rewritten.
DEV bool shade_span(const Scene &S, int cam, f3 ro, f3 d, unsigned m, const unsigned char *stpx, unsigned &rgb) {
    const float dd = dot(d, d);
    float tbest = d.z < -1e-6f ? -ro.z * rcp(d.z) : 1e30f;   // the floor hides what lies below it
    int kbest = -1;
    f3 nbest = mk(0.f, 0.f, 1.f);
    for (unsigned mm = m & ((1u << NCAP) - 1u); mm; mm &= mm - 1u) {
        const int k = __builtin_ctz(mm);
        const float *c = S.capc[cam][k];
        const f3 ba = mk(c[0], c[1], c[2]), oa = mk(c[4], c[5], c[6]);
        const float baba = c[3], baoa = c[7];
        const float bard = dot(ba, d), rdoa = dot(oa, d);
        const float A = baba * dd - bard * bard, B = baba * rdoa - baoa * bard;
        const float h = B * B - A * c[11];
        const bool hc = h >= 0.f && A > 1e-12f;
        if (!__any(hc)) continue;
        float t = (-B - sqrtf(fmaxf(h, 0.f))) * rcp(A);
        const float y = baoa + t * bard;
        const bool body = y > 0.f && y < baba;
        if (__any(hc && !body)) {   // end caps
            const bool lo = y <= 0.f;
            const f3 oc = mk(lo ? oa.x : c[8], lo ? oa.y : c[9], lo ? oa.z : c[10]);
            const float Bc = dot(d, oc), h2 = Bc * Bc - dd * (lo ? c[12] : c[13]);
            const float tc = h2 > 0.f ? (-Bc - sqrtf(fmaxf(h2, 0.f))) * rcp(dd) : -1.f;
            t = body ? t : tc;
        }
        if (hc && t > 0.f && t < tbest) { tbest = t; kbest = k; }
    }
    float talpha = 0.f;
    f3 tcol = mk(0.f, 0.f, 0.f);
    const float inv_len = rsq(dd);
    for (unsigned mm = m >> NCAP; mm; mm &= mm - 1u) {
        const int k = __builtin_ctz(mm);
        if (k >= S.nbox) break;
        const f3 ol = S.box_ol[cam][k];
        const f3 dl = mk(dot(S.bX[k], d), dot(S.bY[k], d), dot(S.bZ[k], d));
        const f3 inv = mk(rcp(fabsf(dl.x) > 1e-9f ? dl.x : 1e-9f), rcp(fabsf(dl.y) > 1e-9f ? dl.y : 1e-9f), rcp(fabsf(dl.z) > 1e-9f ? dl.z : 1e-9f));
        const float tx1 = (-S.bh[k].x - ol.x) * inv.x, tx2 = (S.bh[k].x - ol.x) * inv.x;
        const float ty1 = (-S.bh[k].y - ol.y) * inv.y, ty2 = (S.bh[k].y - ol.y) * inv.y;
        const float tz1 = (-S.bh[k].z - ol.z) * inv.z, tz2 = (S.bh[k].z - ol.z) * inv.z;
        const float tnx = fminf(tx1, tx2), tny = fminf(ty1, ty2), tnz = fminf(tz1, tz2);
        const float tmin = fmaxf(tnx, fmaxf(tny, tnz)), tmax = fminf(fmaxf(tx1, tx2), fminf(fmaxf(ty1, ty2), fmaxf(tz1, tz2)));
        if (tmin <= tmax && tmin > 0.f && tmin < tbest) {
            const f3 n = tmin == tnx ? (dl.x > 0.f ? neg(S.bX[k]) : S.bX[k]) : (tmin == tny ? (dl.y > 0.f ? neg(S.bY[k]) : S.bY[k]) : (dl.z > 0.f ? neg(S.bZ[k]) : S.bZ[k]));
            if (S.balpha[k] < 1.f) {
                const float lam = 0.3f + 0.6f * fmaxf(0.f, -dot(n, d) * inv_len);
                tcol = lam * S.bcol[k]; talpha = S.balpha[k];
            } else { tbest = tmin; nbest = n; kbest = NCAP + k; talpha = 0.f; }
        }
    }
    if (!__any(kbest >= 0 || talpha > 0.f)) return false;
    f3 col = mk(0.f, 0.f, 0.f);
    if (kbest >= 0 && kbest < NCAP) {   // capsule normal of the winning primitive (lane-varying index)
        const float *c = S.capc[cam][kbest];
        const f3 ba = mk(c[0], c[1], c[2]);
        const f3 pa = axpy(tbest, d, mk(c[4], c[5], c[6]));
        const float hh = clampf(dot(pa, ba) * c[15], 0.f, 1.f);
        nbest = c[14] * (pa - hh * ba);
        col = kbest >= 5 ? mk(0.75f, 0.75f, 0.75f) : mk(0.8f, 0.8f, 0.8f);
    } else if (kbest >= NCAP) {
        col = S.bcol[kbest - NCAP];
    }
    const float lam = fminf(0.3f + 0.6f * fmaxf(0.f, -dot(nbest, d) * inv_len), 1.f);  // ambient + headlight (reach_cube.xml:8)
    f3 out = lam * col;
    if (kbest < 0) out = mk(stpx[0] * (1.f / 255.f), stpx[1] * (1.f / 255.f), stpx[2] * (1.f / 255.f));
    if (talpha > 0.f) out = axpy(talpha, tcol, (1.f - talpha) * out);
    rgb = pack_rgb(out);
    return kbest >= 0 || talpha > 0.f;
}

// background frames (checker floor + sky) of the two observation cameras: identical for every env and every step, so they
// are rendered ONCE at lcr_create into P.img_bg ([2][240][320][3], 460 800 B, L2-resident) and copied row-wise afterwards.
__global__ __launch_bounds__(256) void lcr_render_bg_kernel(LcrDev P, LcrCam front, LcrCam top) {
    const int W = 320, H = 240;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= 2 * W * H) return;
    const bool is_top = pix >= W * H;
    const int p = is_top ? pix - W * H : pix;
    const int row = p / W, px = p - row * W;
    const LcrCam &C = is_top ? top : front;
    const float sy = -(row + 0.5f - 0.5f * H) * C.s, sx = (px + 0.5f - 0.5f * W) * C.s;
    const f3 rbase = mk(C.yx * sy - C.zx, C.yy * sy - C.zy, C.yz * sy - C.zz);
    const unsigned rgb = shade_background(mk(C.px, C.py, C.pz), axpy(sx, mk(C.xx, C.xy, C.xz), rbase));
    P.img_bg[3 * (size_t)pix + 0] = (unsigned char)rgb;
    P.img_bg[3 * (size_t)pix + 1] = (unsigned char)(rgb >> 8);
    P.img_bg[3 * (size_t)pix + 2] = (unsigned char)(rgb >> 16);
}

__global__ __launch_bounds__(256) void lcr_render_obs_kernel(LcrDev P, LcrCam front, LcrCam top) {
    // A workgroup owns one env (480 rows: front frame then top frame); a wave handles one ROW at a time (960 B = 60 lanes x
    // 16 B).  The row starts as a copy of the cached background row (L2 hit); if no primitive's screen bounding box touches
    // the row (wave-uniform ballot) it leaves straight away as one non-temporal 16-B store per lane.  Otherwise the row is
    // staged in LDS, the 64-pixel spans that contain primitives are ray-cast (pixel = lane + 64 g) and overwrite their
    // bytes, and the row is stored from LDS.  The next row's background is prefetched while the current one is processed.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    __shared__ Scene S;
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][4 * 960];
    const int env = blockIdx.x;
    const int W = 320, H = 240;
    if (threadIdx.x == 0) build_scene(P, env, S);
    __syncthreads();
    if (threadIdx.x < 2 * NPRIM) {
        const int cam_id = threadIdx.x / NPRIM, prim = threadIdx.x - cam_id * NPRIM;
        build_bbox(cam_id ? top : front, W, H, S, prim, S.bb[cam_id][prim], prim < NCAP ? S.seg[cam_id][prim] : nullptr,
                   prim < NCAP ? S.capc[cam_id][prim] : nullptr, prim < NCAP ? nullptr : &S.box_ol[cam_id][prim - NCAP]);
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t img_bytes = (size_t)H * W * 3;
    unsigned char *st = stage[wave];
    // waves 0,1 render camera_front (even / odd bands), waves 2,3 camera_top: everything camera-dependent is wave-uniform
    const int cam = wave >> 1;
    const LcrCam &C = cam ? top : front;
    const f3 ro = mk(C.px, C.py, C.pz), CX = mk(C.xx, C.xy, C.xz), CY = mk(C.yx, C.yy, C.yz), CZ = mk(C.zx, C.zy, C.zz);
    // Culling data lives in registers, one primitive per lane (lane k <-> primitive k): per band, every lane computes the
    // pixel interval [xa, xb] its primitive can cover on the band's 4 rows (capsules: 2D swept-disc silhouette, boxes:
    // bounding box) -> a 20-bit mask of the 16-pixel tile columns it touches.
    const int pk = lane < NPRIM ? lane : 0, ck = lane < NCAP ? lane : 0;
    const int by0 = S.bb[cam][pk][2], by1 = S.bb[cam][pk][3];
    const float fbx0 = (float)S.bb[cam][pk][0], fbx1 = (float)S.bb[cam][pk][1];
    const float au = S.seg[cam][ck][0], av = S.seg[cam][ck][1], du = S.seg[cam][ck][2], dv = S.seg[cam][ck][3];
    const float R = S.seg[cam][ck][5];
    const bool flat = fabsf(dv) < 1e-4f;
    const float inv_dv = flat ? 0.f : 1.0f / dv;
    const u32x4 *bg = reinterpret_cast<const u32x4 *>(P.img_bg) + (size_t)cam * H * 60;
    u32x4 *out = reinterpret_cast<u32x4 *>((cam ? P.img_top : P.img_front) + (size_t)env * img_bytes);
    const bool l3 = lane < 48;               // a band = 240 vectors = 3 full wave loads + 48 lanes
    const int lq = l3 ? lane : 47;
    // co-resident workgroups start at different bands (hashed phase) so that their ray-cast (VALU-bound) and copy
    // (memory-bound) stretches overlap instead of all waves of a SIMD hitting the arm's rows together
    constexpr int NB = H / 4;
    const int rot = (int)((blockIdx.x * 0x9E3779B1u) >> 29) * 8;
    const int tx = lane & 15, ty = lane >> 4;   // pixel of this lane inside a 16 x 4 tile
    auto band_of = [&](int it) { int b = it + rot; return b >= NB ? b - NB : b; };
    auto load_band = [&](int b, u32x4 &a0, u32x4 &a1, u32x4 &a2, u32x4 &a3) {
        const u32x4 *src = bg + b * 240;
        a0 = src[lane]; a1 = src[64 + lane]; a2 = src[128 + lane]; a3 = src[192 + lq];
    };
    // one band: v0..v3 hold its background (240 16-B vectors)
    auto do_band = [&](int b, const u32x4 &v0, const u32x4 &v1, const u32x4 &v2, const u32x4 &v3) {
        u32x4 *dst = out + b * 240;
        const int row0 = 4 * b;
        const bool in_band = lane < NPRIM && row0 + 3 >= by0 && row0 <= by1;
        unsigned tm = 0u;
        if (in_band) {   // tile columns this lane's primitive can touch on rows row0 .. row0+3
            const float f0 = (float)row0 - R - av, f1 = (float)(row0 + 3) + R - av;
            const float sA = f0 * inv_dv, sB = f1 * inv_dv;
            const float s0 = flat ? 0.f : clampf(fminf(sA, sB), 0.f, 1.f), s1 = flat ? 1.f : clampf(fmaxf(sA, sB), 0.f, 1.f);
            const float e0 = s0 * du, e1 = s1 * du;
            const float xa = lane < NCAP ? au + fminf(e0, e1) - R : fbx0;
            const float xb = lane < NCAP ? au + fmaxf(e0, e1) + R : fbx1;
            if (xb >= 0.f && xa <= (float)(W - 1)) {
                const int ta = (int)fmaxf(xa * (1.f / 16.f), 0.f), tb = (int)fminf(xb * (1.f / 16.f), (float)(W / 16 - 1));
                tm = (2u << tb) - (1u << ta);
            }
        }
        unsigned U = 0u;
        if (__any(tm != 0u)) {
#pragma unroll
            for (int k = 0; k < NPRIM; k++) U |= (unsigned)__builtin_amdgcn_readlane((int)tm, k);
        }
        if (U == 0u) {
            __builtin_nontemporal_store(v0, dst + lane);
            __builtin_nontemporal_store(v1, dst + 64 + lane);
            __builtin_nontemporal_store(v2, dst + 128 + lane);
            if (l3) __builtin_nontemporal_store(v3, dst + 192 + lane);
            return;
        }
        u32x4 *sv = reinterpret_cast<u32x4 *>(st);
        sv[lane] = v0; sv[64 + lane] = v1; sv[128 + lane] = v2;
        if (l3) sv[192 + lane] = v3;
        // rays of this lane's tile row: rd(px) = rbase + X * sx(px), camera looks along -Z
        const float sy = -((float)(row0 + ty) + 0.5f - 0.5f * H) * C.s;
        const f3 rbase = axpy(sy, CY, neg(CZ));
        for (; U; U &= U - 1u) {
            const int t = __builtin_ctz(U);
            const unsigned m = (unsigned)__ballot((tm >> t) & 1u);
            const int px = 16 * t + tx;
            const float sx = ((float)px + 0.5f - 0.5f * W) * C.s;
            const f3 rdu = axpy(sx, CX, rbase);
            unsigned char *stpx = st + ty * 960 + 3 * px;
            unsigned rgb = 0u;
            if (shade_span(S, cam, ro, rdu, m, stpx, rgb)) {
                stpx[0] = (unsigned char)rgb;
                stpx[1] = (unsigned char)(rgb >> 8);
                stpx[2] = (unsigned char)(rgb >> 16);
            }
        }
        __builtin_nontemporal_store(sv[lane], dst + lane);
        __builtin_nontemporal_store(sv[64 + lane], dst + 64 + lane);
        __builtin_nontemporal_store(sv[128 + lane], dst + 128 + lane);
        if (l3) __builtin_nontemporal_store(sv[192 + lane], dst + 192 + lane);
    };
    // two register sets in ping-pong: the next band's background is in flight while the current band is processed
    const int par = wave & 1;
    u32x4 a0, a1, a2, a3, b0, b1, b2, b3;
    load_band(band_of(par), a0, a1, a2, a3);
    for (int it = par; it < NB; it += 4) {          // NB/2 = 30 bands per wave, 15 pairs
        load_band(band_of(it + 2), b0, b1, b2, b3);
        do_band(band_of(it), a0, a1, a2, a3);
        if (it + 4 < NB) load_band(band_of(it + 4), a0, a1, a2, a3);
        do_band(band_of(it + 2), b0, b1, b2, b3);
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

// terminal poses of the listed envs (the step kernel has already reset them: term_obs / term_quat hold the last pose of the episode) gathered into a
// compact [nq][count] qpos + [3][count] target block, so that lcr_render_obs_kernel can draw their last frames as one batch (lcr_render_terminal)
__global__ __launch_bounds__(256) void lcr_gather_terminal_kernel(LcrDev P, const int *ids, int count, float *qpos_out, float *target_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const size_t N = (size_t)P.n, C = (size_t)count;
    const int e = ids[i];
    const float *t = P.term_obs + e, *tq = P.term_quat + e;   // term_obs rows: arm_qpos 0-5, arm_qvel 6-11, cube 12-14, aux 15-17
    for (int j = 0; j < 6; j++) qpos_out[j * C + i] = t[j * N];
    for (int j = 0; j < 3; j++) qpos_out[(6 + j) * C + i] = t[(12 + j) * N];
    for (int j = 0; j < 4; j++) qpos_out[(9 + j) * C + i] = tq[j * N];
    if (P.task == 4) {
        for (int j = 0; j < 3; j++) qpos_out[(13 + j) * C + i] = t[(15 + j) * N];
        for (int j = 0; j < 4; j++) qpos_out[(16 + j) * C + i] = tq[(4 + j) * N];
    }
    for (int j = 0; j < 3; j++) target_out[j * C + i] = P.has_target ? t[(15 + j) * N] : 0.f;
}

}  // namespace

int lcr_launch_gather_terminal(const LcrDev &P, const int *ids_dev, int count, float *qpos_out, float *target_out, void *stream) {
    hipLaunchKernelGGL(lcr_gather_terminal_kernel, dim3((count + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, ids_dev, count, qpos_out, target_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int lcr_launch_render_obs(const LcrDev &P, const LcrCam &front, const LcrCam &top, void *stream) {
    if (!P.img_front || !P.img_top) return 0;
    hipLaunchKernelGGL(lcr_render_obs_kernel, dim3(P.n), dim3(256), 0, (hipStream_t)stream, P, front, top);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int lcr_launch_render_bg(const LcrDev &P, const LcrCam &front, const LcrCam &top, void *stream) {
    if (!P.img_bg) return 0;
    hipLaunchKernelGGL(lcr_render_bg_kernel, dim3((2 * 320 * 240 + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, front, top);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

int lcr_launch_render_single(const LcrDev &P, const LcrCam &cam, int env, int W, int H, unsigned char *out_dev, void *stream) {
    hipLaunchKernelGGL(lcr_render_single_kernel, dim3((W * H + 255) / 256), dim3(256), 0, (hipStream_t)stream, P, cam, env, W, H, out_dev);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
