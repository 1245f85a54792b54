// lcr_render.hip -- image observations: a small ray-caster for the two 240x320 observation cameras of every env
// (get_observation, envs/reach_cube_env.py:288-292: renderer.update_scene(camera="camera_front"/"camera_top"); render())
// and for the 640x640 `camera_vizu` frame of render() (envs/reach_cube_env.py:350-355).
//
// It is an APPROXIMATE restatement, not MuJoCo's OpenGL renderer (which cannot run here): pinhole cameras with the
// poses of the scene xmls (reach_cube.xml:29-31) and MuJoCo's default fovy 45 deg; checker floor (texrepeat 5 -> 0.1 m
// squares, reach_cube.xml:14-16), gradient sky, the cube(s) as exact oriented boxes, the arm -- round 5 -- as the bounding boxes of its seven collision
// hulls (base_link, link_1 .. link_6: mesh extents of the golden model file, follower.xml:54-97; the 20 STL meshes themselves are not shipped; rounds 1-4
// drew capsules between the link origins), ambient 0.3 + headlight 0.6 Lambert shading, no shadows.
//
// Mapping: a workgroup owns one env's 480 image rows (both frames); a wave handles one BAND of 4 rows at a time and writes its 3 840
// bytes with non-temporal 16-B stores.  What does not depend on the env -- floor, sky and the arm's base -- is rendered once into a
// cached frame pair, and bands no other primitive touches are plain copies of it.  The per-env scene (FK of the arm, cube frames, per camera the
// ray-test constants and the screen-space silhouette of every primitive) is built once per workgroup in LDS; culling is wave-uniform (per band and
// 16-pixel column), only 16 x 4 tiles that a silhouette touches are ray-cast.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "lcr_arm.h"
#include "lcr_device.h"

using namespace lcrdev;

namespace {

constexpr int NARM = 7;    // base_link, link_1 .. link_6 as the bounding boxes of their collision hulls (lcr_model_gen.h ARMB*: model_golden.json "mesh_aabb")
constexpr int NBOX = NARM + 3;      // the arm boxes, then cube, second cube (Stack), target marker (Push / PickPlace)
constexpr int NPRIM = NBOX;
constexpr int BASE = 0;    // the base box is the same in every env: part of the cached background, ray-cast only where another primitive may hide it / hide behind it

struct Scene {
    f3 bc[NBOX], bX[NBOX], bY[NBOX], bZ[NBOX], bh[NBOX];
    f3 bcol[NBOX];
    float balpha[NBOX];
    int nbox;
    int marker;   // index of the translucent target marker among the boxes, -1 if the task has none
    // per camera: the ray direction in the box frame is affine in the pixel, dl = C0 + sy B + sx A:  A(3) ol.x | B(3) ol.y | C0(3) ol.z | half(3) alpha
    // (ol = box-frame coordinates of the camera position)
    __attribute__((aligned(16))) float boxc[2][NBOX][16];
    // culling record per camera and primitive, read back by the primitive's lane for every band:
    // y0 y1 x0 x1 of the screen bounding box | stadium au av du R | du/dv 1/dv strip half-width, flat flag
    __attribute__((aligned(16))) float cull[2][NPRIM][12];
};

DEV void arm_box(int i, f3 &c, f3 &h) {
    const float bc[NARM][3] = {{lcrm::ARMB0cx, lcrm::ARMB0cy, lcrm::ARMB0cz}, {lcrm::ARMB1cx, lcrm::ARMB1cy, lcrm::ARMB1cz}, {lcrm::ARMB2cx, lcrm::ARMB2cy, lcrm::ARMB2cz},
                               {lcrm::ARMB3cx, lcrm::ARMB3cy, lcrm::ARMB3cz}, {lcrm::ARMB4cx, lcrm::ARMB4cy, lcrm::ARMB4cz}, {lcrm::ARMB5cx, lcrm::ARMB5cy, lcrm::ARMB5cz},
                               {lcrm::ARMB6cx, lcrm::ARMB6cy, lcrm::ARMB6cz}};
    const float bh[NARM][3] = {{lcrm::ARMB0hx, lcrm::ARMB0hy, lcrm::ARMB0hz}, {lcrm::ARMB1hx, lcrm::ARMB1hy, lcrm::ARMB1hz}, {lcrm::ARMB2hx, lcrm::ARMB2hy, lcrm::ARMB2hz},
                               {lcrm::ARMB3hx, lcrm::ARMB3hy, lcrm::ARMB3hz}, {lcrm::ARMB4hx, lcrm::ARMB4hy, lcrm::ARMB4hz}, {lcrm::ARMB5hx, lcrm::ARMB5hy, lcrm::ARMB5hz},
                               {lcrm::ARMB6hx, lcrm::ARMB6hy, lcrm::ARMB6hz}};
    c = mk(bc[i][0], bc[i][1], bc[i][2]); h = mk(bh[i][0], bh[i][1], bh[i][2]);
}

// the base box (body frame of base_link: Rz(-90 deg) at the origin, follower.xml:51)
DEV void base_box(f3 &c, f3 &X, f3 &Y, f3 &Z, f3 &h) {
    X = mk(0.f, -1.f, 0.f); Y = mk(1.f, 0.f, 0.f); Z = mk(0.f, 0.f, 1.f);
    f3 lc;
    arm_box(0, lc, h);
    c = axpy(lc.x, X, axpy(lc.y, Y, lc.z * Z));
}

DEV void build_scene(const LcrDev &P, int env, Scene &S) {
    const int N = P.n;
    float q[6];
    for (int j = 0; j < 6; j++) q[j] = P.qpos[(size_t)j * N + env];
    ArmFrames F;
    arm_frames(q, F);
    base_box(S.bc[0], S.bX[0], S.bY[0], S.bZ[0], S.bh[0]);
    S.bcol[0] = mk(0.8f, 0.8f, 0.8f); S.balpha[0] = 1.f;
    for (int i = 1; i < NARM; i++) {   // link_i: frame i - 1 of the chain
        f3 lc;
        arm_box(i, lc, S.bh[i]);
        S.bc[i] = local_point(F, i - 1, lc.x, lc.y, lc.z);
        S.bX[i] = F.X[i - 1]; S.bY[i] = F.Y[i - 1]; S.bZ[i] = F.Z[i - 1];
        const float g = i >= 5 ? 0.75f : 0.8f;   // the two fingers a shade darker
        S.bcol[i] = mk(g, g, g);
        S.balpha[i] = 1.f;
    }
    const int ncube = P.task == 4 ? 2 : 1;
    int nb = NARM;
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
    S.marker = P.has_target ? nb : -1;
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

// ray-test constants of one box for one camera (16 floats, see Scene::boxc)
DEV void box_consts(const LcrCam &C, f3 bc, f3 bX, f3 bY, f3 bZ, f3 bh, float alpha, float *c) {
    const f3 ro = mk(C.px, C.py, C.pz), CX = mk(C.xx, C.xy, C.xz), CY = mk(C.yx, C.yy, C.yz), CZ = mk(C.zx, C.zy, C.zz);
    const f3 d = ro - bc;
    c[0] = dot(bX, CX); c[1] = dot(bY, CX); c[2] = dot(bZ, CX); c[3] = dot(bX, d);
    c[4] = dot(bX, CY); c[5] = dot(bY, CY); c[6] = dot(bZ, CY); c[7] = dot(bY, d);
    c[8] = -dot(bX, CZ); c[9] = -dot(bY, CZ); c[10] = -dot(bZ, CZ); c[11] = dot(bZ, d);
    c[12] = bh.x; c[13] = bh.y; c[14] = bh.z; c[15] = alpha;
}

// per-camera constants of ONE primitive (called by one thread per (camera, primitive)): ray-test constants, screen bounding box, and for the arm boxes the
// 2D stadium that bounds the 8 projected corners -- axis through the projected centres of the two faces across the box's longest edge, half-width = the
// largest distance of a corner from that axis, ends at the extreme corners along it
DEV void build_prim(const LcrCam &C, int cam, int W, int H, Scene &S, int k) {
    float *cl = S.cull[cam][k];
    for (int i = 4; i < 12; i++) cl[i] = 0.f;
    if (k >= S.nbox) { cl[0] = (float)H; cl[1] = -1.f; cl[2] = (float)W; cl[3] = -1.f; return; }
    box_consts(C, S.bc[k], S.bX[k], S.bY[k], S.bZ[k], S.bh[k], S.balpha[k], S.boxc[cam][k]);
    const f3 ro = mk(C.px, C.py, C.pz), CX = mk(C.xx, C.xy, C.xz), CY = mk(C.yx, C.yy, C.yz), CZ = mk(C.zx, C.zy, C.zz);
    float uv[8][2];
    bool ok = true;
    float x0 = 1e30f, x1 = -1e30f, y0 = 1e30f, y1 = -1e30f;
    for (int i = 0; i < 8; i++) {
        const f3 p = axpy((i & 1) ? S.bh[k].x : -S.bh[k].x, S.bX[k], axpy((i & 2) ? S.bh[k].y : -S.bh[k].y, S.bY[k],
                     axpy((i & 4) ? S.bh[k].z : -S.bh[k].z, S.bZ[k], S.bc[k])));
        const f3 e = p - ro;
        const float xc = dot(e, CX), yc = dot(e, CY);
        float zc = -dot(e, CZ);
        if (zc < 0.02f) { ok = false; zc = 0.02f; }
        const float inv = 1.0f / (zc * C.s);
        uv[i][0] = 0.5f * W + xc * inv - 0.5f; uv[i][1] = 0.5f * H - yc * inv - 0.5f;   // pixel (px, row) has its centre at (px, row)
        x0 = fminf(x0, uv[i][0]); x1 = fmaxf(x1, uv[i][0]); y0 = fminf(y0, uv[i][1]); y1 = fmaxf(y1, uv[i][1]);
    }
    if (!ok) { cl[0] = 0.f; cl[1] = (float)(H - 1); cl[2] = 0.f; cl[3] = (float)(W - 1); cl[7] = 1e15f; cl[10] = 1e15f; cl[11] = 1.f; return; }
    cl[0] = floorf(y0 - 1.f); cl[1] = ceilf(y1 + 1.f); cl[2] = floorf(x0 - 1.f); cl[3] = ceilf(x1 + 1.f);
    if (k >= NARM) return;   // cubes, marker: the bounding box is all there is
    const int ax = S.bh[k].x >= S.bh[k].y ? (S.bh[k].x >= S.bh[k].z ? 0 : 2) : (S.bh[k].y >= S.bh[k].z ? 1 : 2);
    const int bit = 1 << ax;
    float a0u = 0.f, a0v = 0.f, a1u = 0.f, a1v = 0.f;
    for (int i = 0; i < 8; i++) {
        if (i & bit) { a1u += 0.25f * uv[i][0]; a1v += 0.25f * uv[i][1]; }
        else { a0u += 0.25f * uv[i][0]; a0v += 0.25f * uv[i][1]; }
    }
    float au = a1u - a0u, av = a1v - a0v;
    const float len = sqrtf(au * au + av * av);
    if (len > 1e-3f) { au /= len; av /= len; } else { au = 1.f; av = 0.f; }
    float smin = 1e30f, smax = -1e30f, wmax = 0.f;
    for (int i = 0; i < 8; i++) {
        const float eu = uv[i][0] - a0u, ev = uv[i][1] - a0v;
        const float sc = eu * au + ev * av;
        smin = fminf(smin, sc); smax = fmaxf(smax, sc);
        wmax = fmaxf(wmax, fabsf(eu * av - ev * au));
    }
    const float du = (smax - smin) * au, dv = (smax - smin) * av, R = wmax + 1.0f;
    const bool flat = fabsf(dv) < 1e-4f;
    const float inv_dv = flat ? 0.f : 1.0f / dv;
    cl[4] = a0u + smin * au; cl[5] = a0v + smin * av; cl[6] = du; cl[7] = R;
    // the stadium lies inside the infinite strip of half-width R around its axis: on row y the strip spans cx(y) -+ R / |sin(axis, row)|
    cl[8] = du * inv_dv; cl[9] = inv_dv; cl[10] = flat ? 1e15f : R * sqrtf(du * du + dv * dv) * fabsf(inv_dv); cl[11] = flat ? 1.f : 0.f;
}

// rgb in [0,1] -> 0x00BBGGRR with v_cvt_pk_u8_f32 (saturating float->byte conversion and byte insert in one instruction)
DEV unsigned pack_rgb(f3 c) {
    unsigned v = 0u;
    v = __builtin_amdgcn_cvt_pk_u8_f32(c.x * 255.f, 0, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(c.y * 255.f, 1, v);
    v = __builtin_amdgcn_cvt_pk_u8_f32(c.z * 255.f, 2, v);
    return v;
}

// slab test of one box along the UN-normalised ray ro + t d, d = C0 + sy B + sx A in the box frame (c = Scene::boxc record).  Returns the entry parameter
// (>= 0 means hit when it is also <= the exit) and |n . d| of the entry face: the face's normal is a box axis, so the headlight's Lambert term is a component of
// the box-frame direction -- no normal vector is ever formed.
DEV bool box_hit(const float *c, float sx, float sy, float tlimit, float &tmin, float &ld) {
    const float dx = fmaf(sx, c[0], fmaf(sy, c[4], c[8])), dy = fmaf(sx, c[1], fmaf(sy, c[5], c[9])), dz = fmaf(sx, c[2], fmaf(sy, c[6], c[10]));
    const float ix = rcp(fabsf(dx) > 1e-9f ? dx : 1e-9f), iy = rcp(fabsf(dy) > 1e-9f ? dy : 1e-9f), iz = rcp(fabsf(dz) > 1e-9f ? dz : 1e-9f);
    // slab: centre crossing -ol/dl, half width half/|dl|
    const float cx = -c[3] * ix, cy = -c[7] * iy, cz = -c[11] * iz;
    const float hx = c[12] * fabsf(ix), hy = c[13] * fabsf(iy), hz = c[14] * fabsf(iz);
    const float tnx = cx - hx, tny = cy - hy, tnz = cz - hz;
    tmin = fmaxf(tnx, fmaxf(tny, tnz));
    const float tmax = fminf(cx + hx, fminf(cy + hy, cz + hz));
    ld = tmin == tnx ? fabsf(dx) : (tmin == tny ? fabsf(dy) : fabsf(dz));
    return tmin <= tmax && tmin > 0.f && tmin < tlimit;
}

DEV f3 floor_or_sky(f3 ro, f3 d, float inv_len, float &tfloor) {
    // checker floor below the horizon (builtin checker, 0.1 m squares; its normal is +z so the Lambert term is -d_z / |d|), gradient sky above (unshaded)
    const float rdz = d.z * inv_len;
    if (rdz < -1e-6f) {
        tfloor = -ro.z * rcp(d.z);
        const float fx = fmaf(tfloor, d.x, ro.x), fy = fmaf(tfloor, d.y, ro.y);
        const int cell = ((int)floorf(fx * 10.f) + (int)floorf(fy * 10.f)) & 1;
        const float lam = fminf(fmaf(-0.6f, rdz, 0.3f), 1.f);
        return lam * (cell ? mk(0.2f, 0.3f, 0.4f) : mk(0.1f, 0.2f, 0.3f));
    }
    tfloor = 1e30f;
    const float a = clampf(rdz * 2.f, 0.f, 1.f);
    return mk(0.15f + a * 0.15f, 0.25f + a * 0.25f, 0.35f + a * 0.35f);
}

// one pixel, every primitive of the scene (cached background: the base only; render(): all): linear rgb in [0,1]
DEV f3 shade_pixel(const LcrCam &C, const float (*boxc)[16], const f3 *bcol, int nbox, int marker, float sx, float sy) {
    const f3 ro = mk(C.px, C.py, C.pz);
    const f3 d = mk(C.xx * sx + C.yx * sy - C.zx, C.xy * sx + C.yy * sy - C.zy, C.xz * sx + C.yz * sy - C.zz);
    const float inv_len = rsq(dot(d, d));
    float tbest;
    f3 out = floor_or_sky(ro, d, inv_len, tbest);
    float talpha = 0.f, tlamd = 0.f;
    for (int k = 0; k < nbox; k++) {
        float tmin, ld;
        if (!box_hit(boxc[k], sx, sy, tbest, tmin, ld)) continue;
        if (k == marker) { tlamd = ld; talpha = boxc[k][15]; }
        else { tbest = tmin; out = fminf(fmaf(0.6f * inv_len, ld, 0.3f), 1.f) * bcol[k]; }
    }
    if (talpha > 0.f) out = axpy(talpha * fmaf(0.6f * inv_len, tlamd, 0.3f), bcol[marker], (1.f - talpha) * out);
    return out;
}

// One tile of an observation band: ray-cast the boxes of mask `m` (wave-uniform).  The staged rows already hold the background (floor, sky, base), so a
// lane only reports a colour when its ray hits something else.  `stpx` = this pixel's 3 staged background bytes (read for the translucent target marker
// only).  Returns true and sets rgb when the pixel has to be rewritten.
DEV bool shade_span(const Scene &S, int cam, int marker, f3 ro, float sx, float sy, f3 d, unsigned m, const unsigned char *stpx, unsigned &rgb) {
    float tbest = d.z < -1e-6f ? -ro.z * rcp(d.z) : 1e30f;   // the floor hides what lies below it
    int kbest = -1;
    float lamd = 0.f;   // |n . d| of the nearest hit
    float talpha = 0.f, tlamd = 0.f;
    for (unsigned mm = m; mm; mm &= mm - 1u) {
        const int k = __builtin_ctz(mm);
        float tmin, ld;
        if (box_hit(S.boxc[cam][k], sx, sy, tbest, tmin, ld)) {
            if (k == marker) { tlamd = ld; talpha = S.boxc[cam][k][15]; }     // (the marker is the last box: every opaque primitive has been seen)
            else { tbest = tmin; lamd = ld; kbest = k; }
        }
    }
    const bool draw = kbest > BASE || talpha > 0.f;   // (a pixel whose nearest hit is the base keeps its background bytes)
    if (!__any(draw)) return false;
    const float inv_len = rsq(dot(d, d));
    const float lam = fminf(fmaf(0.6f * inv_len, lamd, 0.3f), 1.f);  // ambient + headlight (reach_cube.xml:8)
    f3 out = lam * S.bcol[kbest < 0 ? 0 : kbest];
    if (__any(talpha > 0.f)) {   // the translucent marker: over the opaque hit, or over the staged background
        if (kbest <= BASE) out = mk(stpx[0] * (1.f / 255.f), stpx[1] * (1.f / 255.f), stpx[2] * (1.f / 255.f));
        const float tl = fmaf(0.6f * inv_len, tlamd, 0.3f);
        if (talpha > 0.f) out = axpy(talpha * tl, S.bcol[marker < 0 ? 0 : marker], (1.f - talpha) * out);
    }
    rgb = pack_rgb(out);
    return draw;
}

// background frames of the two observation cameras -- checker floor, sky and the arm's base: identical for every env and every step, so they
// are rendered ONCE at lcr_create into P.img_bg ([2][240][320][3], 460 800 B, L2-resident) and copied band-wise afterwards.
__global__ __launch_bounds__(256) void lcr_render_bg_kernel(LcrDev P, LcrCam front, LcrCam top) {
    const int W = 320, H = 240;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= 2 * W * H) return;
    const bool is_top = pix >= W * H;
    const int p = is_top ? pix - W * H : pix;
    const int row = p / W, px = p - row * W;
    const LcrCam &C = is_top ? top : front;
    const float sy = -(row + 0.5f - 0.5f * H) * C.s, sx = (px + 0.5f - 0.5f * W) * C.s;
    f3 bc, bX, bY, bZ, bh;
    base_box(bc, bX, bY, bZ, bh);
    float boxc[1][16];
    box_consts(C, bc, bX, bY, bZ, bh, 1.f, boxc[0]);
    const f3 col = mk(0.8f, 0.8f, 0.8f);
    const unsigned rgb = pack_rgb(shade_pixel(C, boxc, &col, 1, -1, sx, sy));
    P.img_bg[3 * (size_t)pix + 0] = (unsigned char)rgb;
    P.img_bg[3 * (size_t)pix + 1] = (unsigned char)(rgb >> 8);
    P.img_bg[3 * (size_t)pix + 2] = (unsigned char)(rgb >> 16);
}

// OR of a value over lanes 0 .. 15 (the primitives live in lanes 0 .. NPRIM-1): four row_shr DPP steps, the result is read from lane 15
DEV unsigned or_row0(unsigned x) {
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);   // row_shr:1
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);   // row_shr:2
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);   // row_shr:4
    x |= (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);   // row_shr:8
    return (unsigned)__builtin_amdgcn_readlane((int)x, 15);
}

// COUNT: diagnostics build (LCR_RENDER_COUNT=1 and lcr_config.diagnostics): ray-cast passes / primitive tests / pixels written per env into
// active_count / choice / max_sweeps (tools/render_work.py)
// Six waves per SIMD: the 80 registers that takes spill six values of the scene set-up (the chain of link frames), none in the band loop (measured, 32 768 envs:
// 5 waves 2.88 ms, 6: 2.77, 7: 2.92, 8: 2.95; the rounds 1-4 structure with its background prefetch in registers ran 4).
template <bool COUNT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 6))) void lcr_render_obs_kernel(LcrDev P, LcrCam front, LcrCam top) {
    // A workgroup owns one env (480 rows: front frame then top frame); a wave handles one BAND of 4 rows at a time (3 840 B = 240 lanes x
    // 16 B).  The band starts as a copy of the cached background band (L2 hit); if no primitive's silhouette touches
    // it (wave-uniform) it leaves straight away as non-temporal 16-B stores.  Otherwise the band is
    // staged in LDS, its 16 x 4-pixel tiles that a silhouette touches are ray-cast and
    // overwrite their bytes, and the band is stored from LDS.  No software prefetch across bands and the culling records in LDS rather than in
    // registers: both buy occupancy (six waves per SIMD), which is what hides the L2 latency of the background rows and keeps the store queue fed while
    // other waves ray-cast.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int W = 320, H = 240, NT = W / 16;
    static_assert(NPRIM <= 16, "one primitive per lane of a DPP row");
    __shared__ Scene S;
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][4 * 960];
    const int env = blockIdx.x;
    if (threadIdx.x == 0) build_scene(P, env, S);
    __syncthreads();
    if (threadIdx.x < 2 * NPRIM) {
        const int cam_id = threadIdx.x / NPRIM, prim = threadIdx.x - cam_id * NPRIM;
        build_prim(cam_id ? top : front, cam_id, W, H, S, prim);
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t img_bytes = (size_t)H * W * 3;
    unsigned char *st = stage[wave];
    // waves 0,1 render camera_front (even / odd bands), waves 2,3 camera_top: everything camera-dependent is wave-uniform
    const int cam = wave >> 1;
    const LcrCam &C = cam ? top : front;
    const f3 ro = mk(C.px, C.py, C.pz), CX = mk(C.xx, C.xy, C.xz), CY = mk(C.yx, C.yy, C.yz), CZ = mk(C.zx, C.zy, C.zz);
    const int marker = __builtin_amdgcn_readfirstlane(S.marker);
    // culling: one primitive per lane (lane k <-> primitive k).  Per band, every lane computes the pixel interval [xa, xb] its primitive can cover on the
    // band's 4 rows (arm boxes: 2D stadium silhouette, the other boxes: bounding box) -> the interval [ta, tb] of 16-pixel tile columns it touches.
    const float *cull = S.cull[cam][lane < NPRIM ? lane : 0];
    const u32x4 *bg = reinterpret_cast<const u32x4 *>(P.img_bg) + (size_t)cam * H * 60;
    u32x4 *out = reinterpret_cast<u32x4 *>((cam ? P.img_top : P.img_front) + (size_t)env * img_bytes);
    const bool l3 = lane < 48;               // a band = 240 vectors = 3 full wave loads + 48 lanes
    const int lq = l3 ? lane : 47;
    // co-resident workgroups start at different bands (hashed phase) so that their ray-cast (VALU-bound) and copy
    // (memory-bound) stretches overlap instead of all waves of a SIMD hitting the arm's rows together
    constexpr int NB = H / 4;
    const int rot = (int)((blockIdx.x * 0x9E3779B1u) >> 29) * 8;
    const int tx = lane & 15, ty = lane >> 4;   // pixel of this lane inside a 16 x 4 tile
    for (int it = wave & 1; it < NB; it += 2) {
        const int b = it + rot >= NB ? it + rot - NB : it + rot;
        u32x4 *dst = out + b * 240;
        const u32x4 *src = bg + b * 240;
        const u32x4 v0 = src[lane], v1 = src[64 + lane], v2 = src[128 + lane], v3 = src[192 + lq];   // in flight while the band is culled
        const int row0 = 4 * b;
        const f32x4 cb = *reinterpret_cast<const f32x4 *>(cull);   // y0 y1 x0 x1
        // (the base alone does not make a band worth ray-casting: it is in the background already)
        const bool in_band = lane < NPRIM && (float)(row0 + 3) >= cb.x && (float)row0 <= cb.y;
        int ta = 1 << 20, tb = -1;   // tile columns this lane's primitive can touch on rows row0 .. row0+3 (empty: ta > tb)
        if (in_band) {
            float xa = cb.z, xb = cb.w;
            if (lane < NARM) {
                const f32x4 cs = *reinterpret_cast<const f32x4 *>(cull + 4), ct = *reinterpret_cast<const f32x4 *>(cull + 8);
                const float au = cs.x, av = cs.y, du = cs.z, R = cs.w, slope = ct.x, inv_dv = ct.y, strip_hw = ct.z;
                const bool flat = ct.w != 0.f;
                const float f0 = (float)row0 - R - av, f1 = (float)(row0 + 3) + R - av;
                const float sA = f0 * inv_dv, sB = f1 * inv_dv;
                const float s0 = flat ? 0.f : clampf(fminf(sA, sB), 0.f, 1.f), s1 = flat ? 1.f : clampf(fmaxf(sA, sB), 0.f, 1.f);
                const float e0 = s0 * du, e1 = s1 * du;
                const float c0 = fmaf((float)row0 - av, slope, au), c1 = fmaf((float)(row0 + 3) - av, slope, au);
                xa = fmaxf(fmaxf(au + fminf(e0, e1) - R, fminf(c0, c1) - strip_hw), xa);
                xb = fminf(fminf(au + fmaxf(e0, e1) + R, fmaxf(c0, c1) + strip_hw), xb);
            }
            if (xb >= 0.f && xa <= (float)(W - 1) && xa <= xb) {
                ta = (int)fmaxf(xa * (1.f / 16.f), 0.f); tb = (int)fminf(xb * (1.f / 16.f), (float)(NT - 1));
            }
        }
        // tile columns of the band that a primitive other than the base touches
        unsigned U = 0u;
        if (__any(ta <= tb && lane != BASE)) U = or_row0(ta <= tb && lane != BASE ? (2u << tb) - (1u << ta) : 0u);
        if (U == 0u) {
            __builtin_nontemporal_store(v0, dst + lane);
            __builtin_nontemporal_store(v1, dst + 64 + lane);
            __builtin_nontemporal_store(v2, dst + 128 + lane);
            if (l3) __builtin_nontemporal_store(v3, dst + 192 + lane);
            continue;
        }
        u32x4 *sv = reinterpret_cast<u32x4 *>(st);
        sv[lane] = v0; sv[64 + lane] = v1; sv[128 + lane] = v2;
        if (l3) sv[192 + lane] = v3;
        // rays of this lane's tile row: d(px) = -Z + sy Y + sx(px) X, camera looks along -Z
        const float sy = -((float)(row0 + ty) + 0.5f - 0.5f * H) * C.s;
        const f3 rbase = axpy(sy, CY, neg(CZ));
        for (; U; U &= U - 1u) {
            const int t = __builtin_ctz(U);
            const unsigned m = (unsigned)__ballot(ta <= t && tb >= t);
            const int px = 16 * t + tx;
            const float sx = ((float)px + 0.5f - 0.5f * W) * C.s;
            const f3 rdu = axpy(sx, CX, rbase);
            unsigned char *stpx = st + ty * 960 + 3 * px;
            unsigned rgb = 0u;
            const bool wrote = shade_span(S, cam, marker, ro, sx, sy, rdu, m, stpx, rgb);
            if (wrote) {
                stpx[0] = (unsigned char)rgb;
                stpx[1] = (unsigned char)(rgb >> 8);
                stpx[2] = (unsigned char)(rgb >> 16);
            }
            if (COUNT && P.active_count) {
                const int nw = __popcll(__ballot(wrote));
                if (lane == 0) { atomicAdd(&P.active_count[env], 1u); atomicAdd(&P.choice[env], (unsigned)__popc(m)); atomicAdd(&P.max_sweeps[env], (unsigned)nw); }
            }
        }
        __builtin_nontemporal_store(sv[lane], dst + lane);
        __builtin_nontemporal_store(sv[64 + lane], dst + 64 + lane);
        __builtin_nontemporal_store(sv[128 + lane], dst + 128 + lane);
        if (l3) __builtin_nontemporal_store(sv[192 + lane], dst + 192 + lane);
    }
}

// one env, arbitrary camera / resolution (render(), 640x640 camera_vizu): one thread per pixel, no culling, no cached background
__global__ __launch_bounds__(256) void lcr_render_single_kernel(LcrDev P, LcrCam cam, int env, int W, int H, unsigned char *out) {
    __shared__ Scene S;
    if (threadIdx.x == 0) build_scene(P, env, S);
    __syncthreads();
    if (threadIdx.x < NBOX && (int)threadIdx.x < S.nbox)
        box_consts(cam, S.bc[threadIdx.x], S.bX[threadIdx.x], S.bY[threadIdx.x], S.bZ[threadIdx.x], S.bh[threadIdx.x], S.balpha[threadIdx.x], S.boxc[0][threadIdx.x]);
    __syncthreads();
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= W * H) return;
    const int v = pix / W, u = pix - v * W;
    const float sx = (u + 0.5f - 0.5f * W) * cam.s, sy = -(v + 0.5f - 0.5f * H) * cam.s;
    const unsigned rgb = pack_rgb(shade_pixel(cam, S.boxc[0], S.bcol, S.nbox, S.marker, sx, sy));
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
    static const int count = [] { const char *e = getenv("LCR_RENDER_COUNT"); return e ? atoi(e) : 0; }();   // diagnostics: tools/render_work.py
    if (count) hipLaunchKernelGGL(lcr_render_obs_kernel<true>, dim3(P.n), dim3(256), 0, (hipStream_t)stream, P, front, top);
    else hipLaunchKernelGGL(lcr_render_obs_kernel<false>, dim3(P.n), dim3(256), 0, (hipStream_t)stream, P, front, top);
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
