/*
 * lcr_oracle.c -- CPU ORACLE (test infrastructure, see lcr_oracle.h for scope and parity status).
 *
 * Every function cites the reference lines it restates.  "MJ-DOC" marks statements about the
 * absent third-party `mujoco` library taken from its public documentation (Computation chapter,
 * XML reference); those cannot be checked in this image ("parity unpinned").
 *
 * The code is deliberately GENERIC and DENSE (world-frame Jacobian sums for the mass matrix, a dense
 * constraint Jacobian, a dense Delassus matrix, generic Cholesky) so that it shares neither source
 * nor formulation with the hand-unrolled, matrix-free HIP kernel it checks.
 */
#include "lcr_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef orc_real real;

/* ------------------------------------------------------------------------------------------------
 * L0: model constants, transcribed from the MJCF (numbers only).
 * follower.xml:3   integrator=implicitfast cone=elliptic impratio=100 timestep=0.002
 * follower.xml:7   joint armature=0.1 damping=1 actuatorfrcrange=+-10
 * follower.xml:8   position kp=1000 kv=10 inheritrange=1 (ctrlrange := joint range)
 * follower.xml:51  base quat (-0.707 0 0 0.707)  == Rz(-90deg) after normalisation
 * follower.xml:56-98 bodies / joints / inertials; :91 site
 * ---------------------------------------------------------------------------------------------- */
#define H_STEP 0.002
#define ARMATURE 0.1
#define DAMPING 1.0
#define KP 1000.0
#define KV 10.0
#define FRC_LIM 10.0
#define GRAV 9.81
#define CUBE_HALF 0.015 /* reach_cube.xml:26 size="0.015 0.015 0.015" */
#define MJ_MINVAL 1e-15
#define MJ_MINIMP 0.0001
#define MJ_MAXIMP 0.9999

static const double LINK_POS[6][3] = {
    {0.012, 0, 0.0409},          /* follower.xml:56 */
    {0, -0.0209, 0.0154},        /* :63 */
    {-0.0148, 0.0065, 0.1083},   /* :70 */
    {-0.10048, 5e-05, 0.0026999},/* :77 */
    {-0.045, 0.013097, 0},       /* :84 */
    {-0.01315, -0.0075, 0.0145}, /* :93 */
};
static const double LINK_AXIS[6][3] = {
    {0, 0, -1}, {0, 1, 0}, {0, -1, 0}, {0, 1, 0}, {1, 0, 0}, {0, 0, -1}, /* :58,65,72,79,86,95 */
};
static const double LINK_IPOS[6][3] = {
    {0.011924, -0.00048792, 0.013381}, {0.0011747, 0.02097, 0.071547},  {-0.05537, 0.014505, 0.0028659},
    {-0.02652, 0.019195, -9.0614e-06}, {-0.019091, 0.0053379, 0.00018011}, {-0.02507, 0.0010817, -0.01414},
};
static const double LINK_IQUAT[6][4] = {
    {-0.0190903, 0.705417, 0.0178052, 0.708312},   {0.998768, 2.01447e-05, 0.0496266, 0.000367169},
    {8.17663e-05, 0.710999, -4.16983e-05, 0.703193}, {0.707361, 0.706812, 0.00580344, 0.00484124},
    {0.105295, 0.703509, -0.0986543, 0.695885},    {0.528148, 0.5474, 0.466496, 0.451436},
};
static const double LINK_MASS[6] = {0.05014, 0.050177, 0.06379, 0.019805, 0.029277, 0.012831};
static const double LINK_DIAGI[6][3] = {
    {1.44921e-05, 1.2371e-05, 7.59138e-06}, {3.73065e-05, 3.3772e-05, 7.94901e-06},
    {2.45081e-05, 2.2231e-05, 7.34061e-06}, {2.95813e-06, 2.8759e-06, 1.07787e-06},
    {8.11303e-06, 7.14908e-06, 3.27429e-06}, {3.49922e-06, 2.45768e-06, 1.4645e-06},
};
/* follower.xml:58-95 joint ranges (== actuator ctrlrange through inheritrange) */
static const double JNT_LO[6] = {-3.14, -3.14, -3.14, -3.14, -3.14, -2.45};
static const double JNT_HI[6] = {3.14, 3.14, 3.14, 3.14, 3.14, 0.032};
/* reach_cube_env.py:249-250 hard-coded joint-mode clip (REF-QUIRK-6) */
static const double TGT_LO[6] = {-3.14159, -1.5708, -1.48353, -1.91986, -2.96706, -1.74533};
static const double TGT_HI[6] = {3.14159, 1.22173, 1.74533, 1.91986, 2.96706, 0.0523599};
static const double SITE_POS[3] = {-0.06429, 0.00327, 0.0011}; /* follower.xml:91, on link_5 */

/* (D3) finger proxies: ONE sphere per finger geom (MuJoCo's convex mesh collider also yields one contact
 * point per geom pair), fitted to the tip of the fixed finger of the link_5_collision hull and to the
 * jaw tip of the link_6_collision hull (follower.xml:89,97; extents in SURVEY.md 8(c)).
 * {link index 0..5, centre in link frame, radius} */
#define NSPH 2
static const int SPH_LINK[NSPH] = {4, 5};
static const double SPH_POS[NSPH][3] = {{-0.0610, 0.0142, 0.0005}, {-0.0490, 0.0072, -0.0140}};
static const double SPH_RAD[NSPH] = {0.0065, 0.0065};
/* (round 5, orc_params.finger_geom = 1 = preset faithful) the finger pads as POLYTOPES: one box per finger -- the bounding box of the two outermost slabs of
 * the finger's collision hull (tests/golden/model_golden.json "mesh_slabs_x" link_5_collision / link_6_collision [0..1]; follower.xml:15,89,97: the finger geoms
 * are those hulls) -- in the link frame: centre, half extents.  17 x 12 x 15 mm and 16 x 15 x 16 mm; the spheres above are inscribed in them. */
static const double PAD_C[NSPH][3] = {{-0.06051, 0.01414, 0.000345}, {-0.04772, 0.005415, -0.01394}};
static const double PAD_H[NSPH][3] = {{0.0087, 0.00579, 0.007575}, {0.008, 0.007715, 0.00799}};
#define PAD_BLEND 0.0005   /* vertices within 0.5 mm of the deepest one share the contact point (weights linear in depth): a flat pad face rests on its middle */
/* (D3) arm-link proxies: the remaining arm geoms that can reach the floor or the cube (follower.xml:10 visual geoms have
 * contype=conaffinity=1, :13 collision hulls; geoms :70-98) are restated as spheres inscribed in the motor / bracket
 * volumes of their hulls (mesh extents: tests/golden/model_golden.json "mesh_aabb").  link_1 / link_2 / base cannot
 * reach the floor inside the joint-mode target box (reach_cube_env.py:249-250) and carry no proxy.
 * Together the proxies yield at most ONE contact per substep -- the deepest penetration among all candidates (ties: the first
 * in the order proxy 0 floor, proxy 0 cubes, proxy 1 floor, ...) -- the way MuJoCo's convex collider yields one point per
 * geom pair; every proxy is tested against the floor, the gripper-body proxies (link_5 motor body, link_6 jaw root) also
 * against the cube(s).  (Measured under a random policy: a proxy touches in 3 % of the env-steps, two links at once in
 * < 0.1 %; the one that is not served keeps its warm start for the next substep in which it is the deepest.)
 * {link index 0..5, centre in link frame, radius, collides with cubes} */
#define NLPX 5
#define NLGRP 3   /* contact groups the proxies may form: 1 in the product (one shared contact, = the kernels); 3 in the study of deviation D3 (orc_params.proxy_groups) */
static const int LPX_LINK[NLPX] = {2, 2, 3, 4, 5};
static const double LPX_POS[NLPX][3] = {
    {-0.0100, 0.0145, 0.0030},  /* elbow end of link_3            (link_3_collision x[-0.111,0.009] y[-0.004,0.033] z[-0.009,0.015]) */
    {-0.0950, 0.0145, 0.0030},  /* wrist-motor end of link_3      (same hull, x[-0.111,-0.081]) */
    {-0.0320, 0.0206, 0.0000},  /* link_4 motor                   (link_4_collision x[-0.045,-0.018] y[0.003,0.038] z[-0.010,0.010]) */
    {-0.0130, 0.0015, 0.0000},  /* link_5 motor body              (link_5_collision x[-0.026,0] y[-0.018,0.021] z[-0.015,0.015]) */
    {-0.0120, 0.0000, -0.0145}, /* jaw root on link_6             (link_6_collision x[-0.032,0.008] y[-0.008,0.008] z[-0.032,0.003]) */
};
/* (against a plane a set of spheres acts like its convex hull, so the two ends of a link stand for the whole link; the
 * fixed finger between the link_5 body and the finger-tip sphere needs no proxy of its own, and the grasp gap stays free) */
static const double LPX_RAD[NLPX] = {0.0120, 0.0120, 0.0105, 0.0150, 0.0078};
static const int LPX_GROUP[NLPX] = {0, 0, 0, 0, 0};
static const int LPX_GROUP3[NLPX] = {0, 1, 2, 2, 2};   /* study: one contact per end of link_3 and one for the wrist / gripper body (tools/proxy_groups_effect.py) */
static const int LPX_CUBE[NLPX] = {0, 0, 0, 1, 1};
/* link geoms: MuJoCo geom defaults friction (1, 0.005, 0.0001), condim 3, priority 0.  vs floor (priority 0, friction 0.1):
 * max rule -> mu 1, condim 3, default solref/solimp.  vs cube (priority 1): the cube's condim 4, friction, solimp win (P9). */
static const double MU_LINK_FLOOR[5] = {1.0, 1.0, 0.005, 0.0001, 0.0001};

/* default solver parameters (MJ-DOC XML reference): solref=(0.02,1) solimp=(0.9,0.95,0.001,0.5,2) */
static const double SOLREF[2] = {0.02, 1.0};
static const double SOLIMP_DEFAULT[5] = {0.9, 0.95, 0.001, 0.5, 2.0};
/* follower.xml:15 finger class solimp="0.015 1 0.036" (trailing values default), friction 1.5, priority 1 */
static const double SOLIMP_FINGER[5] = {0.015, 1.0, 0.036, 0.5, 2.0};
/* P9: cube (priority 1) vs finger (priority 1): equal priority -> solimp averaged (solmix 1:1), friction max */
static const double SOLIMP_FINGER_CUBE[5] = {0.4575, 0.975, 0.0185, 0.5, 2.0};
static const double MU_CUBE[5] = {0.5, 0.5, 0.005, 0.0001, 0.0001};   /* reach_cube.xml:26 friction="0.5" (+default torsional, rolling) */
static const double MU_FINGER[5] = {1.5, 1.5, 0.005, 0.0001, 0.0001}; /* follower.xml:15 friction="1.5"; entries 3, 4 (rolling) only with condim6 */
/* push_cube_loop.xml:31 cube friction="1.5 1.5 1.5" (tangential, torsional, rolling), priority 1: used against the floor
 * and the walls (priority 0) and, by the max rule, against the fingers (priority 1) */
static const double MU_LOOP[5] = {1.5, 1.5, 1.5, 1.5, 1.5};   /* push_cube_loop.xml:31 friction="1.5 1.5 1.5" */
/* push_cube_loop.xml:44-47 rails: inner faces of the four wall boxes, top of the walls */
#define WALL_X 0.115
#define WALL_Y0 0.10
#define WALL_Y1 0.17
#define WALL_TOP 0.012
#define RAIL_CLAMP 0.012   /* largest penetration a rail's inner face reports for the cube */
#define WALL_THICK 0.02   /* the rail boxes are 2 x 0.01 thick: outer faces WALL_THICK beyond the inner ones (push_cube_loop.xml:45-48 size) */

typedef struct {
    int ncube;
    double cube_mass, cube_inertia;
    int has_target;
    int walls;            /* PushCubeLoop rails */
    const double *mu_cube; /* cube geom friction */
    const double *mu_finger_cube;
} task_model;

static task_model get_task_model(int task) {
    task_model t;
    t.ncube = 1;
    t.cube_mass = 0.1;          /* reach_cube.xml:25, lift_cube.xml:27, push_cube.xml:27 */
    t.cube_inertia = 0.00016667;
    t.has_target = 0;
    if (task == ORC_TASK_PICK_PLACE) { t.cube_mass = 10.0; t.has_target = 1; } /* pick_place_cube.xml:27 REF-QUIRK-4 */
    if (task == ORC_TASK_PUSH) t.has_target = 1;
    if (task == ORC_TASK_STACK) { t.ncube = 2; t.cube_inertia = 0.00001125; }   /* stack_two_cubes.xml:27,33 */
    t.walls = 0; t.mu_cube = MU_CUBE; t.mu_finger_cube = MU_FINGER;
    if (task == ORC_TASK_PUSH_LOOP) { /* push_cube_loop.xml:29-31 */
        t.cube_mass = 0.05; t.cube_inertia = 0.00001125; t.walls = 1; t.mu_cube = MU_LOOP; t.mu_finger_cube = MU_LOOP;
    }
    return t;
}

/* ------------------------------------------------------------------------------------------------ */
/* small linear algebra                                                                             */
/* ------------------------------------------------------------------------------------------------ */
static inline void v3set(real *a, real x, real y, real z) { a[0] = x; a[1] = y; a[2] = z; }
static inline void v3copy(real *a, const real *b) { a[0] = b[0]; a[1] = b[1]; a[2] = b[2]; }
static inline void v3add(real *o, const real *a, const real *b) { o[0] = a[0] + b[0]; o[1] = a[1] + b[1]; o[2] = a[2] + b[2]; }
static inline void v3sub(real *o, const real *a, const real *b) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; }
static inline void v3axpy(real *o, real s, const real *b) { o[0] += s * b[0]; o[1] += s * b[1]; o[2] += s * b[2]; }
static inline real v3dot(const real *a, const real *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void v3cross(real *o, const real *a, const real *b) {
    real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline real v3norm(const real *a) { return (real)sqrt((double)v3dot(a, a)); }
/* o = R v  (R row-major 3x3) */
static inline void m3v(real *o, const real *R, const real *v) {
    real x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    real y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    real z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void m3tv(real *o, const real *R, const real *v) {
    real x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    real y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    real z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void m3mul(real *o, const real *A, const real *B) {
    real t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(o, t, sizeof t);
}
static void quat2mat(real *R, const real *q) { /* q = (w,x,y,z), unit */
    real w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - w * z); R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z); R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y); R[7] = 2 * (y * z + w * x); R[8] = 1 - 2 * (x * x + y * y);
}
static void axisangle2mat(real *R, const real *a, real th) { /* Rodrigues, |a| = 1 */
    real c = (real)cos((double)th), s = (real)sin((double)th), C = 1 - c;
    R[0] = c + a[0] * a[0] * C;        R[1] = a[0] * a[1] * C - a[2] * s; R[2] = a[0] * a[2] * C + a[1] * s;
    R[3] = a[1] * a[0] * C + a[2] * s; R[4] = c + a[1] * a[1] * C;        R[5] = a[1] * a[2] * C - a[0] * s;
    R[6] = a[2] * a[0] * C - a[1] * s; R[7] = a[2] * a[1] * C + a[0] * s; R[8] = c + a[2] * a[2] * C;
}
/* dense Cholesky A = L L^T in place (lower), n <= ORC_NV_MAX; returns 0 on success */
static int chol(real *A, int n) {
    for (int j = 0; j < n; j++) {
        real d = A[j * n + j];
        for (int k = 0; k < j; k++) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0)) return -1;
        d = (real)sqrt((double)d);
        A[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            real s = A[i * n + j];
            for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
            A[i * n + j] = s / d;
        }
    }
    return 0;
}
static void chol_solve(const real *L, int n, real *x) { /* x <- (L L^T)^-1 x */
    for (int i = 0; i < n; i++) {
        real s = x[i];
        for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k];
        x[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; i--) {
        real s = x[i];
        for (int k = i + 1; k < n; k++) s -= L[k * n + i] * x[k];
        x[i] = s / L[i * n + i];
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* kinematics (MJ-DOC mj_kinematics + mj_comPos for this fixed tree)                                */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    real R[7][9]; /* [0] base_link, [i] link_i */
    real p[7][3];
    real z[6][3];   /* world joint axes */
    real com[6][3]; /* world inertial-frame origins (xipos) */
    real Iw[6][9];  /* world inertia about com */
    real site[3];
    real sph[NSPH][3];
    real lpx[NLPX][3];
    int ncube;
    real cR[2][9], cp[2][3];
    const double *mu_cube, *mu_finger_cube;
    int cc_points;    /* cube<->cube manifold: 4 (default) or 8 points (study of deviation D5) */
} kin_t;

static void arm_kinematics(const real *q, kin_t *K) {
    /* base: follower.xml:51 quat normalised = (-sqrt.5,0,0,sqrt.5) -> [[0,1,0],[-1,0,0],[0,0,1]] */
    real bq[4] = {(real)-0.707, 0, 0, (real)0.707};
    real nn = (real)sqrt((double)(bq[0] * bq[0] + bq[3] * bq[3]));
    bq[0] /= nn; bq[3] /= nn;
    quat2mat(K->R[0], bq);
    v3set(K->p[0], 0, 0, 0);
    for (int i = 0; i < 6; i++) {
        real lp[3] = {(real)LINK_POS[i][0], (real)LINK_POS[i][1], (real)LINK_POS[i][2]};
        real ax[3] = {(real)LINK_AXIS[i][0], (real)LINK_AXIS[i][1], (real)LINK_AXIS[i][2]};
        real t[3], Rl[9];
        m3v(t, K->R[i], lp);
        v3add(K->p[i + 1], K->p[i], t);
        axisangle2mat(Rl, ax, q[i]);
        m3mul(K->R[i + 1], K->R[i], Rl);
        m3v(K->z[i], K->R[i + 1], ax);
        real ip[3] = {(real)LINK_IPOS[i][0], (real)LINK_IPOS[i][1], (real)LINK_IPOS[i][2]};
        m3v(t, K->R[i + 1], ip);
        v3add(K->com[i], K->p[i + 1], t);
        /* world inertia: Rw = R * Riq ; Iw = Rw diag Rw^T */
        real iq[4], n2 = 0, Riq[9], Rw[9];
        for (int k = 0; k < 4; k++) { iq[k] = (real)LINK_IQUAT[i][k]; n2 += iq[k] * iq[k]; }
        n2 = (real)sqrt((double)n2);
        for (int k = 0; k < 4; k++) iq[k] /= n2;
        quat2mat(Riq, iq);
        m3mul(Rw, K->R[i + 1], Riq);
        for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) {
                real s = 0;
                for (int k = 0; k < 3; k++) s += Rw[3 * a + k] * (real)LINK_DIAGI[i][k] * Rw[3 * b + k];
                K->Iw[i][3 * a + b] = s;
            }
    }
    real sp[3] = {(real)SITE_POS[0], (real)SITE_POS[1], (real)SITE_POS[2]}, t[3];
    m3v(t, K->R[5], sp);
    v3add(K->site, K->p[5], t);
    for (int s = 0; s < NSPH; s++) {
        real c[3] = {(real)SPH_POS[s][0], (real)SPH_POS[s][1], (real)SPH_POS[s][2]};
        m3v(t, K->R[SPH_LINK[s] + 1], c);
        v3add(K->sph[s], K->p[SPH_LINK[s] + 1], t);
    }
    for (int s = 0; s < NLPX; s++) {
        real c[3] = {(real)LPX_POS[s][0], (real)LPX_POS[s][1], (real)LPX_POS[s][2]};
        m3v(t, K->R[LPX_LINK[s] + 1], c);
        v3add(K->lpx[s], K->p[LPX_LINK[s] + 1], t);
    }
}

/* point Jacobians of body b (0..5 arm link, 6/7 cube) at world point pt: Jp,Jr are [3][nv] */
static void jac_point(const kin_t *K, int nv, int b, const real *pt, real *Jp, real *Jr) {
    memset(Jp, 0, sizeof(real) * 3 * nv);
    memset(Jr, 0, sizeof(real) * 3 * nv);
    if (b < 0) return;
    if (b < 6) {
        for (int j = 0; j <= b; j++) {
            real r[3], c[3];
            v3sub(r, pt, K->p[j + 1]);
            v3cross(c, K->z[j], r);
            for (int k = 0; k < 3; k++) { Jp[k * nv + j] = c[k]; Jr[k * nv + j] = K->z[j][k]; }
        }
    } else {
        int c = b - 6, o = 6 + 6 * c;
        real r[3];
        v3sub(r, pt, K->cp[c]);
        for (int i = 0; i < 3; i++) {
            Jp[i * nv + o + i] = 1; /* linear dofs: world frame */
            real col[3] = {K->cR[c][i], K->cR[c][3 + i], K->cR[c][6 + i]}, cr[3]; /* R e_i */
            v3cross(cr, col, r); /* d(v_pt)/d(omega_body_i) = (R e_i) x r   (MJ-DOC: free-joint angular dofs are body-frame) */
            for (int k = 0; k < 3; k++) { Jp[k * nv + o + 3 + i] = cr[k]; Jr[k * nv + o + 3 + i] = col[k]; }
        }
    }
}

/* arm joint-space inertia: M = sum_i m_i Jv_i^T Jv_i + Jw_i^T I_i Jw_i (+ armature)  (== MJ-DOC CRBA result) */
static void arm_mass(const kin_t *K, int with_armature, real *M /*6x6*/) {
    memset(M, 0, sizeof(real) * 36);
    for (int i = 0; i < 6; i++) {
        real Jp[18], Jr[18];
        jac_point(K, 6, i, K->com[i], Jp, Jr);
        for (int a = 0; a <= i; a++)
            for (int b = 0; b <= i; b++) {
                real s = 0;
                for (int k = 0; k < 3; k++) s += (real)LINK_MASS[i] * Jp[k * 6 + a] * Jp[k * 6 + b];
                for (int k = 0; k < 3; k++)
                    for (int l = 0; l < 3; l++) s += Jr[k * 6 + a] * K->Iw[i][3 * k + l] * Jr[l * 6 + b];
                M[a * 6 + b] += s;
            }
    }
    if (with_armature)
        for (int j = 0; j < 6; j++) M[j * 6 + j] += (real)ARMATURE;
}

/* arm bias forces c(q,qd) incl. gravity: recursive Newton-Euler with zero joint acceleration (MJ-DOC mj_rne) */
static void arm_bias(const kin_t *K, const real *qd, real *bias) {
    real w[7][3], wd[7][3], a[7][3]; /* angular vel/acc, linear acc of body origin */
    real F[6][3], N[6][3];
    v3set(w[0], 0, 0, 0); v3set(wd[0], 0, 0, 0); v3set(a[0], 0, 0, (real)GRAV);
    for (int i = 0; i < 6; i++) {
        real zq[3] = {K->z[i][0] * qd[i], K->z[i][1] * qd[i], K->z[i][2] * qd[i]}, t[3], r[3];
        v3add(w[i + 1], w[i], zq);
        v3cross(t, w[i], zq);
        v3add(wd[i + 1], wd[i], t);
        v3sub(r, K->p[i + 1], K->p[i]);
        v3cross(t, wd[i], r);
        v3add(a[i + 1], a[i], t);
        real wr[3];
        v3cross(wr, w[i], r);
        v3cross(t, w[i], wr);
        v3add(a[i + 1], a[i + 1], t);
        /* com acceleration */
        real rc[3], ac[3];
        v3sub(rc, K->com[i], K->p[i + 1]);
        v3cross(t, wd[i + 1], rc);
        v3add(ac, a[i + 1], t);
        v3cross(wr, w[i + 1], rc);
        v3cross(t, w[i + 1], wr);
        v3add(ac, ac, t);
        for (int k = 0; k < 3; k++) F[i][k] = (real)LINK_MASS[i] * ac[k];
        real Iw_[3], Iwd[3];
        m3v(Iw_, K->Iw[i], w[i + 1]);
        m3v(Iwd, K->Iw[i], wd[i + 1]);
        v3cross(t, w[i + 1], Iw_);
        v3add(N[i], Iwd, t);
    }
    real f[3] = {0, 0, 0}, n[3] = {0, 0, 0}; /* force/torque transmitted through joint i+1, torque about p[i+2] */
    for (int i = 5; i >= 0; i--) {
        /* n_i (about p[i+1]) = N_i + (com - p) x F_i + n_{i+1} + (p_{i+2} - p_{i+1}) x f_{i+1} */
        real rc[3], t[3], nn[3];
        v3sub(rc, K->com[i], K->p[i + 1]);
        v3cross(t, rc, F[i]);
        v3add(nn, N[i], t);
        if (i < 5) {
            real r[3];
            v3sub(r, K->p[i + 2], K->p[i + 1]);
            v3cross(t, r, f);
            v3add(nn, nn, t);
            v3add(nn, nn, n);
        }
        v3add(f, f, F[i]);
        v3copy(n, nn);
        bias[i] = v3dot(K->z[i], n);
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* constraint model (MJ-DOC "soft constraint model", engine_core_constraint semantics)              */
/* ------------------------------------------------------------------------------------------------ */
static double clampd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* impedance d(r) and spring-damper (k,b) from solref/solimp; MJ-DOC "Solver parameters" */
static void kbi(const double *solref, const double *solimp, double pos, double *k, double *b, double *imp) {
    double d0 = clampd(solimp[0], MJ_MINIMP, MJ_MAXIMP), dw = clampd(solimp[1], MJ_MINIMP, MJ_MAXIMP);
    double width = solimp[2] > MJ_MINVAL ? solimp[2] : MJ_MINVAL;
    double mid = clampd(solimp[3], MJ_MINIMP, MJ_MAXIMP), power = solimp[4] < 1 ? 1 : solimp[4];
    *k = 1.0 / (dw * dw * solref[0] * solref[0] * solref[1] * solref[1]);
    *b = 2.0 / (dw * solref[0]);
    if (d0 == dw || width <= MJ_MINVAL) { *imp = 0.5 * (d0 + dw); return; }
    double x = fabs(pos) / width, y;
    if (x > 1) x = 1;
    if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
    else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
    *imp = d0 + y * (dw - d0);
}

/* contact frame from normal (MJ-DOC mju_makeFrame): rows n, t1, t2 */
static void make_frame(real *fr /*[9]*/, const real *n) {
    v3copy(fr, n);
    real y[3] = {0, 0, 0};
    if (n[1] < (real)0.5 && n[1] > (real)-0.5) y[1] = 1; else y[2] = 1;
    real d = v3dot(n, y);
    v3axpy(y, -d, n);
    real l = v3norm(y);
    for (int k = 0; k < 3; k++) fr[3 + k] = y[k] / l;
    v3cross(fr + 6, fr, fr + 3);
}

typedef struct {
    int b1, b2;       /* body ids: -1 world, 0..5 arm link, 6,7 cubes; frame normal points b1 -> b2 */
    real pos[3], frame[9], dist;
    const double *mu; /* [5] tan, tan, torsional, rolling, rolling */
    const double *solimp;
    int slot; /* warm-start slot id: 0-3 floor-cube0, 4-7 floor-cube1, 8-11 cube-cube / rails, 12-13 sphere-cube, 14-15 sphere-floor,
                 16 arm-link proxies */
    int dim;  /* rows: 3 (n, t1, t2), 4 (+ torsion) or 6 (+ rolling about t1, t2: only with orc_params.condim6, D4) */
    int sel;  /* discrete choice behind this contact (which vertex / candidate / face / member): diagnostics, see lag_t.choice */
} contact_t;

#define MAX_CONTACTS (8 + 8 + ORC_MAX_ARM_CONTACTS + NLGRP)
#define MAX_ROWS (12 + 6 * MAX_CONTACTS)
#define ORC_PGS_CAP 50

/* plane z=0 (geom1, world) vs box (geom2): MJ-DOC mjc_PlaneBox -- vertex i=(+-,+-,+-) by bits 0,1,2;
 * contact where vertex height <= 0... margin 0 => strictly below counts (dist<0); at most 4; pos midway */
static int collide_plane_box(const kin_t *K, int c, contact_t *out) {
    int n = 0;
    for (int i = 0; i < 8 && n < 4; i++) {
        real v[3] = {(i & 1) ? (real)CUBE_HALF : (real)-CUBE_HALF, (i & 2) ? (real)CUBE_HALF : (real)-CUBE_HALF,
                     (i & 4) ? (real)CUBE_HALF : (real)-CUBE_HALF}, w[3];
        m3v(w, K->cR[c], v);
        v3add(w, w, K->cp[c]);
        real dist = w[2];
        if (!(dist < 0)) continue;
        contact_t *ct = &out[n++];
        ct->slot = 4 * c + (n - 1);
        ct->b1 = -1; ct->b2 = 6 + c; ct->dist = dist;
        v3set(ct->pos, w[0], w[1], w[2] - dist * (real)0.5);
        real nz[3] = {0, 0, 1};
        make_frame(ct->frame, nz);
        ct->mu = K->mu_cube; ct->solimp = SOLIMP_DEFAULT; /* P9: cube priority 1 beats floor priority 0 */
        ct->dim = 4;
        ct->sel = i;
    }
    return n;
}
/* box c (geom1) vs a sphere (centre, radius) carried by arm link `link` (geom2) -- (D3) */
static int collide_box_sphere_g(const kin_t *K, int c, const real *centre, double radius, int link, contact_t *ct) {
    real d[3], l[3], q[3];
    v3sub(d, centre, K->cp[c]);
    m3tv(l, K->cR[c], d);
    int outside = 0;
    for (int k = 0; k < 3; k++) {
        q[k] = l[k];
        if (l[k] > (real)CUBE_HALF) { q[k] = (real)CUBE_HALF; outside = 1; }
        if (l[k] < (real)-CUBE_HALF) { q[k] = (real)-CUBE_HALF; outside = 1; }
    }
    real nl[3], dist;
    int selcode = 7;
    if (outside) {
        real df[3];
        v3sub(df, l, q);
        real dn = v3norm(df);
        dist = dn - (real)radius;
        if (!(dist < 0)) return 0;
        for (int k = 0; k < 3; k++) nl[k] = df[k] / dn;
    } else {
        int best = 0; real bd = (real)CUBE_HALF - (real)fabs((double)l[0]);
        for (int k = 1; k < 3; k++) { real dk = (real)CUBE_HALF - (real)fabs((double)l[k]); if (dk < bd) { bd = dk; best = k; } }
        real sg = l[best] < 0 ? (real)-1 : (real)1;
        v3set(nl, 0, 0, 0); nl[best] = sg;
        q[best] = sg * (real)CUBE_HALF;
        selcode = 2 * best + (sg < 0 ? 1 : 0);
        dist = -(bd + (real)radius);
    }
    real pl[3] = {q[0] + nl[0] * dist * (real)0.5, q[1] + nl[1] * dist * (real)0.5, q[2] + nl[2] * dist * (real)0.5};
    real nw[3], pw[3];
    m3v(nw, K->cR[c], nl);
    m3v(pw, K->cR[c], pl);
    v3add(ct->pos, pw, K->cp[c]);
    make_frame(ct->frame, nw);
    ct->b1 = 6 + c; ct->b2 = link; ct->dist = dist;
    ct->dim = 4;
    ct->sel = 8 * c + selcode + 16 * ((ct->frame[1] < (real)0.5 && ct->frame[1] > (real)-0.5) ? 0 : 1); /* cube, face case, frame branch */
    return 1;
}
static int collide_box_sphere(const kin_t *K, int c, int s, contact_t *ct) { /* finger sphere s */
    if (!collide_box_sphere_g(K, c, K->sph[s], SPH_RAD[s], SPH_LINK[s], ct)) return 0;
    ct->slot = 12 + s;
    ct->mu = K->mu_finger_cube; ct->solimp = SOLIMP_FINGER_CUBE;
    return 1;
}
/* (round 5: the rails as BOXES for the arm -- VERDICT r3 / r4)  The surface of the world a point p with a margin r (a sphere's radius; 0 for a pad vertex)
 * is deepest inside: the floor (depth r - p_z, normal +z) or one of the four rail boxes of push_cube_loop.xml:44-47, inflated by r -- inside one, the face it is
 * shallowest below: the top (code 1 + 5 b) or a side (+x, -x, +y, -y: codes 2 .. 5 + 5 b), so a finger that comes in sideways at floor height is STOPPED by the rail's
 * side face instead of being lifted onto its top (rounds 2-4 knew the top faces only).  Returns the depth (> 0: penetrating), the outward normal and the code. */
static const double RAIL_C[4][2] = {{-(WALL_X + 0.5 * WALL_THICK), 0.5 * (WALL_Y0 + WALL_Y1)}, {WALL_X + 0.5 * WALL_THICK, 0.5 * (WALL_Y0 + WALL_Y1)},
                                    {0.0, WALL_Y0 - 0.5 * WALL_THICK}, {0.0, WALL_Y1 + 0.5 * WALL_THICK}};
static const double RAIL_H[4][2] = {{0.5 * WALL_THICK, 0.5 * (WALL_Y1 - WALL_Y0) + WALL_THICK}, {0.5 * WALL_THICK, 0.5 * (WALL_Y1 - WALL_Y0) + WALL_THICK},
                                    {WALL_X + 0.5 * WALL_THICK, 0.5 * WALL_THICK}, {WALL_X + 0.5 * WALL_THICK, 0.5 * WALL_THICK}};
static real world_surface(const real *p, real r, int walls, real *n, int *code) {
    real best = r - p[2];
    v3set(n, 0, 0, 1);
    *code = 0;
    if (!walls) return best;
    for (int b = 0; b < 4; b++) {
        const real dx = p[0] - (real)RAIL_C[b][0], dy = p[1] - (real)RAIL_C[b][1];
        const real ex = (real)RAIL_H[b][0] + r - (dx < 0 ? -dx : dx), ey = (real)RAIL_H[b][1] + r - (dy < 0 ? -dy : dy), ez = (real)WALL_TOP + r - p[2];
        if (!(ex > 0 && ey > 0 && ez > 0)) continue;
        real d = ez; int f = 0;
        if (ex < d) { d = ex; f = dx < 0 ? 2 : 1; }
        if (ey < d) { d = ey; f = dy < 0 ? 4 : 3; }
        if (d > best) {
            best = d; *code = 1 + 5 * b + f;
            v3set(n, f == 1 ? (real)1 : (f == 2 ? (real)-1 : 0), f == 3 ? (real)1 : (f == 4 ? (real)-1 : 0), f == 0 ? (real)1 : 0);
        }
    }
    return best;
}
static int collide_plane_sphere_g(const real *centre, double radius, int link, int walls, contact_t *ct) {
    real n[3];
    int code;
    const real depth = world_surface(centre, (real)radius, walls, n, &code);
    if (!(depth > 0)) return 0;
    const real dist = -depth, back = (real)radius + dist * (real)0.5;
    v3set(ct->pos, centre[0] - n[0] * back, centre[1] - n[1] * back, centre[2] - n[2] * back);
    make_frame(ct->frame, n);
    ct->b1 = -1; ct->b2 = link; ct->dist = dist;
    ct->sel = code;   /* which surface: part of the decision signature */
    return 1;
}
/* ---- finger pads as boxes (orc_params.finger_geom = 1).  MuJoCo's convex collider returns ONE contact for a mesh hull against a box, and the deepest points of a
 * hull against a plane; here: vertex-in-box tests both ways (the pad's 8 vertices in the cube, the cube's 8 in the pad; edge-edge crossings without a vertex
 * inside are not seen -- remaining part of deviation D3), the deepest one decides face and normal, and the vertices of the same face within PAD_BLEND of it
 * share the contact point. ---- */
static void pad_vertices(const kin_t *K, int s, real v[8][3], real centre[3]) {
    const real *R = K->R[SPH_LINK[s] + 1];
    real c[3] = {(real)PAD_C[s][0], (real)PAD_C[s][1], (real)PAD_C[s][2]}, t[3];
    m3v(t, R, c);
    v3add(centre, K->p[SPH_LINK[s] + 1], t);
    for (int i = 0; i < 8; i++) {
        real l[3] = {(i & 1) ? (real)PAD_H[s][0] : (real)-PAD_H[s][0], (i & 2) ? (real)PAD_H[s][1] : (real)-PAD_H[s][1], (i & 4) ? (real)PAD_H[s][2] : (real)-PAD_H[s][2]};
        m3v(t, R, l);
        v3add(v[i], centre, t);
    }
}
static int collide_plane_pad(const kin_t *K, int s, int walls, contact_t *ct) {
    real v[8][3], pc[3], depth[8], nn[8][3];
    int code[8];
    pad_vertices(K, s, v, pc);
    int best = 0;
    for (int i = 0; i < 8; i++) {
        depth[i] = world_surface(v[i], 0, walls, nn[i], &code[i]);
        if (depth[i] > depth[best]) best = i;
    }
    if (!(depth[best] > 0)) return 0;
    real wsum = 0, q[3] = {0, 0, 0};
    for (int i = 0; i < 8; i++) {   /* the vertices on the SAME surface within PAD_BLEND of the deepest one share the contact point */
        real w = depth[i] - depth[best] + (real)PAD_BLEND;
        if (!(w > 0) || code[i] != code[best]) continue;
        wsum += w;
        for (int a = 0; a < 3; a++) q[a] += w * v[i][a];
    }
    const real dist = -depth[best];
    const real *n = nn[best];
    /* the blended point, set to the deepest vertex' level along the normal, then midway to the surface */
    real lev = 0, levb = 0;
    for (int a = 0; a < 3; a++) { q[a] /= wsum; lev += q[a] * n[a]; levb += v[best][a] * n[a]; }
    for (int a = 0; a < 3; a++) ct->pos[a] = q[a] + n[a] * (levb - lev - dist * (real)0.5);
    make_frame(ct->frame, n);
    ct->b1 = -1; ct->b2 = SPH_LINK[s]; ct->dist = dist;
    ct->sel = best + 8 * code[best];
    ct->slot = 14 + s;
    ct->mu = MU_FINGER; ct->solimp = SOLIMP_FINGER;
    ct->dim = 4;
    return 1;
}
/* point q (world) inside the box (centre c, rotation R columns = axes, half extents h)?  depth to the nearest face, that face (2 k + (negative side)), local coords */
static int point_in_box(const real *q, const real *c, const real *R, const real *h, real *depth, int *face, real *l) {
    real d[3];
    v3sub(d, q, c);
    m3tv(l, R, d);
    real bd = 0; int bf = -1;
    for (int k = 0; k < 3; k++) {
        const real a = l[k] < 0 ? -l[k] : l[k];
        if (!(a < h[k])) return 0;
        const real dk = h[k] - a;
        if (bf < 0 || dk < bd) { bd = dk; bf = 2 * k + (l[k] < 0 ? 1 : 0); }
    }
    *depth = bd; *face = bf;
    return 1;
}
static int collide_box_pad(const kin_t *K, int c, int s, contact_t *ct) {
    real v[8][3], pc[3], cv[8][3];
    pad_vertices(K, s, v, pc);
    const real *Rp = K->R[SPH_LINK[s] + 1];
    const real hc[3] = {(real)CUBE_HALF, (real)CUBE_HALF, (real)CUBE_HALF}, hp[3] = {(real)PAD_H[s][0], (real)PAD_H[s][1], (real)PAD_H[s][2]};
    for (int i = 0; i < 8; i++) {
        real l[3] = {(i & 1) ? hc[0] : -hc[0], (i & 2) ? hc[1] : -hc[1], (i & 4) ? hc[2] : -hc[2]}, t[3];
        m3v(t, K->cR[c], l);
        v3add(cv[i], K->cp[c], t);
    }
    /* candidates 0-7: pad vertex i inside the cube; 8-15: cube vertex i inside the pad */
    real depth[16], loc[16][3];
    int face[16], in[16], best = -1;
    for (int i = 0; i < 16; i++) {
        in[i] = i < 8 ? point_in_box(v[i], K->cp[c], K->cR[c], hc, &depth[i], &face[i], loc[i]) : point_in_box(cv[i - 8], pc, Rp, hp, &depth[i], &face[i], loc[i]);
        if (in[i] && (best < 0 || depth[i] > depth[best])) best = i;
    }
    if (best < 0) return 0;
    const int typeB = best >= 8, k = face[best] >> 1;
    const real sg = (face[best] & 1) ? (real)-1 : (real)1;
    /* contact point: the blended vertices projected onto the penetrated face, then half the depth back inside */
    real wsum = 0, q[3] = {0, 0, 0};
    for (int i = typeB ? 8 : 0; i < (typeB ? 16 : 8); i++) {
        if (!in[i] || face[i] != face[best]) continue;
        real w = depth[i] - depth[best] + (real)PAD_BLEND;
        if (!(w > 0)) continue;
        wsum += w;
        for (int a = 0; a < 3; a++) q[a] += w * loc[i][a];
    }
    for (int a = 0; a < 3; a++) q[a] /= wsum;
    const real dist = -depth[best];
    q[k] = sg * (typeB ? hp[k] : hc[k]) + sg * dist * (real)0.5;
    real nl[3] = {0, 0, 0}, nw[3], pw[3];
    nl[k] = sg;
    if (!typeB) { m3v(nw, K->cR[c], nl); m3v(pw, K->cR[c], q); v3add(ct->pos, pw, K->cp[c]); }          /* outward normal of the cube's face: cube -> finger */
    else { m3v(nw, Rp, nl); for (int a = 0; a < 3; a++) nw[a] = -nw[a]; m3v(pw, Rp, q); v3add(ct->pos, pw, pc); }   /* outward normal of the pad's face points at the cube */
    make_frame(ct->frame, nw);
    ct->b1 = 6 + c; ct->b2 = SPH_LINK[s]; ct->dist = dist;
    ct->dim = 4;
    ct->sel = c + 2 * ((ct->frame[1] < (real)0.5 && ct->frame[1] > (real)-0.5) ? 0 : 1) + 4 * ((typeB * 6 + face[best]) * 8 + (best & 7));   /* cube, frame branch, (kind, face, vertex) */
    ct->slot = 12 + s;
    ct->mu = K->mu_finger_cube; ct->solimp = SOLIMP_FINGER_CUBE;
    return 1;
}
static int collide_plane_sphere(const kin_t *K, int s, int walls, contact_t *ct) {
    if (!collide_plane_sphere_g(K->sph[s], SPH_RAD[s], SPH_LINK[s], walls, ct)) return 0;
    ct->slot = 14 + s;
    ct->mu = MU_FINGER; ct->solimp = SOLIMP_FINGER; /* P9: finger priority 1 beats floor */
    ct->dim = 4;
    return 1;
}
/* arm-link proxies (group g = 0): the deepest penetration among them against the floor and (gripper body) the cubes */
static int collide_link_group(const kin_t *K, int g, int ngroups, int walls, contact_t *out) {
    int have = 0;
    for (int s = 0; s < NLPX; s++) {
        if ((ngroups == 3 ? LPX_GROUP3[s] : LPX_GROUP[s]) != g) continue;
        contact_t tmp;
        if (collide_plane_sphere_g(K->lpx[s], LPX_RAD[s], LPX_LINK[s], walls, &tmp)) {   /* (D7: floor or a rail's top / side face) */
            tmp.mu = MU_LINK_FLOOR; tmp.solimp = SOLIMP_DEFAULT; tmp.dim = 3;
            tmp.sel += 64 * (s + 1);
            if (!have || tmp.dist < out->dist) { *out = tmp; have = 1; }
        }
        if (LPX_CUBE[s])
            for (int c = 0; c < K->ncube; c++)
                if (collide_box_sphere_g(K, c, K->lpx[s], LPX_RAD[s], LPX_LINK[s], &tmp)) {
                    tmp.mu = K->mu_cube; tmp.solimp = SOLIMP_DEFAULT; tmp.dim = 4; /* P9: cube priority 1 beats the link geoms */
                    tmp.sel += 64 * (s + 1) + 32;
                    if (!have || tmp.dist < out->dist) { *out = tmp; have = 1; }
                }
    }
    if (have) out->slot = g < 2 ? 16 + g : 28;   /* (bits 18-23 of the decision mask are the joint limits, 24-27 the extra cube<->cube points) */
    return have;
}
/* (D7) PushCubeLoop rails (push_cube_loop.xml:44-47): the four wall boxes are restated as their inner faces -- vertical
 * half-spaces that only act below the wall top (z < 0.012) and only while the cube CENTRE is inside the outer rectangle of the rails
 * (inner faces + box thickness 0.02; round 3: before, a cube knocked over a rail was "deep inside" a half-space -- 0.5 % of the
 * env-states of a random-policy run, with cube speeds up to 1 200 m/s in the tail; now it rests outside the pen as next to the reference's boxes).
 * The pen is wider than the cube in x and in y, so the cube
 * can reach at most one x rail and one y rail: contact slots 8,9 belong to the x pair (the x = -0.115 rail if any vertex
 * is beyond it, else the x = +0.115 rail), slots 10,11 to the y pair (y = 0.10 rail if touched, else y = 0.17); each
 * pair keeps its two deepest vertices (ties: lower vertex index first).  Frame normal points from the wall into the pen. */
static int cube_in_pen(const real *c) {   /* cube centre inside the outer rectangle of the rails (inner faces + box thickness) */
    return c[0] < (real)WALL_X + (real)WALL_THICK && -c[0] < (real)WALL_X + (real)WALL_THICK
        && c[1] > (real)WALL_Y0 - (real)WALL_THICK && c[1] < (real)WALL_Y1 + (real)WALL_THICK;
}
static int collide_walls(const kin_t *K, contact_t *out) {
    int n = 0;
    if (!cube_in_pen(K->cp[0])) return 0;
    real P[8][3];
    for (int i = 0; i < 8; i++) {
        real v[3] = {(i & 1) ? (real)CUBE_HALF : (real)-CUBE_HALF, (i & 2) ? (real)CUBE_HALF : (real)-CUBE_HALF,
                     (i & 4) ? (real)CUBE_HALF : (real)-CUBE_HALF};
        m3v(P[i], K->cR[0], v);
        v3add(P[i], P[i], K->cp[0]);
    }
    for (int pr = 0; pr < 2; pr++) {
        /* which rail of the pair: the low-coordinate one if any vertex is beyond it */
        int lo = 0;
        for (int i = 0; i < 8; i++) {
            const real *p = P[i];
            real dist = pr == 0 ? p[0] + (real)WALL_X : p[1] - (real)WALL_Y0;
            if (dist < 0 && p[2] < (real)WALL_TOP) lo = 1;
        }
        int w = 2 * pr + (lo ? 0 : 1);
        real nw[3] = {0, 0, 0};
        if (w == 0) nw[0] = 1; else if (w == 1) nw[0] = -1; else if (w == 2) nw[1] = 1; else nw[1] = -1;
        real d1 = 0, d2 = 0; int i1 = -1, i2 = -1;
        for (int i = 0; i < 8; i++) {
            const real *p = P[i];
            real dist = w == 0 ? p[0] + (real)WALL_X : (w == 1 ? (real)WALL_X - p[0] : (w == 2 ? p[1] - (real)WALL_Y0 : (real)WALL_Y1 - p[1]));
            if (!(dist < 0) || !(p[2] < (real)WALL_TOP)) continue;
            if (i1 < 0 || dist < d1) { d2 = d1; i2 = i1; d1 = dist; i1 = i; }
            else if (i2 < 0 || dist < d2) { d2 = dist; i2 = i; }
        }
        for (int c = 0; c < 2; c++) {
            int i = c == 0 ? i1 : i2;
            real dist = c == 0 ? d1 : d2;
            if (i < 0) continue;
            contact_t *ct = &out[n];
            ct->slot = 8 + 2 * pr + c;
            n++;
            /* (round 5) a rail sees at most RAIL_CLAMP of penetration: a cube whose centre has just come back inside the outer rectangle is 2 cm "deep" in the
             * half-space at once -- it is pushed in as by a rail it has just touched, not shot in */
            if (dist < (real)-RAIL_CLAMP) dist = (real)-RAIL_CLAMP;
            ct->b1 = -1; ct->b2 = 6; ct->dist = dist;
            v3set(ct->pos, P[i][0] - nw[0] * dist * (real)0.5, P[i][1] - nw[1] * dist * (real)0.5, P[i][2]);
            make_frame(ct->frame, nw);
            ct->mu = K->mu_cube; ct->solimp = SOLIMP_DEFAULT;
            ct->dim = 4;
            ct->sel = i + 8 * w;
        }
    }
    return n;
}

/* (D5) box0 (geom1) vs box1 (geom2).  Restated manifold (own construction, MuJoCo's mjc_BoxBox is not available):
 *  1. separating-axis test over the 6 face normals; the axis of least overlap gives the contact normal n (0 -> 1),
 *     the reference box A (owner of that axis) and the incident box B;
 *  2. B's incident face = the face of B most anti-parallel to m (m = normal pointing A -> B);
 *  3. candidate points of the overlap polygon of B's incident face with A's reference face:
 *       (a) the 4 vertices of B's incident face that lie inside A's face footprint and below A's face,
 *       (b) the 4 vertices of A's face that lie inside B's incident face (only for near-parallel faces),
 *       (c) the <= 16 crossings of B's face edges with A's face boundary lines, below A's face;
 *  4. at most 4 are kept: the extreme candidates along the 4 diagonals (+-u +-v) of A's face frame, first-wins on
 *     ties, duplicates dropped.  Each contact: position midway between the two surfaces, dist < 0 along n. */
static int collide_box_box(const kin_t *K, contact_t *out) {
    const real h = (real)CUBE_HALF;
    const real tol = (real)1e-4;
    real dc[3];
    v3sub(dc, K->cp[1], K->cp[0]);
    real best = (real)-1e30; int bax = -1; real bsgn = 1;
    for (int ax = 0; ax < 6; ax++) {
        int bx = ax / 3, k = ax % 3;
        real n[3] = {K->cR[bx][k], K->cR[bx][3 + k], K->cR[bx][6 + k]};
        int ob = 1 - bx;
        real ext = 0;
        for (int j = 0; j < 3; j++) {
            real col[3] = {K->cR[ob][j], K->cR[ob][3 + j], K->cR[ob][6 + j]};
            ext += (real)fabs((double)v3dot(n, col)) * h;
        }
        real dd = v3dot(n, dc);
        real sep = (real)fabs((double)dd) - h - ext;
        if (sep > best) { best = sep; bax = ax; bsgn = dd < 0 ? (real)-1 : (real)1; }
    }
    if (!(best < 0)) return 0;
    const int A = bax / 3, B = 1 - A, k = bax % 3;
    real n[3] = {K->cR[A][k] * bsgn, K->cR[A][3 + k] * bsgn, K->cR[A][6 + k] * bsgn}; /* box0 -> box1 */
    real m[3] = {A == 0 ? n[0] : -n[0], A == 0 ? n[1] : -n[1], A == 0 ? n[2] : -n[2]}; /* A -> B */
    const int ku = (k + 1) % 3, kv = (k + 2) % 3;
    real u[3] = {K->cR[A][ku], K->cR[A][3 + ku], K->cR[A][6 + ku]}, v[3] = {K->cR[A][kv], K->cR[A][3 + kv], K->cR[A][6 + kv]};
    /* incident face of B */
    int kb = 0; real bestdot = -1;
    real mdot[3];
    for (int j = 0; j < 3; j++) {
        real col[3] = {K->cR[B][j], K->cR[B][3 + j], K->cR[B][6 + j]};
        mdot[j] = v3dot(m, col);
        if ((real)fabs((double)mdot[j]) > bestdot) { bestdot = (real)fabs((double)mdot[j]); kb = j; }
    }
    const real sB = mdot[kb] > 0 ? (real)-1 : (real)1;
    real nb[3] = {sB * K->cR[B][kb], sB * K->cR[B][3 + kb], sB * K->cR[B][6 + kb]};
    const int kp = (kb + 1) % 3, kq = (kb + 2) % 3;
    real pa[3] = {K->cR[B][kp], K->cR[B][3 + kp], K->cR[B][6 + kp]}, qa[3] = {K->cR[B][kq], K->cR[B][3 + kq], K->cR[B][6 + kq]};
    real fB[3] = {K->cp[B][0] + h * nb[0], K->cp[B][1] + h * nb[1], K->cp[B][2] + h * nb[2]};
    const real mnb = v3dot(m, nb); /* ~ -1 for parallel faces */
    static const real SP[4] = {1, -1, -1, 1}, SQ[4] = {1, 1, -1, -1};
    real V[4][3], Vu[4], Vv[4], Vd[4];
    for (int i = 0; i < 4; i++) {
        for (int c = 0; c < 3; c++) V[i][c] = fB[c] + h * (SP[i] * pa[c] + SQ[i] * qa[c]);
        real d[3];
        v3sub(d, V[i], K->cp[A]);
        Vu[i] = v3dot(d, u); Vv[i] = v3dot(d, v); Vd[i] = v3dot(d, m) - h;
    }
    /* selection state: 4 diagonal-extreme slots (+ 4 axis-extreme slots when cc_points == 8: study of D5) */
    const int nsel = K->cc_points == 8 ? 8 : 4;
    real spos[8][3], sdist[8], skey[8];
    int sidx[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
    int cand = 0;
#define CONSIDER(PX, PY, PZ, DIST, CU, CV)                                                        \
    do {                                                                                          \
        real key_[8] = {(CU) + (CV), -(CU) + (CV), -(CU) - (CV), (CU) - (CV), (CU), (CV), -(CU), -(CV)}; \
        for (int s_ = 0; s_ < nsel; s_++)                                                         \
            if (sidx[s_] < 0 || key_[s_] > skey[s_]) {                                            \
                skey[s_] = key_[s_]; sidx[s_] = cand; sdist[s_] = (DIST);                          \
                spos[s_][0] = (PX); spos[s_][1] = (PY); spos[s_][2] = (PZ);                        \
            }                                                                                     \
    } while (0)
    /* (a) vertices of B's incident face */
    for (int i = 0; i < 4; i++, cand++) {
        if ((real)fabs((double)Vu[i]) > h + tol || (real)fabs((double)Vv[i]) > h + tol || !(Vd[i] < 0)) continue;
        real hd = (real)0.5 * Vd[i];
        CONSIDER(V[i][0] - m[0] * hd, V[i][1] - m[1] * hd, V[i][2] - m[2] * hd, Vd[i], Vu[i], Vv[i]);
    }
    /* (b) vertices of A's reference face, near-parallel faces only */
    for (int j = 0; j < 4; j++, cand++) {
        if (!(mnb < (real)-0.5)) continue;
        real a[3], d[3];
        for (int c = 0; c < 3; c++) a[c] = K->cp[A][c] + h * m[c] + h * (SP[j] * u[c] + SQ[j] * v[c]);
        v3sub(d, a, fB);
        if ((real)fabs((double)v3dot(d, pa)) > h + tol || (real)fabs((double)v3dot(d, qa)) > h + tol) continue;
        real t = -v3dot(d, nb) / mnb;
        if (!(t < 0)) continue;
        real ht = (real)0.5 * t;
        CONSIDER(a[0] + m[0] * ht, a[1] + m[1] * ht, a[2] + m[2] * ht, t, SP[j] * h, SQ[j] * h);
    }
    /* (c) crossings of B's face edges with A's face boundary lines */
    for (int e = 0; e < 4; e++) {
        const int e2 = (e + 1) & 3;
        for (int l = 0; l < 4; l++, cand++) {
            const int on_u = l < 2; /* lines u = +-h then v = +-h */
            const real sg = (l & 1) ? (real)-1 : (real)1;
            const real cP = on_u ? Vu[e] : Vv[e], cQ = on_u ? Vu[e2] : Vv[e2];
            const real oP = on_u ? Vv[e] : Vu[e], oQ = on_u ? Vv[e2] : Vu[e2];
            const real fP = cP - sg * h, fQ = cQ - sg * h;
            if (!((fP < 0 && fQ > 0) || (fP > 0 && fQ < 0))) continue;
            const real t = fP / (fP - fQ);
            const real ot = oP + t * (oQ - oP);
            if ((real)fabs((double)ot) > h + tol) continue;
            const real d = Vd[e] + t * (Vd[e2] - Vd[e]);
            if (!(d < 0)) continue;
            real X[3];
            for (int c = 0; c < 3; c++) X[c] = V[e][c] + t * (V[e2][c] - V[e][c]);
            real hd = (real)0.5 * d;
            CONSIDER(X[0] - m[0] * hd, X[1] - m[1] * hd, X[2] - m[2] * hd, d, on_u ? sg * h : ot, on_u ? ot : sg * h);
        }
    }
#undef CONSIDER
    int cnt = 0;
    for (int s = 0; s < nsel; s++) {
        if (sidx[s] < 0) continue;
        int dup = 0;
        for (int s2 = 0; s2 < s; s2++) if (sidx[s2] == sidx[s]) dup = 1;
        if (dup) continue;
        contact_t *ct = &out[cnt];
        ct->slot = s < 4 ? 8 + s : 24 + (s - 4);
        ct->b1 = 6; ct->b2 = 7; ct->dist = sdist[s];
        v3copy(ct->pos, spos[s]);
        make_frame(ct->frame, n);
        ct->mu = MU_CUBE; ct->solimp = SOLIMP_DEFAULT;
        ct->dim = 4;
        ct->sel = sidx[s] + 32 * (bax + 6 * kb) + 1024 * (bsgn < 0 ? 1 : 0);
        cnt++;
    }
    return cnt;
}

/* precomputed at qpos0 (MJ-DOC body_invweight0 / dof_invweight0, used by diagApprox) */
static double g_inv_tran[6], g_inv_rot[6], g_inv_dof[6];
static int g_inv_ready = 0;
static void ensure_invweight0(void) {
    if (g_inv_ready) return;
#ifdef _OPENMP
#pragma omp critical(orc_invw)
#endif
    {
        if (!g_inv_ready) {
            real q[6] = {0, 0, 0, 0, 0, 0}, M[36], L[36];
            kin_t K;
            arm_kinematics(q, &K);
            arm_mass(&K, 1, M);
            memcpy(L, M, sizeof L);
            chol(L, 6);
            for (int j = 0; j < 6; j++) {
                real e[6] = {0, 0, 0, 0, 0, 0};
                e[j] = 1;
                chol_solve(L, 6, e);
                g_inv_dof[j] = (double)e[j];
            }
            for (int b = 0; b < 6; b++) {
                real Jp[18], Jr[18];
                jac_point(&K, 6, b, K.com[b], Jp, Jr);
                double tr = 0, rr = 0;
                for (int k = 0; k < 3; k++) {
                    real x[6], y[6];
                    for (int j = 0; j < 6; j++) { x[j] = Jp[k * 6 + j]; y[j] = Jr[k * 6 + j]; }
                    real x0[6], y0[6];
                    memcpy(x0, x, sizeof x); memcpy(y0, y, sizeof y);
                    chol_solve(L, 6, x); chol_solve(L, 6, y);
                    for (int j = 0; j < 6; j++) { tr += (double)(x0[j] * x[j]); rr += (double)(y0[j] * y[j]); }
                }
                g_inv_tran[b] = tr / 3; g_inv_rot[b] = rr / 3;
            }
            g_inv_ready = 1;
        }
    }
}
static void body_invweight(const task_model *T, int b, double *tr, double *ro) {
    if (b < 0) { *tr = 0; *ro = 0; }
    else if (b < 6) { *tr = g_inv_tran[b]; *ro = g_inv_rot[b]; }
    else { *tr = 1.0 / T->cube_mass; *ro = 1.0 / T->cube_inertia; }
}

static int g_diag_rows, g_diag_contacts;
static double g_diag_res;

/* ---- yard-stick of the contact solver (orc_params.cone = 1, orc_io.kkt): MuJoCo's per-contact PGS block update and the KKT certificate ---- */
/* min 1/2 y'Ay + y'b  s.t. |y| <= r  (n <= 5): Newton on the multiplier of the norm constraint, MJ-DOC mju_QCQP after scaling by the friction coefficients */
static void qcqp_ball(int n, const double *A, const double *b, double r, double *y) {
    double lam = 0, L[25], x[5];
    for (int it = 0; it < 60; it++) {
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) L[i * n + j] = A[i * n + j] + (i == j ? lam : 0.0);
        /* Cholesky (A is positive definite: a diagonal block of A + R) */
        for (int j = 0; j < n; j++) {
            double d = L[j * n + j];
            for (int k = 0; k < j; k++) d -= L[j * n + k] * L[j * n + k];
            d = d > 1e-300 ? sqrt(d) : 1e-150;
            L[j * n + j] = d;
            for (int i = j + 1; i < n; i++) {
                double s = L[i * n + j];
                for (int k = 0; k < j; k++) s -= L[i * n + k] * L[j * n + k];
                L[i * n + j] = s / d;
            }
        }
        /* y = -(A + lam I)^-1 b */
        for (int i = 0; i < n; i++) { double s = -b[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * y[k]; y[i] = s / L[i * n + i]; }
        for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < n; k++) s -= L[k * n + i] * y[k]; y[i] = s / L[i * n + i]; }
        double val = -r * r;
        for (int i = 0; i < n; i++) val += y[i] * y[i];
        if (val <= 1e-14 * (1.0 + r * r)) break;                 /* inside the ball (lam = 0) or on it */
        /* d|y|^2 / dlam = -2 y'(A + lam I)^-1 y */
        for (int i = 0; i < n; i++) { double s = y[i]; for (int k = 0; k < i; k++) s -= L[i * n + k] * x[k]; x[i] = s / L[i * n + i]; }
        double der = 0;
        for (int i = 0; i < n; i++) der += x[i] * x[i];          /* = y'(LL')^-1 y */
        der *= -2.0;
        const double delta = -val / der;
        if (!(delta > 0) || delta < 1e-16 * (1.0 + lam)) break;
        lam += delta;
    }
}
static void pgs_block_exact(const real *A, int nr, const real *bvec, const real *Rr, real *f, int i0, int dm, const double *mu) {
    double AR[36], res[6], old[6];
    for (int j = 0; j < dm; j++) {
        double s = (double)bvec[i0 + j] + (double)Rr[i0 + j] * (double)f[i0 + j];
        for (int m = 0; m < nr; m++) s += (double)A[(size_t)(i0 + j) * nr + m] * (double)f[m];
        res[j] = s; old[j] = (double)f[i0 + j];
        for (int k = 0; k < dm; k++) AR[j * dm + k] = (double)A[(size_t)(i0 + j) * nr + i0 + k] + (j == k ? (double)Rr[i0 + j] : 0.0);
    }
    double fn;
    double cur[6];
    for (int j = 0; j < dm; j++) cur[j] = old[j];
    if (old[0] < MJ_MINVAL) {
        /* The block sits at the apex of its cone.  Zero is the block's optimum iff its gradient c = res - AR old lies in the dual cone, in scaled variables
         * (f~ = (f_n, f_j / mu_j), c~ = (c_n, mu_j c_j)):  |c~_t| <= c~_n.  MuJoCo's PGS only tries the normal row here and therefore stays at zero whenever the
         * normal row alone does not want a force -- although with mu > 1 the convex problem (and MuJoCo's default Newton solver) may have a non-zero optimum
         * there.  To reach THAT optimum: otherwise step along the projection of -c~ onto the cone with an exact line search, then continue with the friction QCQP. */
        double c[6], ct[6], d[6], xn = 0;
        for (int j = 0; j < dm; j++) { c[j] = res[j]; for (int k = 0; k < dm; k++) c[j] -= AR[j * dm + k] * old[k]; }
        ct[0] = c[0];
        for (int j = 1; j < dm; j++) { ct[j] = mu[j - 1] * c[j]; xn += ct[j] * ct[j]; }
        xn = sqrt(xn);
        if (xn <= ct[0]) { for (int j = 0; j < dm; j++) f[i0 + j] = 0; return; }           /* zero is optimal for this block */
        /* P_soc(-c~) */
        if (xn <= -ct[0]) { for (int j = 0; j < dm; j++) d[j] = -ct[j]; }
        else { const double a = 0.5 * (-ct[0] + xn); d[0] = a; for (int j = 1; j < dm; j++) d[j] = -a * ct[j] / xn; }
        for (int j = 1; j < dm; j++) d[j] *= mu[j - 1];                                       /* back to force units */
        double num = 0, den = 0;
        for (int j = 0; j < dm; j++) { num += d[j] * c[j]; for (int k = 0; k < dm; k++) den += d[j] * AR[j * dm + k] * d[k]; }
        const double al = den > 0 ? -num / den : 0.0;
        for (int j = 0; j < dm; j++) cur[j] = al > 0 ? al * d[j] : 0.0;
        fn = cur[0];
    } else {                            /* ray update: exact line search along the current force direction (stays inside the cone) */
        double den = 0, num = 0;
        for (int j = 0; j < dm; j++) { num += old[j] * res[j]; for (int k = 0; k < dm; k++) den += old[j] * AR[j * dm + k] * old[k]; }
        if (den >= MJ_MINVAL) {
            double x = -num / den;
            if (old[0] + x * old[0] < 0) x = -1.0;
            for (int j = 0; j < dm; j++) cur[j] = old[j] + x * old[j];
        }
        fn = cur[0];
    }
    if (fn < MJ_MINVAL) { for (int j = 0; j < dm; j++) f[i0 + j] = (real)(j == 0 ? (fn > 0 ? fn : 0) : 0); return; }
    /* friction rows given the normal force: gradient at friction x is  bc + Ac x,  bc_j = res_j - sum_k AR_jk old_k + AR_j0 fn  (k over all rows of the block) */
    const int n = dm - 1;
    double Ac[25], bc[5], y[5];
    for (int j = 0; j < n; j++) {
        double s = res[j + 1] + AR[(j + 1) * dm] * fn;
        for (int k = 0; k < dm; k++) s -= AR[(j + 1) * dm + k] * old[k];
        bc[j] = s * mu[j];
        for (int k = 0; k < n; k++) Ac[j * n + k] = AR[(j + 1) * dm + k + 1] * mu[j] * mu[k];
    }
    qcqp_ball(n, Ac, bc, fn, y);
    f[i0] = (real)fn;
    for (int j = 0; j < n; j++) f[i0 + 1 + j] = (real)(y[j] * mu[j]);
}
/* A contact block at the apex of its cone (all its forces zero): is zero the block's optimum?  If not, step along the projection of the negative
 * gradient onto the cone with an exact line search (see pgs_block_exact).  Returns 1 if the block moved.  Used by cone = 2: the row-by-row sweep with
 * radial projection (D2) plus this escape, which removes the false fixed point at the apex while keeping the product's cheap row updates. */
static int pgs_apex_escape(const real *A, int nr, const real *bvec, const real *Rr, real *f, int i0, int dm, const double *mu) {
    double c[6], ct[6], d[6], xn = 0;
    for (int j = 0; j < dm; j++) {
        double s = (double)bvec[i0 + j];
        for (int m = 0; m < nr; m++) s += (double)A[(size_t)(i0 + j) * nr + m] * (double)f[m];
        c[j] = s;
    }
    ct[0] = c[0];
    for (int j = 1; j < dm; j++) { ct[j] = mu[j - 1] * c[j]; xn += ct[j] * ct[j]; }
    xn = sqrt(xn);
    if (xn <= ct[0]) return 0;
    if (xn <= -ct[0]) { for (int j = 0; j < dm; j++) d[j] = -ct[j]; }
    else { const double a = 0.5 * (-ct[0] + xn); d[0] = a; for (int j = 1; j < dm; j++) d[j] = -a * ct[j] / xn; }
    for (int j = 1; j < dm; j++) d[j] *= mu[j - 1];
    double num = 0, den = 0;
    for (int j = 0; j < dm; j++) {
        num += d[j] * c[j];
        for (int k = 0; k < dm; k++) den += d[j] * ((double)A[(size_t)(i0 + j) * nr + i0 + k] + (j == k ? (double)Rr[i0 + j] : 0.0)) * d[k];
    }
    const double al = den > 0 ? -num / den : 0.0;
    if (!(al > 0)) return 0;
    for (int j = 0; j < dm; j++) f[i0 + j] = (real)(al * d[j]);
    return 1;
}
/* cone = 3: one projected-gradient step on the whole contact block, in the scaled variables y = (f_n, f_j / mu_j) in which the elliptic cone is the
 * second-order cone: y <- P_soc(y - g~ / Lb), g~ = (g_n, mu_j g_j) the block's gradient at the current forces and Lb = trace of the scaled block of A + R (an
 * upper bound of its largest eigenvalue, so the step never overshoots).  Fixed points are exactly the optima of the convex problem (no apex trap, no radial
 * bias), the cost per sweep is that of the row-by-row update, and all rows of a block are evaluated from the same forces (no serial dependence inside it). */
static void pgs_block_pg(const real *A, int nr, const real *bvec, const real *Rr, real *f, const real *fsee, int i0, int dm, const double *mu, int scalar_step, int sep) {
    /* Projected-gradient step in the metric D = diag(Ln, Lt, .., Lt) of the scaled variables y = (f_n, f_j / mu_j):  y <- P_K^D(y - D^-1 g~).
     * Ln = 2 (A + R)_nn and Lt = 2 sum_j mu_j^2 (A + R)_jj majorise the scaled block (a PSD matrix is below k times its block diagonal for k diagonal blocks,
     * and a PSD block below its trace), so the step never increases the objective; its fixed points are exactly the optima.  The D-projection onto the
     * second-order cone is closed-form for this two-group metric: with v = y - D^-1 g~, N = |v_t|:
     *     inside (N <= v_n): v;   else  y_n = max(0, w v_n + (1 - w) N),  y_t = v_t y_n / N,  w = Ln / (Ln + Lt).
     * In force units:  v_n = f_n - u_n / Ln,  v_j mu_j = f_j - mu_j^2 u_j / Lt  (u = gradient rows).
     * sep (study only: PushCubeLoop's default is cone = 0.  Its cube has torsional and rolling coefficients of 1.5 m, push_cube_loop.xml:31: their scaled
     * curvature mu^2 / I is four orders of magnitude above the tangential rows' and would set the step of all friction rows): the torsional and rolling rows
     * form a THIRD group with their own Ls (all three scaled by 3 instead of 2) and the D-projection is the exact one for three weights (one-dimensional
     * root, below).  Converges to the optimum, but slowly: four sweeps leave a pinched cube's normal forces so far off that cubes are thrown
     * (tools/loop_solver_study.py) -- which is why the loop task keeps the row-wise sweeps. */
    real u[6], Ln = 0, Lt = 0, Ls = 0;
    const real kf = sep ? 3 : 2;
    for (int j = 0; j < dm; j++) {
        real s = bvec[i0 + j] + Rr[i0 + j] * f[i0 + j];
        for (int m = 0; m < nr; m++) s += A[(size_t)(i0 + j) * nr + m] * fsee[m];
        u[j] = s;
        const real d = A[(size_t)(i0 + j) * nr + i0 + j] + Rr[i0 + j];
        if (j == 0) Ln = kf * d;
        else if (sep && j >= 3) Ls += kf * (real)(mu[j - 1] * mu[j - 1]) * d;
        else Lt += kf * (real)(mu[j - 1] * mu[j - 1]) * d;
    }
    if (!sep && scalar_step == 2) {   /* (study, cone = 5) factor 1 + rho instead of the worst case 2: D = (1 + rho) diag(a, T, .., T) majorises the scaled block as soon as rho^2 >= |b|^2 / (a T), b_j = mu_j A_nj
                   * (Schur complement of [[rho a, -b'], [-b, rho T I]]); rho <= 1 for a positive semi-definite block.  Measured: 6 x closer to the optimum at
                   * the 90th percentile, but in the kernels 3 % slower and with a heavier extreme tail for Lift (a cube 36 mm under the floor in 5.6e6 states): not adopted. */
        real b2 = 0;
        for (int j = 1; j < dm; j++) { const real a = A[(size_t)i0 * nr + i0 + j]; b2 += (real)(mu[j - 1] * mu[j - 1]) * a * a; }
        const real ann = Ln / kf, trt = Lt / kf;
        real rho = trt > 0 ? (real)sqrt((double)(b2 / (ann * trt))) : 0;
        if (rho > 1) rho = 1;
        Ln = ((real)1 + rho) * ann; Lt = ((real)1 + rho) * trt;
    }
    if (scalar_step == 1) { Ln = (real)0.5 * (Ln + Lt); Lt = Ln; }   /* (study, cone = 4) one scalar step 1 / trace for the whole block */
    const real iLn = (real)1 / Ln, iLt = (real)1 / Lt, iLs = Ls > 0 ? (real)1 / Ls : 0, w = Ln / (Ln + Lt);
    real fp[6], s2 = 0, s2s = 0;
    fp[0] = f[i0] - u[0] * iLn;
    for (int j = 1; j < dm; j++) {
        const real m2 = (real)(mu[j - 1] * mu[j - 1]);
        if (sep && j >= 3) { fp[j] = f[i0 + j] - m2 * u[j] * iLs; s2s += fp[j] * fp[j] / m2; }
        else { fp[j] = f[i0 + j] - m2 * u[j] * iLt; s2 += fp[j] * fp[j] / m2; }
    }
    const real N = (real)sqrt((double)s2);
    if (sep && Ls > 0) {
        /* exact D-projection onto the cone for three weights: stationarity gives  y_n = Ln p_n / (Ln - lam),  y_t = p_t Lt / (Lt + lam),  y_s = p_s Ls / (Ls + lam)
         * and lam >= 0 is the root of  phi(lam) = (Ln - lam) r(lam) - Ln p_n,  r = sqrt((Lt T / (Lt + lam))^2 + (Ls S / (Ls + lam))^2)  (decreasing; phi(0) > 0
         * outside the cone, phi(inf) < 0 outside the polar cone). */
        const real S = (real)sqrt((double)s2s), p0 = fp[0];
        real y0, sct, scs2;
        if (p0 >= 0 && s2 + s2s <= p0 * p0) { y0 = p0; sct = 1; scs2 = 1; }
        else if (Ln * p0 <= -(real)sqrt((double)(Lt * Lt * s2 + Ls * Ls * s2s))) { y0 = 0; sct = 0; scs2 = 0; }
        else {
            real lam = 0;
            for (int it = 0; it < 60; it++) {
                const real at = Lt / (Lt + lam), as = Ls / (Ls + lam);
                const real r2 = at * at * s2 + as * as * s2s, r = (real)sqrt((double)r2);
                const real phi = (Ln - lam) * r - Ln * p0;
                /* r' = -(at^2 s2 / (Lt + lam) + as^2 s2s / (Ls + lam)) / r */
                const real dr = -(at * at * s2 / (Lt + lam) + as * as * s2s / (Ls + lam)) / r;
                const real dphi = -r + (Ln - lam) * dr;
                real step = phi / dphi;
                lam -= step;
                if (lam < 0) lam = 0;
                if (fabs((double)step) <= 1e-14 * (double)(Ln + lam)) break;
            }
            sct = Lt / (Lt + lam); scs2 = Ls / (Ls + lam);
            y0 = (real)sqrt((double)(sct * sct * s2 + scs2 * scs2 * s2s));
        }
        f[i0] = y0;
        for (int j = 1; j < dm; j++) f[i0 + j] = fp[j] * (j >= 3 ? scs2 : sct);
        (void)S; (void)N; (void)w;
        return;
    }
    real a = w * fp[0] + ((real)1 - w) * N, y0 = fp[0];
    if (a > y0) y0 = a;
    if (y0 < 0) y0 = 0;
    real sc = 1;
    if (N > y0) sc = y0 / N;
    f[i0] = y0;
    for (int j = 1; j < dm; j++) f[i0 + j] = fp[j] * sc;
}
/* natural residual of the KKT conditions of  min_{f in K} 1/2 f'(A + R) f + f'b  (orc_io.kkt) */
static double kkt_residual(const real *A, int nr, const real *bvec, const real *Rr, const real *f, const int *kind, const int *blkdim, const double *const *rowmu) {
    double worst = 0, fmax = 0;
    double g[MAX_ROWS];
    for (int i = 0; i < nr; i++) {
        double s = (double)bvec[i] + (double)Rr[i] * (double)f[i];
        for (int j = 0; j < nr; j++) s += (double)A[(size_t)i * nr + j] * (double)f[j];
        g[i] = s;
        if (fabs((double)f[i]) > fmax) fmax = fabs((double)f[i]);
    }
    for (int i = 0; i < nr; i++) {
        if (kind[i] == 0) {             /* limit: f >= 0, g >= 0, f g = 0 */
            const double r = fabs((double)f[i] < g[i] ? (double)f[i] : g[i]);
            if (r > worst) worst = r;
        } else if (kind[i] == 1) {      /* contact block: z = f~ - g~ projected onto the second-order cone */
            const int dm = blkdim[i];
            const double *mu = rowmu[i];
            double ft[6], z[6];
            ft[0] = (double)f[i]; z[0] = ft[0] - g[i];
            double xn = 0;
            for (int r = 1; r < dm; r++) { ft[r] = (double)f[i + r] / mu[r - 1]; z[r] = ft[r] - mu[r - 1] * g[i + r]; xn += z[r] * z[r]; }
            xn = sqrt(xn);
            double p[6];
            if (xn <= z[0]) { for (int r = 0; r < dm; r++) p[r] = z[r]; }
            else if (xn <= -z[0]) { for (int r = 0; r < dm; r++) p[r] = 0; }
            else { const double a = 0.5 * (z[0] + xn); p[0] = a; for (int r = 1; r < dm; r++) p[r] = a * z[r] / xn; }
            for (int r = 0; r < dm; r++) { const double d = fabs(ft[r] - p[r]); if (d > worst) worst = d;
                if (getenv("ORC_KKT_DEBUG") && d > 1.0) fprintf(stderr, "kkt blk row0=%d dm=%d r=%d d=%g f=%g g=%g ft=%g p=%g mu=%g fn=%g gn=%g\n", i, dm, r, d, (double)f[i + r], g[i + r], ft[r], p[r], r ? mu[r - 1] : 1.0, (double)f[i], g[i]); }
        }
    }
    return worst / (1.0 + fmax);
}

/* ---- exact optimum of MuJoCo's convex constraint problem (orc_params.solver = 1): Newton's method on the PRIMAL, as MuJoCo's default solver --------------
 * (follower.xml:3 sets no solver -> Newton).  MJ-DOC "Computation / Constraint model": the constrained acceleration minimises
 *     F(x) = 1/2 (x - a0)' M (x - a0) + sum_b s_b(J_b x - aref_b),     s_b(z) = max_{f in K_b} ( -f'z - 1/2 f'R_b f ),
 * the Fenchel dual of  min_{f in K} 1/2 f'(A + R) f + f'(J a0 - aref)  (what PGS iterates on); the constraint forces are f_b = argmax.  Because the friction
 * rows are regularised by R_f mu_0^2 / mu_j^2 the block maximiser is closed-form in the scaled variables y = (f_n, f_j / mu_j), w = (z_n, mu_j z_j), N = |w_t|:
 *     top    (w_n >= N):                       y = 0
 *     bottom (N / Rt <= -w_n / Rn):            y_n = -w_n / Rn,  y_t = -w_t / Rt                     (inside the cone: plain quadratic)
 *     middle (otherwise):                      y_n = (N - w_n) / (Rn + Rt),  y_t = -y_n w_t / N       (on the cone's surface)
 * with Rn = R of the normal row, Rt = R_f mu_0^2.  F is convex, C^1 and piecewise quadratic in nv <= 18 unknowns; Newton with a backtracking line search
 * converges in a handful of iterations.  The result is certified independently by kkt_residual() on the DUAL problem. */
typedef struct { int i0, dm; double Rn, Rt; const double *mu; } nblock;
static double newton_eval(int nv, int nr, const double *Md, const double *Jd, const double *aref, const double *Rr, const double *a0, const int *kind,
                          const int *blkdim, const double *const *rowmu, const double *x, double *f, double *W /* nr x 6: per-row block of -df/dz, or NULL */) {
    double cost = 0;
    for (int i = 0; i < nv; i++) {
        double s = 0;
        for (int j = 0; j < nv; j++) s += Md[i * nv + j] * (x[j] - a0[j]);
        cost += 0.5 * (x[i] - a0[i]) * s;
    }
    for (int i = 0; i < nr; i++) {
        if (kind[i] == 2) continue;
        const int dm = kind[i] == 0 ? 1 : blkdim[i];
        double z[6] = {0, 0, 0, 0, 0, 0};
        for (int r = 0; r < dm; r++) {
            double s = -aref[i + r];
            for (int d = 0; d < nv; d++) s += Jd[(size_t)(i + r) * nv + d] * x[d];
            z[r] = s;
        }
        if (W) for (int r = 0; r < dm; r++) for (int c = 0; c < 6; c++) W[(size_t)(i + r) * 6 + c] = 0;
        if (kind[i] == 0) {
            const double fi = z[0] < 0 ? -z[0] / Rr[i] : 0.0;
            f[i] = fi;
            cost += 0.5 * Rr[i] * fi * fi;                     /* s(z) = -(f z + R f^2 / 2) = R f^2 / 2 at f = -z / R */
            if (W && z[0] < 0) W[(size_t)i * 6] = 1.0 / Rr[i];
            continue;
        }
        const double *mu = rowmu[i];
        const double Rn = Rr[i], Rt = Rr[i + 1] * mu[0] * mu[0];
        double w[6], N = 0;
        w[0] = z[0];
        for (int r = 1; r < dm; r++) { w[r] = mu[r - 1] * z[r]; N += w[r] * w[r]; }
        N = sqrt(N);
        double y[6];
        if (w[0] >= N) {                                        /* top zone */
            for (int r = 0; r < dm; r++) { y[r] = 0; f[i + r] = 0; }
        } else if (N * Rn <= -w[0] * Rt) {                      /* bottom zone */
            y[0] = -w[0] / Rn;
            for (int r = 1; r < dm; r++) y[r] = -w[r] / Rt;
            if (W) { W[(size_t)i * 6] = 1.0 / Rn; for (int r = 1; r < dm; r++) W[(size_t)(i + r) * 6 + r] = mu[r - 1] * mu[r - 1] / Rt; }
        } else {                                                /* middle zone */
            const double D = Rn + Rt;
            y[0] = (N - w[0]) / D;
            for (int r = 1; r < dm; r++) y[r] = -y[0] * w[r] / N;
            if (W) {   /* -dy/dw = [[1/D, -u'/D], [-u/D, u u'/D + (y0/N)(I - u u')]], u = w_t / N; then scale rows and columns by S = diag(1, mu) */
                double u[6];
                for (int r = 1; r < dm; r++) u[r] = w[r] / N;
                W[(size_t)i * 6] = 1.0 / D;
                for (int r = 1; r < dm; r++) {
                    W[(size_t)i * 6 + r] = -u[r] / D * mu[r - 1];
                    W[(size_t)(i + r) * 6] = -u[r] / D * mu[r - 1];
                    for (int c = 1; c < dm; c++)
                        W[(size_t)(i + r) * 6 + c] = (u[r] * u[c] / D + (y[0] / N) * ((r == c ? 1.0 : 0.0) - u[r] * u[c])) * mu[r - 1] * mu[c - 1];
                }
            }
        }
        double quad = 0.5 * Rn * y[0] * y[0], lin = y[0] * w[0];
        for (int r = 1; r < dm; r++) { quad += 0.5 * Rt * y[r] * y[r]; lin += y[r] * w[r]; }
        cost += -(quad + lin);
        f[i] = y[0];
        for (int r = 1; r < dm; r++) f[i + r] = y[r] * mu[r - 1];
    }
    return cost;
}
static int newton_primal(int nv, int nr, const real *M, const real *J, const real *aref_r, const real *Rr_r, const real *a0_r, const int *kind,
                         const int *blkdim, const double *const *rowmu, real *f_out, int max_iter) {
    double *Md = (double *)malloc(sizeof(double) * ((size_t)nv * nv * 2 + (size_t)nr * nv + (size_t)nr * 8 + (size_t)nv * 6 + (size_t)nr * 6));
    double *H = Md + (size_t)nv * nv, *Jd = H + (size_t)nv * nv, *aref = Jd + (size_t)nr * nv, *Rr = aref + nr, *f = Rr + nr, *ftry = f + nr;
    double *a0 = ftry + nr + (size_t)nr * 4, *x = a0 + nv, *g = x + nv, *dx = g + nv, *xt = dx + nv, *tmp = xt + nv, *W = tmp + nv;
    for (int i = 0; i < nv * nv; i++) Md[i] = (double)M[i];
    for (size_t i = 0; i < (size_t)nr * nv; i++) Jd[i] = (double)J[i];
    for (int i = 0; i < nr; i++) { aref[i] = (double)aref_r[i]; Rr[i] = (double)Rr_r[i]; }
    for (int i = 0; i < nv; i++) { a0[i] = (double)a0_r[i]; x[i] = a0[i]; }
    int it;
    for (it = 0; it < max_iter; it++) {
        const double cost = newton_eval(nv, nr, Md, Jd, aref, Rr, a0, kind, blkdim, rowmu, x, f, W);
        double gn = 0, scale = 0;
        for (int i = 0; i < nv; i++) {
            double s = 0;
            for (int j = 0; j < nv; j++) s += Md[i * nv + j] * (x[j] - a0[j]);
            for (int r = 0; r < nr; r++) s -= Jd[(size_t)r * nv + i] * f[r];
            g[i] = s; gn += s * s; scale += Md[i * nv + i];
        }
        if (sqrt(gn) <= 1e-13 * (1.0 + scale)) break;
        /* H = M + J' W J (W block-diagonal) */
        for (int i = 0; i < nv * nv; i++) H[i] = Md[i];
        for (int i = 0; i < nr; i++) {
            if (kind[i] == 2) continue;
            const int dm = kind[i] == 0 ? 1 : blkdim[i];
            for (int r = 0; r < dm; r++)
                for (int c = 0; c < dm; c++) {
                    const double wv = W[(size_t)(i + r) * 6 + c];
                    if (wv == 0) continue;
                    for (int a = 0; a < nv; a++) {
                        const double ja = Jd[(size_t)(i + r) * nv + a] * wv;
                        if (ja == 0) continue;
                        for (int b = 0; b < nv; b++) H[a * nv + b] += ja * Jd[(size_t)(i + c) * nv + b];
                    }
                }
        }
        /* Cholesky solve H dx = -g */
        int ok = 1;
        for (int j = 0; j < nv && ok; j++) {
            double d = H[j * nv + j];
            for (int k = 0; k < j; k++) d -= H[j * nv + k] * H[j * nv + k];
            if (!(d > 0)) { ok = 0; break; }
            d = sqrt(d); H[j * nv + j] = d;
            for (int i = j + 1; i < nv; i++) {
                double s = H[i * nv + j];
                for (int k = 0; k < j; k++) s -= H[i * nv + k] * H[j * nv + k];
                H[i * nv + j] = s / d;
            }
        }
        if (!ok) break;
        for (int i = 0; i < nv; i++) { double s = -g[i]; for (int k = 0; k < i; k++) s -= H[i * nv + k] * tmp[k]; tmp[i] = s / H[i * nv + i]; }
        for (int i = nv - 1; i >= 0; i--) { double s = tmp[i]; for (int k = i + 1; k < nv; k++) s -= H[k * nv + i] * dx[k]; dx[i] = s / H[i * nv + i]; }
        double slope = 0;
        for (int i = 0; i < nv; i++) slope += g[i] * dx[i];
        if (!(slope < 0)) break;
        /* EXACT line search, as MuJoCo's Newton solver does: phi(al) = F(x + al dx) is convex and C^1 (piecewise quadratic: the Hessian jumps where a
         * contact changes zone, which is where a backtracking Newton step zig-zags), so its minimiser is the root of the increasing function
         * phi'(al) = grad F(x + al dx) . dx -- bracketed by doubling, then bisected */
        double lo = 0.0, hi = 1.0, dhi = 0;
        int ls;
        for (ls = 0; ls < 60; ls++) {
            for (int i = 0; i < nv; i++) xt[i] = x[i] + hi * dx[i];
            newton_eval(nv, nr, Md, Jd, aref, Rr, a0, kind, blkdim, rowmu, xt, ftry, NULL);
            dhi = 0;
            for (int i = 0; i < nv; i++) {
                double sg = 0;
                for (int j = 0; j < nv; j++) sg += Md[i * nv + j] * (xt[j] - a0[j]);
                for (int r = 0; r < nr; r++) sg -= Jd[(size_t)r * nv + i] * ftry[r];
                dhi += sg * dx[i];
            }
            if (dhi >= 0) break;
            lo = hi; hi *= 2.0;
        }
        double al = hi;
        if (dhi > 0) {
            for (int b = 0; b < 80; b++) {
                al = 0.5 * (lo + hi);
                for (int i = 0; i < nv; i++) xt[i] = x[i] + al * dx[i];
                newton_eval(nv, nr, Md, Jd, aref, Rr, a0, kind, blkdim, rowmu, xt, ftry, NULL);
                double dm_ = 0;
                for (int i = 0; i < nv; i++) {
                    double sg = 0;
                    for (int j = 0; j < nv; j++) sg += Md[i * nv + j] * (xt[j] - a0[j]);
                    for (int r = 0; r < nr; r++) sg -= Jd[(size_t)r * nv + i] * ftry[r];
                    dm_ += sg * dx[i];
                }
                if (dm_ > 0) hi = al; else lo = al;
                if (hi - lo <= 1e-15 * hi) break;
            }
            al = 0.5 * (lo + hi);
        }
        for (int i = 0; i < nv; i++) xt[i] = x[i] + al * dx[i];
        if (getenv("ORC_NEWTON_DEBUG")) fprintf(stderr, "newton it=%d cost=%.17g |g|=%.3g slope=%.3g al=%.3g\n", it, cost, sqrt(gn), slope, al);
        for (int i = 0; i < nv; i++) x[i] = xt[i];
    }
    newton_eval(nv, nr, Md, Jd, aref, Rr, a0, kind, blkdim, rowmu, x, f, NULL);
    if (getenv("ORC_NEWTON_DEBUG2")) {
        for (int i = 0; i < nr; i++) {
            double z = -aref[i];
            for (int d = 0; d < nv; d++) z += Jd[(size_t)i * nv + d] * x[d];
            fprintf(stderr, "row %d kind %d blk %d z=%g R=%g f=%g z+Rf=%g mu=%g\n", i, kind[i], blkdim[i], z, Rr[i], f[i], z + Rr[i] * f[i], kind[i] == 2 ? rowmu[i][0] : 0.0);
        }
    }
    for (int i = 0; i < nr; i++) f_out[i] = (real)f[i];
    free(Md);
    return it;
}

/* ---- the PRODUCT's faithful solver (orc_params.solver = 2, round 5; = the NEWTON kernels): MuJoCo's default algorithm with a fixed budget ---------------
 * Newton's method on the primal problem (see newton_primal above for F and the zones), all nv accelerations at once, in `real` arithmetic (the fp32 twin of this
 * oracle exercises the float behaviour of the kernels' iteration):
 *   start     x0 = a0 + M^-1 J' f_carried   (the constraint forces carried from the previous substep / control step; MuJoCo: qacc_warmstart)
 *   iteration g = M (x - a0) - J' f(x),  H = M + J' W(x) J  (W: block-diagonal Jacobian of -f w.r.t. the row residuals, zero / diagonal / rank-structured in
 *             the top / bottom / middle zone of a contact's cone),  dx = -H^-1 g  (Cholesky);  stop when the Newton decrement -g'dx <= newton_tol^2 (1 + |x0|_M^2)
 *   line search on phi(al) = F(x + al dx), convex and C^1: one gradient pass gives phi'(al) = grad F(x + al dx) . dx and phi''(al) = dx'M dx + sum_b jd_b'W_b jd_b.
 *             First al = 1 (exact while no contact changes zone); no bracket yet: the Newton step on phi' from the last point; bracket [lo, hi]: the Newton candidate
 *             from the end with the smaller |phi'| (from the stale end after two updates of the same end in a row), a candidate outside the open bracket being none
 *             (phi' is piecewise linear apart from the cones' middle zones: a candidate is the exact root unless a kink lies in between -- a joint limit or contact
 *             switching on along the step is found in one evaluation from the steep side).  Both ends pointing outside: a steep piece hides between two flat ones,
 *             a sliding contact that comes to rest within the step (its friction force turns around where N(al) = |w_t(al)| passes its minimum) -- try the closed-form
 *             minimiser of that block's N(al), else the Illinois secant point, else the midpoint.  At most ls_iters evaluations, stop when |phi'(al)| <= ls_tol |phi'(0)|;
 *             budget spent: the lower end of the bracket (F decreases on [0, root]).  (Round-5 record, profiles/r05_solver_decision.txt: the search must be accurate.
 *             The first round-5 kernels ran a derivative-only Illinois search, ORC_LS_ILLINOIS=1 here: it needs a third more evaluations and creeps when a limit row
 *             of 1e4 x the slope switches on inside the bracket -- tests/test_gpu_parity.py::test_joint_limit_rows was 0.08 rad off.)
 *   at most newton_iters iterations (default 20: a cold start on a finger deep in the floor with both fingers and a proxy down needs 12).  On the GPU both loops are left wave-uniformly (when every lane of the wave has met the criterion), so a lane may
 *   iterate further than here -- at the optimum that changes nothing beyond rounding.
 * Returns the forces f(x) and the accelerations x themselves: the integration uses x (M (x - a0) = J' f at the optimum). */
#define DEC_FLOOR 3e-10   /* a Newton decrement below DEC_FLOOR |x - a0|_M^2 that no longer shrinks is rounding (see newton_product) */
#define LS_NOISE 1e-5   /* relative rounding floor of phi'(al) evaluated in float (about 40 terms of either sign) */
static void prim_forces(int nr, const real *z, const real *Rr, const int *kind, const int *blkdim, const double *const *rowmu, real *f, real *W /* nr x 6 or NULL */) {
    for (int i = 0; i < nr; i++) {
        if (kind[i] == 2) continue;
        const int dm = kind[i] == 0 ? 1 : blkdim[i];
        if (W) for (int r = 0; r < dm; r++) for (int c = 0; c < 6; c++) W[(size_t)(i + r) * 6 + c] = 0;
        if (kind[i] == 0) {
            f[i] = z[i] < 0 ? -z[i] / Rr[i] : 0;
            if (W && z[i] < 0) W[(size_t)i * 6] = (real)1 / Rr[i];
            continue;
        }
        const double *mu = rowmu[i];
        const real Rn = Rr[i], Rt = Rr[i + 1] * (real)(mu[0] * mu[0]);
        real w[6], N = 0, y[6];
        w[0] = z[i];
        for (int r = 1; r < dm; r++) { w[r] = (real)mu[r - 1] * z[i + r]; N += w[r] * w[r]; }
        N = (real)sqrt((double)N);
        if (w[0] >= N) { for (int r = 0; r < dm; r++) y[r] = 0; }
        else if (N * Rn <= -w[0] * Rt) {
            y[0] = -w[0] / Rn;
            for (int r = 1; r < dm; r++) y[r] = -w[r] / Rt;
            if (W) { W[(size_t)i * 6] = (real)1 / Rn; for (int r = 1; r < dm; r++) W[(size_t)(i + r) * 6 + r] = (real)(mu[r - 1] * mu[r - 1]) / Rt; }
        } else {
            const real D = Rn + Rt;
            y[0] = (N - w[0]) / D;
            for (int r = 1; r < dm; r++) y[r] = -y[0] * w[r] / N;
            if (W) {
                real u[6];
                for (int r = 1; r < dm; r++) u[r] = w[r] / N;
                W[(size_t)i * 6] = (real)1 / D;
                for (int r = 1; r < dm; r++) {
                    W[(size_t)i * 6 + r] = -u[r] / D * (real)mu[r - 1];
                    W[(size_t)(i + r) * 6] = -u[r] / D * (real)mu[r - 1];
                    for (int c = 1; c < dm; c++)
                        W[(size_t)(i + r) * 6 + c] = (u[r] * u[c] / D + (y[0] / N) * ((r == c ? (real)1 : (real)0) - u[r] * u[c])) * (real)(mu[r - 1] * mu[c - 1]);
                }
            }
        }
        f[i] = y[0];
        for (int r = 1; r < dm; r++) f[i + r] = y[r] * (real)mu[r - 1];
    }
}
/* study aid (tools/newton_cost_study.py): per (env, substep) Newton iterations and line-search evaluations of newton_product, [n][substeps][2] int32 */
static int32_t *g_newton_trace = NULL;
static int g_newton_trace_sub = 0;
void orc_set_newton_trace(int32_t *buf, int substeps) { g_newton_trace = buf; g_newton_trace_sub = substeps; }
static __thread int32_t *t_trace_slot = NULL;
static int newton_product(int nv, int nr, const real *M, const real *Lm, const real *J, const real *aref, const real *Rr, const real *a0, const int *kind,
                          const int *blkdim, const double *const *rowmu, real *f, real *x_out, int iters, int ls_iters, double tol, double ls_tol) {
    real x[ORC_NV_MAX], xa[ORC_NV_MAX], g[ORC_NV_MAX], H[ORC_NV_MAX * ORC_NV_MAX], dx[ORC_NV_MAX], tmp[ORC_NV_MAX];
    real *z = (real *)malloc(sizeof(real) * (size_t)(nr + 1) * 10), *W = z + nr + 1, *jd = W + (size_t)(nr + 1) * 6 + nr + 1, *z0 = jd + nr + 1;
    static int np_debug = -1, ls_illinois = 0;
    if (np_debug < 0) { ls_illinois = getenv("ORC_LS_ILLINOIS") != NULL; np_debug = getenv("ORC_NEWTON_DEBUG3") != NULL; }
    for (int d = 0; d < nv; d++) { real acc = 0; for (int i = 0; i < nr; i++) acc += J[(size_t)i * nv + d] * f[i]; tmp[d] = acc; }
    chol_solve(Lm, nv, tmp);
    real scale = 1;
    for (int d = 0; d < nv; d++) { x[d] = a0[d] + tmp[d]; real acc = 0; for (int j = 0; j < nv; j++) acc += M[d * nv + j] * a0[j]; scale += a0[d] * acc; }
    /* gradient of F at xx (and, with Wm, the blocks of W) */
#define NP_GRAD(xx, Wm)                                                                                                                        \
    do {                                                                                                                                       \
        for (int i = 0; i < nr; i++) { real acc = -aref[i]; for (int d = 0; d < nv; d++) acc += J[(size_t)i * nv + d] * (xx)[d]; z[i] = acc; } \
        prim_forces(nr, z, Rr, kind, blkdim, rowmu, f, (Wm));                                                                                  \
        for (int i = 0; i < nv; i++) {                                                                                                         \
            real acc = 0;                                                                                                                      \
            for (int j = 0; j < nv; j++) acc += M[i * nv + j] * ((xx)[j] - a0[j]);                                                             \
            for (int r = 0; r < nr; r++) acc -= J[(size_t)r * nv + i] * f[r];                                                                  \
            g[i] = acc;                                                                                                                        \
        }                                                                                                                                      \
    } while (0)
    int it;
    double dprev = 1e300;
    for (it = 0; it < iters; it++) {
        NP_GRAD(x, W);
        for (int a = 0; a < nv; a++) for (int c = 0; c < nv; c++) H[a * nv + c] = M[a * nv + c];
        for (int i = 0; i < nr; i++) {
            if (kind[i] == 2) continue;
            const int dm = kind[i] == 0 ? 1 : blkdim[i];
            for (int r = 0; r < dm; r++) for (int c = 0; c < dm; c++) {
                const real wv = W[(size_t)(i + r) * 6 + c];
                if (wv == 0) continue;
                for (int a = 0; a < nv; a++) { const real ja = J[(size_t)(i + r) * nv + a] * wv; if (ja == 0) continue;
                    for (int c2 = 0; c2 < nv; c2++) H[a * nv + c2] += ja * J[(size_t)(i + c) * nv + c2]; }
            }
        }
        for (int a = 0; a < nv; a++) dx[a] = -g[a];
        if (chol(H, nv)) break;
        chol_solve(H, nv, dx);
        real d0 = 0;
        for (int a = 0; a < nv; a++) d0 += g[a] * dx[a];
        if (np_debug) {
            real dd2 = 0;
            for (int a = 0; a < nv; a++) { real acc = 0; for (int c = 0; c < nv; c++) acc += M[a * nv + c] * (x[c] - a0[c]); dd2 += (x[a] - a0[a]) * acc; }
            fprintf(stderr, "np it=%d decrement=%.3e tol2=%.3e dist2=%.3e zones:", it, (double)-d0, tol * tol * (double)scale, (double)dd2);
            for (int i = 0; i < nr; i++) { if (kind[i] == 2) continue; fprintf(stderr, " [%d k%d z=%.3e f=%.3e R=%.1e]", i, kind[i], (double)z[i], (double)f[i], (double)Rr[i]); }
            fprintf(stderr, "\n");
        }
        /* Newton decrement: converged (or no descent).  Second exit, for float arithmetic: the gradient M (x - a0) - J'f is a difference of two vectors of the same
         * size whose force part carries the cancellation of stiff rows (f = -z / R, z = J x - aref, R ~ 1e-4): its rounding leaves a decrement that no iteration
         * removes -- a finger pressed 5 mm into the floor: 1e-5 against newton_tol^2 (1 + |a0|_M^2) = 4e-9, and the float solve ran into its iteration budget in
         * 9 of 10 such envs.  The floor is recognised by what it is: a decrement at rounding level RELATIVE to the problem (<= 3e-10 |x - a0|_M^2) that has stopped
         * shrinking (not below a quarter of the previous iteration's).  In double a converging iteration never meets both (it shrinks quadratically down there). */
        real dist2 = 0;
        for (int a = 0; a < nv; a++) { real acc = 0; for (int c = 0; c < nv; c++) acc += M[a * nv + c] * (x[c] - a0[c]); dist2 += (x[a] - a0[a]) * acc; }
        if (!((double)-d0 > tol * tol * (double)scale)) break;
        if ((double)-d0 <= DEC_FLOOR * (double)dist2 && (double)-d0 >= 0.25 * dprev) break;
        dprev = (double)-d0;
        /* line search on phi'(al) = grad F(x + al dx) . dx, monotone increasing, with phi''(al) = dx'M dx + sum_b jd_b' W_b(al) jd_b from the same pass */
        real q1 = 0;
        for (int a = 0; a < nv; a++) { real acc = 0; for (int c = 0; c < nv; c++) acc += M[a * nv + c] * dx[c]; q1 += dx[a] * acc; }
        for (int i = 0; i < nr; i++) { real acc = 0; for (int d = 0; d < nv; d++) acc += J[(size_t)i * nv + d] * dx[d]; jd[i] = acc; }
        real al = 1, lo_a = 0, hi_a = -1, dlo = d0, dhi = 0, hlo = -d0, hhi = 0, dlo_m = d0, dhi_m = 0;
        int last_side = 0, same = 0, ls_done = 0;
        for (int i = 0; i < nr; i++) z0[i] = z[i];
        if (np_debug && getenv("ORC_LS_SCAN")) {
            for (double e = -14; e <= 0; e += 1) {
                const real aa = e < -13.5 ? 0 : (real)pow(10.0, e);
                for (int d = 0; d < nv; d++) xa[d] = x[d] + aa * dx[d];
                NP_GRAD(xa, W);
                real dphi = 0;
                for (int a = 0; a < nv; a++) dphi += g[a] * dx[a];
                fprintf(stderr, "   scan al=%.1e dphi=%.6e  |", (double)aa, (double)dphi);
                for (int i = 0; i < nr; i++) fprintf(stderr, " %.3e", (double)z[i]);
                fprintf(stderr, "\n");
            }
        }
        for (int ls = 0; ls < ls_iters; ls++) {
            for (int d = 0; d < nv; d++) xa[d] = x[d] + al * dx[d];
            NP_GRAD(xa, W);
            /* phi'(al) = [M (x + al dx - a0)] . dx - f(al) . jd: two partial sums that cancel at the root -- in float arithmetic their rounding, not ls_tol, bounds what
             * the search can resolve near convergence, so the stopping rule carries a noise floor relative to their magnitudes (harmless in double) */
            real mpart = 0, fpart = 0, ddphi = q1;
            for (int a = 0; a < nv; a++) { real acc = 0; for (int c = 0; c < nv; c++) acc += M[a * nv + c] * (xa[c] - a0[c]); mpart += acc * dx[a]; }
            for (int i = 0; i < nr; i++) fpart += f[i] * jd[i];
            const real dphi = mpart - fpart;
            for (int i = 0; i < nr; i++) {
                if (kind[i] == 2) continue;
                const int dm = kind[i] == 0 ? 1 : blkdim[i];
                for (int r = 0; r < dm; r++) for (int c = 0; c < dm; c++) ddphi += jd[i + r] * W[(size_t)(i + r) * 6 + c] * jd[i + c];
            }
            const int done = fabs((double)dphi) <= ls_tol * fabs((double)d0) + LS_NOISE * (fabs((double)mpart) + fabs((double)fpart));
            if (t_trace_slot) t_trace_slot[1]++;
            if (np_debug) fprintf(stderr, "     ls=%d al=%.6g dphi=%.3e ddphi=%.3e (d0=%.3e)\n", ls, (double)al, (double)dphi, (double)ddphi, (double)d0);
            if (done) { ls_done = 1; break; }
            real an;
            if (ls_illinois) {   /* (study: the derivative-only search of the first round-5 kernels) */
                if (dphi < 0) { if (hi_a >= 0 && lo_a > 0) dhi *= (real)0.5; lo_a = al; dlo = dphi; }
                else { if (hi_a >= 0) dlo *= (real)0.5; hi_a = al; dhi = dphi; }
                if (hi_a < 0) an = 2 * al;
                else { an = lo_a - dlo * (hi_a - lo_a) / (dhi - dlo); if (!(an > lo_a && an < hi_a)) an = (real)0.5 * (lo_a + hi_a); }
            } else {
                const int side = dphi < 0 ? -1 : 1;
                /* (the Illinois rule for the secant fall-back: a second update of the same end in a row halves the value kept for the stale end) */
                if (dphi < 0) { if (last_side < 0) dhi_m *= (real)0.5; lo_a = al; dlo = dphi; hlo = ddphi; dlo_m = dphi; }
                else { if (last_side > 0) dlo_m *= (real)0.5; hi_a = al; dhi = dphi; hhi = ddphi; dhi_m = dphi; }
                same = side == last_side ? same + 1 : 0;
                last_side = side;
                if (hi_a < 0) an = lo_a - dlo / hlo;      /* no bracket yet: the Newton step from the last point (phi'' >= dx'M dx > 0) moves right */
                else {
                    /* bracket [lo, hi]: Newton candidates from both ends -- on a piecewise linear phi' each is the exact root unless a kink lies in between.  Take
                     * the one from the end whose |phi'| is smaller (from the stale end after two updates of the same side in a row); a candidate outside the open
                     * bracket is no candidate.  When both ends point outside, a steep piece hides between two flat ones: a sliding contact that comes to rest within
                     * the step -- its tangential residual passes (almost) through zero, where the friction force turns around; the root sits in the narrow sticking
                     * zone around the minimum of that block's N(al) = |w_t(al)|, a quadratic in al whose minimiser is known in closed form.  Take the block minimiser
                     * inside the bracket that is closest to the (Illinois) secant point; without one, the secant point; then the midpoint */
                    const real cl = lo_a - dlo / hlo, ch = hi_a - dhi / hhi;
                    const real mg = (real)1e-4 * (hi_a - lo_a), blo = lo_a + mg, bhi = hi_a - mg;   /* (inside by a margin: a candidate that repeats an end teaches nothing) */
                    const int vl = cl > blo && cl < bhi, vh = ch > blo && ch < bhi;
                    int from_lo = fabs((double)dlo) <= fabs((double)dhi);
                    if (same >= 2) from_lo = side > 0;
                    real sec = lo_a - dlo_m * (hi_a - lo_a) / (dhi_m - dlo_m);
                    if (!(sec > lo_a && sec < hi_a)) sec = (real)0.5 * (lo_a + hi_a);
                    if (vl || vh) an = from_lo ? (vl ? cl : ch) : (vh ? ch : cl);
                    else {
                        an = sec;
                        real bestd = -1;
                        for (int i = 0; i < nr; i++) {
                            if (kind[i] != 1) continue;
                            const double *mu = rowmu[i];
                            real qa = 0, qb = 0;
                            for (int r = 1; r < blkdim[i]; r++) { const real m2 = (real)(mu[r - 1] * mu[r - 1]); qa += m2 * jd[i + r] * jd[i + r]; qb += m2 * z0[i + r] * jd[i + r]; }
                            if (!(qa > 0)) continue;
                            const real am = -qb / qa;
                            if (!(am > blo && am < bhi)) continue;
                            /* ... and only where the block IS in its sticking (bottom) zone at the minimiser: N_min Rn <= -w_n Rt */
                            real qc = 0;
                            for (int r = 1; r < blkdim[i]; r++) qc += (real)(mu[r - 1] * mu[r - 1]) * z0[i + r] * z0[i + r];
                            const real n2 = qc - qb * qb / qa, wn = z0[i] + am * jd[i], Rtb = Rr[i + 1] * (real)(mu[0] * mu[0]);
                            if (!(wn < 0) || (n2 > 0 ? n2 : 0) * Rr[i] * Rr[i] > wn * wn * Rtb * Rtb) continue;
                            const real dist = (real)fabs((double)(am - sec));
                            if (bestd < 0 || dist < bestd) { bestd = dist; an = am; }
                        }
                    }
                }
            }
            al = an;
        }
        /* budget spent without meeting ls_tol: step to the lower end of the bracket (F decreases on [0, root]); none found: to the smallest point evaluated */
        if (!ls_done && !ls_illinois) al = lo_a > 0 ? lo_a : hi_a;
        for (int d = 0; d < nv; d++) x[d] += al * dx[d];
        if (t_trace_slot) t_trace_slot[0]++;
    }
    NP_GRAD(x, NULL);
#undef NP_GRAD
    for (int d = 0; d < nv; d++) x_out[d] = x[d];
    free(z);
    return it;
}

/* ------------------------------------------------------------------------------------------------ */
/* one physics substep == mujoco.mj_step (reach_cube_env.py:276-277) -- MJ-DOC restatement          */
/* ------------------------------------------------------------------------------------------------ */
typedef struct {
    real ee[3]; real cube[2][3];
    uint32_t active_mask;  /* OR over the substeps of the control step: bit = warm-slot id of an active contact (0..17), 18+j joint-limit of dof j */
    uint32_t active_count; /* sum over the substeps of the number of active contacts + limits */
    uint32_t max_sweeps;   /* largest PGS sweep count of a substep (adaptive mode) */
    double kkt;            /* largest KKT natural residual of a substep's solution (orc_io.kkt), only when asked for */
    int want_kkt;
    uint32_t choice;       /* wrapping sum over substeps s (weight 2s+1) and active constraints of (slot+1)(sel+1) 2654435761: the discrete choices
                              behind the contacts (which vertex, manifold candidate, box face, proxy member, limit side), plus the
                              number of IK iterations x 0x9E3779B1 */
} lag_t;
/* constraint forces carried from one substep to the next WITHIN a control step (zero at its start, so that a
 * control step stays a pure function of (qpos, qvel, action)); MuJoCo warm-starts its solver likewise */
typedef struct { real lim[12]; real slot[30][6]; } warm_t;   /* slots 24..27: the extra cube<->cube points of the D5 study */
int orc_warm_bytes(void) { return (int)sizeof(warm_t); } /* stride of orc_io.warm */

static void substep(const orc_params *P, const task_model *T, real *qpos, real *qvel, const real *ctrl, lag_t *lag,
                    warm_t *warm, int diag, int sub_index) {
    const int nc = T->ncube, nv = 6 + 6 * nc;
    const real h = (real)H_STEP;
    kin_t K;
    /* -- position stage: normalise quaternions, kinematics */
    K.ncube = nc;
    K.mu_cube = T->mu_cube; K.mu_finger_cube = T->mu_finger_cube;
    K.cc_points = P->cc_points == 8 ? 8 : 4;
    for (int c = 0; c < nc; c++) {
        real *qq = qpos + 6 + 7 * c + 3, n2 = 0;
        for (int k = 0; k < 4; k++) n2 += qq[k] * qq[k];
        n2 = (real)sqrt((double)n2);
        for (int k = 0; k < 4; k++) qq[k] /= n2;
        quat2mat(K.cR[c], qq);
        v3copy(K.cp[c], qpos + 6 + 7 * c);
    }
    arm_kinematics(qpos, &K);
    /* P8: what data.site_xpos / data.xpos hold after this mj_step returns */
    v3copy(lag->ee, K.site);
    for (int c = 0; c < nc; c++) v3copy(lag->cube[c], K.cp[c]);

    /* -- inertia */
    real M[ORC_NV_MAX * ORC_NV_MAX], L[ORC_NV_MAX * ORC_NV_MAX], Ma[36];
    memset(M, 0, sizeof M);
    arm_mass(&K, 1, Ma);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) M[i * nv + j] = Ma[i * 6 + j];
    for (int c = 0; c < nc; c++)
        for (int k = 0; k < 3; k++) {
            M[(6 + 6 * c + k) * nv + 6 + 6 * c + k] = (real)T->cube_mass;
            M[(9 + 6 * c + k) * nv + 9 + 6 * c + k] = (real)T->cube_inertia;
        }
    memcpy(L, M, sizeof(real) * nv * nv);
    chol(L, nv);

    /* -- velocity stage + actuation: qfrc_smooth = passive - bias + actuator */
    real tau[ORC_NV_MAX], bias[6];
    arm_bias(&K, qvel, bias);
    for (int j = 0; j < 6; j++) {
        real c = ctrl[j];
        if (c < (real)JNT_LO[j]) c = (real)JNT_LO[j]; /* ctrlrange clamp (inheritrange) */
        if (c > (real)JNT_HI[j]) c = (real)JNT_HI[j];
        real f = (real)KP * (c - qpos[j]) - (real)KV * qvel[j]; /* position actuator: gain kp, bias (0,-kp,-kv) */
        if (f > (real)FRC_LIM) f = (real)FRC_LIM;               /* actuatorfrcrange */
        if (f < (real)-FRC_LIM) f = (real)-FRC_LIM;
        tau[j] = -(real)DAMPING * qvel[j] + f - bias[j];
    }
    for (int c = 0; c < nc; c++) {
        real *t = tau + 6 + 6 * c;
        t[0] = 0; t[1] = 0; t[2] = -(real)T->cube_mass * (real)GRAV; /* isotropic inertia: no gyroscopic term */
        t[3] = t[4] = t[5] = 0;
    }
    real a0[ORC_NV_MAX];
    memcpy(a0, tau, sizeof(real) * nv);
    chol_solve(L, nv, a0); /* qacc_smooth */

    /* -- collision (D3, D5) in fixed order: floor-cube(s), cube-cube, rails, sphere-cube, sphere-floor, arm-link proxies */
    contact_t con[MAX_CONTACTS];
    int ncon = 0;
    for (int c = 0; c < nc; c++) ncon += collide_plane_box(&K, c, con + ncon);
    if (nc == 2) ncon += collide_box_box(&K, con + ncon);
    if (T->walls) ncon += collide_walls(&K, con + ncon);
    /* one contact per finger sphere against the cube it penetrates deepest (tie: cube 0), then the floor */
    for (int s = 0; s < NSPH; s++) {
        contact_t cand[2];
        int have = 0;
        for (int c = 0; c < nc; c++) {
            contact_t tmp;
            if ((P->finger_geom == 1 ? collide_box_pad(&K, c, s, &tmp) : collide_box_sphere(&K, c, s, &tmp)) && (!have || tmp.dist < cand[0].dist)) { cand[0] = tmp; have = 1; }
        }
        if (have) con[ncon++] = cand[0];
    }
    for (int s = 0; s < NSPH; s++)
        if (P->finger_geom == 1 ? collide_plane_pad(&K, s, T->walls, con + ncon) : collide_plane_sphere(&K, s, T->walls, con + ncon)) ncon++;
    if (P->arm_collision)
        for (int g = 0; g < (P->proxy_groups == 3 ? 3 : 1); g++)
            if (collide_link_group(&K, g, P->proxy_groups == 3 ? 3 : 1, T->walls, con + ncon)) ncon++;

    /* -- constraint rows: joint limits first, then contacts (n, t1, t2 [, torsion]) */
    real J[MAX_ROWS * ORC_NV_MAX], aref[MAX_ROWS], Rr[MAX_ROWS];
    int kind[MAX_ROWS]; /* 0 limit, 1 contact-normal (block start), 2 friction */
    int blk0[MAX_ROWS], blkdim[MAX_ROWS]; /* first row and row count of the contact a row belongs to */
    int grp[MAX_ROWS];                    /* sweep group of the row (orc_params.jacobi) */
    real *wptr[MAX_ROWS]; /* where this row's force is kept between substeps */
    const double *rowmu[MAX_ROWS];
    int nr = 0;
    uint32_t amask = 0, choice = 0;
    memset(J, 0, sizeof J);
    for (int j = 0; j < 6; j++)
        for (int side = 0; side < 2; side++) {
            real pos = side == 0 ? qpos[j] - (real)JNT_LO[j] : (real)JNT_HI[j] - qpos[j];
            if (!(pos < 0)) continue;
            real sg = side == 0 ? (real)1 : (real)-1;
            J[nr * nv + j] = sg;
            double k, b, imp;
            kbi(SOLREF, SOLIMP_DEFAULT, (double)pos, &k, &b, &imp);
            real vel = sg * qvel[j];
            aref[nr] = (real)(-b * (double)vel - k * imp * (double)pos);
            double r = (1 - imp) / imp * g_inv_dof[j];
            Rr[nr] = (real)(r > MJ_MINVAL ? r : MJ_MINVAL);
            kind[nr] = 0; rowmu[nr] = 0; blk0[nr] = nr; blkdim[nr] = 1; grp[nr] = 0;
            wptr[nr] = &warm->lim[2 * j + side];
            amask |= 1u << (18 + j);
            choice += (uint32_t)(18 + j + 1) * (uint32_t)(side + 1) * 2654435761u;
            nr++;
        }
    for (int ci = 0; ci < ncon; ci++) {
        contact_t *ct = &con[ci];
        /* finger geoms: follower.xml:15 condim="6" wins the max rule.  condim6 = 1: finger<->cube contacts get the two rolling rows
         * (the kernel's finger_cube_condim = 6); 2: finger<->floor contacts as well (study of deviation D4 only) */
        if ((P->condim6 >= 1 && (ct->slot == 12 || ct->slot == 13)) || (P->condim6 >= 2 && (ct->slot == 14 || ct->slot == 15))) ct->dim = 6;
        real Jp1[3 * ORC_NV_MAX], Jr1[3 * ORC_NV_MAX], Jp2[3 * ORC_NV_MAX], Jr2[3 * ORC_NV_MAX];
        jac_point(&K, nv, ct->b1, ct->pos, Jp1, Jr1);
        jac_point(&K, nv, ct->b2, ct->pos, Jp2, Jr2);
        double k, b, imp, t1, r1, t2, r2;
        kbi(SOLREF, ct->solimp, (double)ct->dist, &k, &b, &imp);
        body_invweight(T, ct->b1, &t1, &r1);
        body_invweight(T, ct->b2, &t2, &r2);
        double Rn = (1 - imp) / imp * (t1 + t2);
        if (Rn < MJ_MINVAL) Rn = MJ_MINVAL;
        double impr = P->impratio > MJ_MINVAL ? P->impratio : MJ_MINVAL;
        double Rf = Rn / impr; /* elliptic cone: friction rows regularised by R/impratio, scaled mu0^2/mu_j^2 */
        amask |= 1u << ct->slot;
        choice += (uint32_t)(ct->slot + 1) * (uint32_t)(ct->sel + 1) * 2654435761u;
        for (int r = 0; r < ct->dim; r++) {
            real *Jrow = J + (size_t)(nr + r) * nv;
            const real *fr = ct->frame + 3 * (r < 3 ? r : (r == 3 ? 0 : r - 3)); /* rows 4, 5: rotation about t1, t2 */
            for (int d = 0; d < nv; d++) {
                real s = 0;
                if (r < 3) for (int k3 = 0; k3 < 3; k3++) s += fr[k3] * (Jp2[k3 * nv + d] - Jp1[k3 * nv + d]);
                else for (int k3 = 0; k3 < 3; k3++) s += fr[k3] * (Jr2[k3 * nv + d] - Jr1[k3 * nv + d]);
                Jrow[d] = s;
            }
            real vel = 0;
            for (int d = 0; d < nv; d++) vel += Jrow[d] * qvel[d];
            double posr = r == 0 ? (double)ct->dist : 0.0;
            aref[nr + r] = (real)(-b * (double)vel - k * imp * posr);
            double Rrow = r == 0 ? Rn : Rf * ct->mu[0] * ct->mu[0] / (ct->mu[r - 1] * ct->mu[r - 1]);
            Rr[nr + r] = (real)Rrow;
            kind[nr + r] = r == 0 ? 1 : 2;
            rowmu[nr + r] = ct->mu;
            blk0[nr + r] = nr; blkdim[nr + r] = ct->dim;
            /* group A (with the limits): finger<->floor, arm-link proxies -- and, with TWO cubes, the finger<->cube contacts too: group A is then "every row that touches the
             * arm", group B "rows between cubes and floor / cube and cube" (StackTwoCubes: its group B carries two cubes' floor rows and the cube<->cube rows, so the kernels'
             * cube wave is the long one; measured on the oracle the assignment does not change the distance to the optimum).  One cube: finger<->cube rows stay in B. */
            grp[nr + r] = ((ct->slot >= 14 && ct->slot < 24) || (nc == 2 && (ct->slot == 12 || ct->slot == 13))) ? 0 : 1;
            wptr[nr + r] = &warm->slot[ct->slot][r];
        }
        nr += ct->dim;
    }
    lag->active_mask |= amask;
    lag->active_count += (uint32_t)__builtin_popcount(amask);
    lag->choice += choice * (uint32_t)(2 * sub_index + 1); /* odd weight: the same choice in another substep hashes differently */

    /* -- dual problem: A = J M^-1 J^T, b = J a0 - aref ; PGS, warm start, fixed or adaptive sweep count (D1, D2) */
    real f[MAX_ROWS];
    real qfc[ORC_NV_MAX];
    real xsol[ORC_NV_MAX];   /* solver = 3: the primal iterate (accelerations) */
    int use_x = 0;
    memset(qfc, 0, sizeof qfc);
    if (nr > 0) {
        real *MiJt = (real *)malloc(sizeof(real) * (size_t)nr * nv);
        real *A = (real *)malloc(sizeof(real) * (size_t)nr * nr);
        real bvec[MAX_ROWS];
        for (int i = 0; i < nr; i++) {
            memcpy(MiJt + (size_t)i * nv, J + (size_t)i * nv, sizeof(real) * nv);
            chol_solve(L, nv, MiJt + (size_t)i * nv);
        }
        for (int i = 0; i < nr; i++) {
            for (int j = 0; j < nr; j++) {
                real s = 0;
                for (int d = 0; d < nv; d++) s += J[(size_t)i * nv + d] * MiJt[(size_t)j * nv + d];
                A[(size_t)i * nr + j] = s;
            }
            real s = 0;
            for (int d = 0; d < nv; d++) s += J[(size_t)i * nv + d] * a0[d];
            bvec[i] = s - aref[i];
            f[i] = P->warm_start ? *wptr[i] : 0; /* cold start when warm_start == 0 */
        }
        /* pgs_iters > 0: exactly that many sweeps.  pgs_iters < 0 ("converged" mode): sweep until the largest force change of
         * a sweep is <= pgs_tol * (1 + largest |force|), at most ORC_PGS_CAP sweeps */
        const int adaptive = P->pgs_iters < 0;
        int max_it = adaptive ? (P->pgs_cap > 0 ? P->pgs_cap : ORC_PGS_CAP) : P->pgs_iters;
        if (P->solver == 1) {   /* the exact optimum of the convex problem (primal Newton, as MuJoCo's default solver) instead of PGS sweeps */
            const int nit = newton_primal(nv, nr, M, J, aref, Rr, a0, kind, blkdim, rowmu, f, 100);
            if (getenv("ORC_SWEEP_SUM")) lag->max_sweeps += (uint32_t)nit;
            else if ((uint32_t)nit > lag->max_sweeps) lag->max_sweeps = (uint32_t)nit;
            max_it = 0;
        }
        if (P->solver == 2) {   /* the product's faithful solver: Newton on the primal with a fixed budget (see newton_product) */
            const int nit = newton_product(nv, nr, M, L, J, aref, Rr, a0, kind, blkdim, rowmu, f, xsol, P->newton_iters > 0 ? P->newton_iters : 30,
                                           P->ls_iters > 0 ? P->ls_iters : 8, P->newton_tol > 0 ? P->newton_tol : 1e-6, P->ls_tol > 0 ? P->ls_tol : 1e-2);
            use_x = 1;
            if (getenv("ORC_SWEEP_SUM")) lag->max_sweeps += (uint32_t)nit;
            else if ((uint32_t)nit > lag->max_sweeps) lag->max_sweeps = (uint32_t)nit;
            max_it = 0;
        }
        double lastchange = 0;
        int sweeps = 0;
        for (int it = 0; it < max_it; it++) {
            lastchange = 0;
            double fmaxabs = 0;
            real f_start[MAX_ROWS];   /* stopping test of the converged mode: NET change of a row over the sweep (after the cone projection) */
            for (int i = 0; i < nr; i++) f_start[i] = f[i];
            for (int i = 0; i < nr; i++) {
                if (P->cone == 2 && kind[i] == 1) {   /* block at the apex: escape if zero is not its optimum, then the ordinary row updates */
                    int at_apex = 1;
                    for (int r = 0; r < blkdim[i]; r++) at_apex = at_apex && f[i + r] == 0;
                    if (at_apex) pgs_apex_escape(A, nr, bvec, Rr, f, i, blkdim[i], rowmu[i]);
                }
                /* orc_params.jacobi (default, = the kernels): the rows form two groups -- A: joint limits, finger<->floor, arm-link proxies (what the kernels' arm wave
                 * owns); B: floor<->cube, cube<->cube, rails, finger<->cube (the cube wave) -- that sweep CONCURRENTLY: Gauss-Seidel inside a group, while the other
                 * group's forces are seen as they were at the start of the sweep (block Jacobi between the two groups; for two blocks of a positive definite problem
                 * that always converges).  The groups interact only where a finger or a gripper-body proxy touches a cube; everywhere else this IS Gauss-Seidel. */
                const real *fsee = f;
                real fmix[MAX_ROWS];
                if (P->jacobi) {
                    for (int m = 0; m < nr; m++) fmix[m] = grp[m] == grp[i] ? f[m] : f_start[m];
                    fsee = fmix;
                }
                if ((P->cone == 3 || P->cone == 4 || P->cone == 5) && kind[i] == 2) continue;
                if ((P->cone == 3 || P->cone == 4 || P->cone == 5) && kind[i] == 1) { pgs_block_pg(A, nr, bvec, Rr, f, fsee, i, blkdim[i], rowmu[i], P->cone == 4 ? 1 : (P->cone == 5 ? 2 : 0), T->walls); continue; }
                if (P->cone == 1 && kind[i] == 2) continue;                      /* (handled with its block below) */
                if (P->cone == 1 && kind[i] == 1) { pgs_block_exact(A, nr, bvec, Rr, f, i, blkdim[i], rowmu[i]); continue; }
                real res = bvec[i] + Rr[i] * f[i];
                for (int j = 0; j < nr; j++) res += A[(size_t)i * nr + j] * fsee[j];
                real old = f[i];
                real nf = f[i] - res / (A[(size_t)i * nr + i] + Rr[i]);
                if (kind[i] != 2 && nf < 0) nf = 0; /* unilateral rows */
                f[i] = nf;
                (void)old;
                if (kind[i] == 2 && i == blk0[i] + blkdim[i] - 1) {
                    /* last friction row of this contact: project onto the elliptic cone (D2) */
                    const int i0 = blk0[i], dm = blkdim[i];
                    const double *mu = rowmu[i];
                    real fn = f[i0], s2 = 0;
                    for (int r = 1; r < dm; r++) { real x = f[i0 + r] / (real)mu[r - 1]; s2 += x * x; }
                    if (fn <= 0) { for (int r = 1; r < dm; r++) f[i0 + r] = 0; }
                    else if (s2 > fn * fn) {
                        real sc = fn / (real)sqrt((double)s2);
                        for (int r = 1; r < dm; r++) f[i0 + r] *= sc;
                    }
                }
            }
            sweeps++;
            for (int i = 0; i < nr; i++) {
                if (fabs((double)f[i]) > fmaxabs) fmaxabs = fabs((double)f[i]);
                if (fabs((double)(f[i] - f_start[i])) > lastchange) lastchange = fabs((double)(f[i] - f_start[i]));
            }
            if (adaptive && lastchange <= P->pgs_tol * (1.0 + fmaxabs)) break;
        }
        for (int i = 0; i < nr; i++)
            for (int d = 0; d < nv; d++) qfc[d] += J[(size_t)i * nv + d] * f[i];
        if (use_x) {   /* integrate with the primal iterate itself: qfrc_constraint := M (x - a0) (= J'f at the optimum) */
            for (int d = 0; d < nv; d++) { real acc = 0; for (int j = 0; j < nv; j++) acc += M[d * nv + j] * (xsol[j] - a0[j]); qfc[d] = acc; }
        }
        if (lag->want_kkt) {
            const double k = kkt_residual(A, nr, bvec, Rr, f, kind, blkdim, rowmu);
            if (k > lag->kkt) lag->kkt = k;
        }
        free(MiJt); free(A);
        {   /* study aid (tools/solver_modes_study.py): ORC_SWEEP_SUM=1 makes max_sweeps the SUM over the substeps of a control step */
            static int sum_mode = -1;
            if (sum_mode < 0) sum_mode = getenv("ORC_SWEEP_SUM") != NULL;
            if (sum_mode) lag->max_sweeps += (uint32_t)sweeps;
            else if ((uint32_t)sweeps > lag->max_sweeps) lag->max_sweeps = (uint32_t)sweeps;
        }
        if (diag) { g_diag_res = lastchange; }
    }
    /* slots that are not active in this substep restart from zero */
    memset(warm, 0, sizeof *warm);
    for (int i = 0; i < nr; i++) *wptr[i] = f[i];
    if (diag) { g_diag_rows = nr; g_diag_contacts = ncon; if (nr == 0) g_diag_res = 0; }

    /* -- implicitfast: (M - h*D) qacc = qfrc_smooth + qfrc_constraint, D = d(passive+actuator)/dqvel
     *    = -(damping + kv) on the arm diagonal (the joint-level force clamp is ignored in D) */
    real Mh[ORC_NV_MAX * ORC_NV_MAX], rhs[ORC_NV_MAX];
    memcpy(Mh, M, sizeof(real) * nv * nv);
    for (int j = 0; j < 6; j++) Mh[j * nv + j] += h * (real)(DAMPING + KV);
    chol(Mh, nv);
    for (int d = 0; d < nv; d++) rhs[d] = tau[d] + qfc[d];
    chol_solve(Mh, nv, rhs);
    for (int d = 0; d < nv; d++) qvel[d] += h * rhs[d];
    for (int j = 0; j < 6; j++) qpos[j] += h * qvel[j];
    for (int c = 0; c < nc; c++) {
        real *pp = qpos + 6 + 7 * c, *qq = pp + 3, *vv = qvel + 6 + 6 * c, *ww = vv + 3;
        for (int k = 0; k < 3; k++) pp[k] += h * vv[k];
        /* MJ-DOC mju_quatIntegrate: q <- q * exp(h w / 2), w in body frame, then normalise */
        real wn = v3norm(ww);
        if (wn > 0) {
            real ang = h * wn, s = (real)sin((double)ang * 0.5) / wn, cw = (real)cos((double)ang * 0.5);
            real dq[4] = {cw, ww[0] * s, ww[1] * s, ww[2] * s};
            real r0 = qq[0] * dq[0] - qq[1] * dq[1] - qq[2] * dq[2] - qq[3] * dq[3];
            real r1 = qq[0] * dq[1] + qq[1] * dq[0] + qq[2] * dq[3] - qq[3] * dq[2];
            real r2 = qq[0] * dq[2] - qq[1] * dq[3] + qq[2] * dq[0] + qq[3] * dq[1];
            real r3 = qq[0] * dq[3] + qq[1] * dq[2] - qq[2] * dq[1] + qq[3] * dq[0];
            real n2 = (real)sqrt((double)(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3));
            qq[0] = r0 / n2; qq[1] = r1 / n2; qq[2] = r2 / n2; qq[3] = r3 / n2;
        }
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* glue: reach_cube_env.py:141-348 and the per-task deltas                                          */
/* ------------------------------------------------------------------------------------------------ */
static void params_base(orc_params *p, int task) {   /* the reference's constructor defaults + the rounds 1-4 solver settings (preset "fast") */
    memset(p, 0, sizeof *p);
    p->task = task;
    p->action_mode = ORC_ACTION_JOINT;   /* reach:80 */
    p->reward_type = ORC_REWARD_SPARSE;  /* reach:81 */
    p->block_gripper = (task == ORC_TASK_REACH || task == ORC_TASK_PUSH || task == ORC_TASK_PUSH_LOOP) ? 1 : 0; /* reach:82 push:84 lift:82 loop:80 */
    p->distance_threshold = 0.05;        /* reach:83 */
    p->cube_xy_range = 0.3;              /* reach:84 */
    p->target_xy_range = 0.3;            /* push:87 */
    p->goal_z_range = 0.1;               /* pick_place:88 */
    p->height_threshold = 0.1;           /* lift:84 */
    p->n_substeps = 20;                  /* reach:85 */
    p->max_episode_steps = 50;           /* gym_lowcostrobot/__init__.py:12-42 */
    p->impratio = 100.0;                 /* follower.xml:3 (after the scene's own <option>; later wins, MJ-DOC unverified) */
    p->pgs_iters = 4;
    p->compat = 0;
    p->auto_reset = 1;
    p->warm_start = 1;
    p->proxy_groups = 1;
    p->arm_collision = 1;
    p->pgs_tol = 1e-6;
    p->cc_points = 4;
    /* what the kernels run: block projected gradient in the second-order-cone variables with the rows in two concurrently swept groups (round 4) -- except
     * PushCubeLoop, which keeps the row-wise Gauss-Seidel sweeps with the radial projection, one sequence (lcr_kernels_loop.hip; deviation D2) */
    p->cone = task == ORC_TASK_PUSH_LOOP ? 0 : 3;
    p->pgs_cap = 0;   /* 50 */
    p->solver = 0;    /* PGS (what the kernels run) */
    p->jacobi = task == ORC_TASK_PUSH_LOOP ? 0 : 1;    /* two sweep groups (arm-only rows | cube rows) that sweep concurrently: what the kernels' two waves do */
    p->condim6 = (task == ORC_TASK_PUSH_LOOP || task == ORC_TASK_STACK) ? 1 : 0; /* as lcr_config_default: rolling rows where they matter (D4) */
    p->newton_iters = 30; p->ls_iters = 8; p->newton_tol = 1e-6; p->ls_tol = 1e-2;   /* (read by solver = 2 only; = lcr_config_default.  ls_tol: MuJoCo's own default ls_tolerance is 0.01) */
}
void orc_default_params(orc_params *p, int task) { orc_preset_params(p, task, ORC_PRESET_FAITHFUL); }
void orc_preset_params(orc_params *p, int task, int preset) {
    params_base(p, task);
    if (preset == ORC_PRESET_FAITHFUL) {
        p->solver = 2;                       /* Newton on the primal: MuJoCo's default solver (follower.xml:3 names none) */
        p->condim6 = 2;                      /* follower.xml:15 condim="6" on every finger contact */
        p->cc_points = 8;                    /* as many points as MuJoCo's box-box collider may return (stack_two_cubes.xml:25-35) */
        p->finger_geom = 1;                  /* the finger pads as boxes fitted to the tips of the collision hulls (follower.xml:15,89,97) instead of inscribed spheres */
    }
}
int orc_nq(int task) { return task == ORC_TASK_STACK ? 20 : 13; }
int orc_nv(int task) { return task == ORC_TASK_STACK ? 18 : 12; }
int orc_action_dim(const orc_params *p) { /* reach:95-96 */
    return (p->action_mode == ORC_ACTION_EE ? 3 : 5) + (p->block_gripper ? 0 : 1);
}
static int gripper_active(const orc_params *p) { return !(p->task == ORC_TASK_REACH || p->task == ORC_TASK_PUSH || p->task == ORC_TASK_PUSH_LOOP); }

static float clip1(float a) { return a < -1.0f ? -1.0f : (a > 1.0f ? 1.0f : a); } /* reach:234 */

/* joint-mode target (reach:248-268; lift:258-277) */
void orc_joint_ctrl(const orc_params *p, const double *q6, const float *action, double *ctrl6) {
    int k = orc_action_dim(p);
    for (int j = 0; j < 5; j++) ctrl6[j] = clampd((double)clip1(action[j]) + q6[j], TGT_LO[j], TGT_HI[j]);
    if (gripper_active(p)) ctrl6[5] = clampd((double)clip1(action[k - 1]) + q6[5], TGT_LO[5], TGT_HI[5]); /* lift:264,274: action[-1] */
    else ctrl6[5] = 0.0;                                                                                     /* reach:255,265 */
}

/* ee-mode glue of apply_action: target = site_xpos + float32(action[:3] * 0.05), z clamped at 0 (reach:236-242); gripper
 * target = clip(qpos[5] + float32(action[3] * 0.2), ctrlrange) for the gripper tasks (lift:253-257), 0 otherwise (reach:247) */
void orc_ee_glue(const orc_params *p, const double *site3, double q_gripper, const float *action, double *target3, double *grip) {
    for (int i = 0; i < 3; i++) target3[i] = site3[i] + (double)(clip1(action[i]) * 0.05f); /* float32 product, float64 sum */
    if (target3[2] < 0) target3[2] = 0;
    if (gripper_active(p)) *grip = clampd(q_gripper + (double)(clip1(action[3]) * 0.2f), JNT_LO[5], JNT_HI[5]);
    else *grip = 0.0;
}

/* inverse_kinematics (reach:148-221) incl. REF-QUIRK-3: writes the sim's qpos */
static int ik_solve(real *q_state /*in: qpos[:6], out: teleported*/, const real *target, real *q_ctrl, real *site_last) {
    real q[6];
    int iters = 0;
    memcpy(q, q_state, sizeof q); /* reach:182 */
    for (int it = 0; it < 10; it++) { /* max_iter reach:156 */
        kin_t K;
        memcpy(q_state, q, sizeof q); /* reach:185 */
        arm_kinematics(q, &K);        /* reach:186 mj_forward */
        v3copy(site_last, K.site);
        iters++;
        real e[3];
        v3sub(e, target, K.site); /* reach:189 */
        if (v3norm(e) < (real)0.01) break; /* reach:193 */
        real Jp[18], Jr[18];
        jac_point(&K, 6, 4, K.site, Jp, Jr); /* reach:197 mj_jacSite; column 6 is zero */
        real A[36], rhs[6];
        for (int a = 0; a < 6; a++) {
            for (int b = 0; b < 6; b++) {
                real s = 0;
                for (int k = 0; k < 3; k++) s += Jp[k * 6 + a] * Jp[k * 6 + b];
                A[a * 6 + b] = s + (a == b ? (real)0.15 : 0); /* lm_damping reach:155,200 */
            }
            rhs[a] = Jp[a] * e[0] + Jp[6 + a] * e[1] + Jp[12 + a] * e[2];
        }
        chol(A, 6);
        chol_solve(A, 6, rhs); /* == inv(JtJ + 0.15 I) Jt e, reach:201-202 ; nullspace term is x 0.0 (reach:205-207) */
        real nn = 0;
        for (int a = 0; a < 6; a++) nn += rhs[a] * rhs[a];
        nn = (real)sqrt((double)nn);
        if (nn > 1) for (int a = 0; a < 6; a++) rhs[a] /= nn; /* reach:210-212 */
        for (int a = 0; a < 6; a++) {
            q[a] += rhs[a] * (real)0.5; /* step reach:154,215 */
            if (q[a] < (real)JNT_LO[a]) q[a] = (real)JNT_LO[a]; /* check_joint_limits reach:141-146 */
            if (q[a] > (real)JNT_HI[a]) q[a] = (real)JNT_HI[a];
        }
    }
    memcpy(q_ctrl, q, sizeof q);
    return iters;
}

void orc_reward(const orc_params *p, const double *a3, const double *b3, float *reward32, double *reward64,
                uint8_t *success) {
    /* goal_distance/is_success/compute_reward reach:335-348 */
    double d = sqrt((a3[0] - b3[0]) * (a3[0] - b3[0]) + (a3[1] - b3[1]) * (a3[1] - b3[1]) + (a3[2] - b3[2]) * (a3[2] - b3[2]));
    *success = d < p->distance_threshold;
    if (p->reward_type == ORC_REWARD_SPARSE) {
        *reward32 = -(float)(d > p->distance_threshold); /* -0.0f when within threshold (REF-QUIRK-7) */
        *reward64 = (double)*reward32;
    } else {
        *reward64 = -d;
        *reward32 = (float)*reward64;
    }
}

/* ---- PushCubeLoop reward (push_cube_loop_env.py:334-383), with numpy's scalar promotion rules (NEP 50) emulated:
 * np.float32 (op) python-float -> float32 ; np.float32 (op) np.float64 -> float64 ; python min/max return one of their
 * arguments unchanged (the first one on ties). */
typedef struct { double v; int f32; } npnum;
static npnum np32(float x) { npnum r = {(double)x, 1}; return r; }
static npnum np64(double x) { npnum r = {x, 0}; return r; }
static npnum np_bin(npnum a, npnum b, char op) {
    npnum r;
    r.f32 = a.f32 && b.f32;
    if (r.f32) {
        float x = (float)a.v, y = (float)b.v, z = op == '+' ? x + y : (op == '-' ? x - y : (op == '*' ? x * y : x / y));
        r.v = (double)z;
    } else r.v = op == '+' ? a.v + b.v : (op == '-' ? a.v - b.v : (op == '*' ? a.v * b.v : a.v / b.v));
    return r;
}
static npnum np_min(npnum a, npnum b) { return b.v < a.v ? b : a; }
static npnum np_max(npnum a, npnum b) { return b.v > a.v ? b : a; }
void orc_loop_reward(const float *cube, int32_t *goal, double *overlap_out, double *reward, uint8_t *success) {
    const double gh0 = 0.035 / 2 - 0.008, gh1 = 0.045 / 2 - 0.008; /* goal_region_high[:2] (:133-134) */
    const double gl1 = gh1 * -1.0;
    const double cx = *goal == 0 ? 0.06 : -0.06, cy = 0.135;
    const float wc = (float)(0.015 / 2); /* cube_size as a weak python float next to a float32 scalar */
    npnum xc = np32(cube[0]), yc = np32(cube[1]), w32 = np32(wc);
    npnum zero = {0.0, 1}; /* python int 0: weakest type, keep f32-ness of the other operand */
    npnum xo = np_max(zero, np_bin(np_min(np_bin(xc, w32, '+'), np64(cx + gh0)), np_max(np_bin(xc, w32, '-'), np64(cx - gh0)), '-'));
    npnum yo = np_max(zero, np_bin(np_min(np_bin(yc, w32, '+'), np64(cy + gh1)), np_max(np_bin(yc, w32, '-'), np64(cy - gh1)), '-'));
    /* python `max(0, v)` returns the int 0 when v <= 0: an exact zero of the weakest type */
    if (!(xo.v > 0)) { xo.v = 0; xo.f32 = 1; }
    if (!(yo.v > 0)) { yo.v = 0; yo.f32 = 1; }
    npnum area = np_bin(xo, yo, '*');
    const double cube_area = (0.015 / 2) * (0.015 / 2) * 4; /* python floats */
    double overlap = area.f32 ? (double)((float)area.v / (float)cube_area) : area.v / cube_area;
    if (xo.v == 0 || yo.v == 0) overlap = 0;
    *overlap_out = overlap;
    *success = 0;
    if (overlap > 0.95) { *success = 1; *reward = 5; *goal = 1 - *goal; }
    else if (overlap > 0.0) *reward = area.f32 ? (double)((float)overlap - 1.0f) : overlap - 1;
    else {
        double edge = gl1 + (*goal == 0 ? 0.135 : 0.135); /* goal_region_low[1] + centre y of the current goal (:352-356) */
        double d = sqrt(((double)cube[1] - edge) * ((double)cube[1] - edge));
        double r = (-d / 0.16) - 1;
        if (r < -2) r = -2;
        if (r > -1) r = -1;
        *reward = r;
    }
}

/* ---- numpy-compatible RNG ---- */
typedef unsigned __int128 u128;
#define PCG_MULT (((u128)0x2360ED051FC65DA4ULL << 64) | 0x4385DF649FCCF645ULL)
static uint32_t ss_hashmix(uint32_t v, uint32_t *hc) {
    v ^= *hc; *hc *= 0x931e8875u; v *= *hc; v ^= v >> 16; return v;
}
static uint32_t ss_mix(uint32_t x, uint32_t y) {
    uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y; r ^= r >> 16; return r;
}
void orc_rng_seed(uint64_t seed, uint64_t rng[4]) {
    /* numpy SeedSequence(seed).generate_state(4, uint64) then PCG64 srandom (numpy/random/_pcg64.pyx, bit_generator.pyx) */
    uint32_t ent[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    int nent = ent[1] ? 2 : 1;
    uint32_t pool[4], hc = 0x43b0d7e5u;
    for (int i = 0; i < 4; i++) pool[i] = ss_hashmix(i < nent ? ent[i] : 0u, &hc);
    for (int s = 0; s < 4; s++)
        for (int d = 0; d < 4; d++)
            if (s != d) pool[d] = ss_mix(pool[d], ss_hashmix(pool[s], &hc));
    uint32_t w[8], hb = 0x8b51f9ddu;
    for (int i = 0; i < 8; i++) {
        uint32_t v = pool[i & 3];
        v ^= hb; hb *= 0x58f38dedu; v *= hb; v ^= v >> 16; w[i] = v;
    }
    uint64_t s64[4];
    for (int i = 0; i < 4; i++) s64[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    u128 initstate = ((u128)s64[0] << 64) | s64[1], initseq = ((u128)s64[2] << 64) | s64[3];
    u128 inc = (initseq << 1) | 1, st = 0;
    st = st * PCG_MULT + inc;
    st += initstate;
    st = st * PCG_MULT + inc;
    rng[0] = (uint64_t)(st >> 64); rng[1] = (uint64_t)st; rng[2] = (uint64_t)(inc >> 64); rng[3] = (uint64_t)inc;
}
double orc_rng_double(uint64_t rng[4]) {
    u128 st = ((u128)rng[0] << 64) | rng[1], inc = ((u128)rng[2] << 64) | rng[3];
    st = st * PCG_MULT + inc;
    rng[0] = (uint64_t)(st >> 64); rng[1] = (uint64_t)st;
    uint64_t hi = rng[0], lo = rng[1], x = hi ^ lo;
    unsigned rot = (unsigned)(hi >> 58);
    uint64_t out = (x >> rot) | (x << ((64 - rot) & 63));
    return (double)(out >> 11) * (1.0 / 9007199254740992.0);
}

/* reset(): reach:297-311, push:308-328, pick_place:316-336, stack:307-324 */
static void reset_one(const orc_params *P, const task_model *T, double *qpos, double *qvel, double *ee_lag, float *target,
                      int32_t *elapsed, uint64_t *rng, int goal) {
    double lo[3] = {-P->cube_xy_range / 2, -P->cube_xy_range / 2, 0}, hi[3] = {P->cube_xy_range / 2, P->cube_xy_range / 2, 0};
    lo[1] += 0.165; hi[1] += 0.10; /* reach:138-139 */
    if (P->task == ORC_TASK_PUSH_LOOP) { /* push_cube_loop_env.py:304-308, constants :130-135 from push_cube_loop.xml:38,41 */
        double gh[3] = {0.035 / 2, 0.045 / 2, 0.007 / 2};
        gh[0] -= 0.008; gh[1] -= 0.008;
        double gl[3] = {gh[0] * -1.0, gh[1] * -1.0, gh[2] * 1.0};
        double c1[2] = {0.06, 0.135}, c2[2] = {-0.06, 0.135};
        for (int k = 0; k < 3; k++) lo[k] = gl[k], hi[k] = gh[k];
        for (int k = 0; k < 3; k++) qpos[6 + k] = lo[k] + (hi[k] - lo[k]) * orc_rng_double(rng);
        for (int k = 0; k < 2; k++) qpos[6 + k] += (1 - goal) * c1[k] + goal * c2[k];
        qpos[9] = 1; qpos[10] = 0; qpos[11] = 0; qpos[12] = 0;
    } else
    for (int c = 0; c < T->ncube; c++) {
        for (int k = 0; k < 3; k++) qpos[6 + 7 * c + k] = lo[k] + (hi[k] - lo[k]) * orc_rng_double(rng); /* np_random.uniform */
        qpos[6 + 7 * c + 3] = 1; qpos[6 + 7 * c + 4] = 0; qpos[6 + 7 * c + 5] = 0; qpos[6 + 7 * c + 6] = 0;
    }
    if (T->has_target) {
        double tl[3] = {-P->target_xy_range / 2, -P->target_xy_range / 2, 0};
        double th[3] = {P->target_xy_range / 2, P->target_xy_range / 2, P->task == ORC_TASK_PICK_PLACE ? P->goal_z_range : 0.0};
        tl[1] += 0.165; th[1] += 0.10; /* push:147-148 */
        for (int k = 0; k < 3; k++) target[k] = (float)(tl[k] + (th[k] - tl[k]) * orc_rng_double(rng)); /* .astype(float32) push:320 */
    }
    for (int j = 0; j < 6; j++) qpos[j] = 0;
    if (P->compat & ORC_COMPAT_ZERO_QVEL_ON_RESET) for (int d = 0; d < ORC_NV_MAX; d++) qvel[d] = 0; /* default: REF-QUIRK-1 keep */
    /* mj_forward (reach:309) refreshes site_xpos at q=0 */
    real q0[6] = {0, 0, 0, 0, 0, 0};
    kin_t K;
    arm_kinematics(q0, &K);
    for (int k = 0; k < 3; k++) ee_lag[k] = (double)K.site[k];
    *elapsed = 0;
}

static void write_obs(const orc_params *P, const task_model *T, const double *qpos, const double *qvel, const float *target,
                      float *obs) { /* get_observation reach:281-295 push:291-306 stack:290-305 */
    for (int j = 0; j < 6; j++) { obs[j] = (float)qpos[j]; obs[6 + j] = (float)qvel[j]; }
    for (int k = 0; k < 3; k++) obs[12 + k] = (float)qpos[6 + k];
    for (int k = 0; k < 3; k++)
        obs[15 + k] = T->has_target ? target[k] : (P->task == ORC_TASK_STACK ? (float)qpos[13 + k] : 0.0f);
}

void orc_reset(const orc_params *p, orc_io *io, int n, const uint8_t *mask, const uint64_t *seeds) {
    task_model T = get_task_model(p->task);
    for (int e = 0; e < n; e++) {
        if (mask && !mask[e]) continue;
        if (seeds) orc_rng_seed(seeds[e], io->rng + 4 * (size_t)e);
        if (io->warm) memset((char *)io->warm + (size_t)e * sizeof(warm_t), 0, sizeof(warm_t));
        reset_one(p, &T, io->qpos + (size_t)e * ORC_NQ_MAX, io->qvel + (size_t)e * ORC_NV_MAX, io->ee_lag + 3 * (size_t)e,
                  io->target + 3 * (size_t)e, io->elapsed + e, io->rng + 4 * (size_t)e, io->goal ? io->goal[e] : 0);
        if (io->obs) write_obs(p, &T, io->qpos + (size_t)e * ORC_NQ_MAX, io->qvel + (size_t)e * ORC_NV_MAX, io->target + 3 * (size_t)e,
                               io->obs + 18 * (size_t)e);
    }
}

static void step_one(const orc_params *P, const task_model *T, orc_io *io, size_t e, const float *action) {
    double *qpos64 = io->qpos + e * ORC_NQ_MAX, *qvel64 = io->qvel + e * ORC_NV_MAX, *ee_lag = io->ee_lag + 3 * e;
    float *target = io->target + 3 * e;
    const int nq = orc_nq(P->task), nv = orc_nv(P->task), k = orc_action_dim(P);
    real qpos[ORC_NQ_MAX], qvel[ORC_NV_MAX], ctrl[6];
    int ik_iters = 0;
    for (int i = 0; i < nq; i++) qpos[i] = (real)qpos64[i];
    for (int i = 0; i < nv; i++) qvel[i] = (real)qvel64[i];

    /* ---- apply_action reach:223-273 */
    if (P->action_mode == ORC_ACTION_EE) {
        real tgt[3], qc[6], sl[3];
        double t64[3], g64;
        orc_ee_glue(P, ee_lag, 0.0, action, t64, &g64); /* target only; the gripper needs the post-IK qpos[5] */
        for (int i = 0; i < 3; i++) tgt[i] = (real)t64[i];
        ik_iters = ik_solve(qpos, tgt, qc, sl);
        for (int j = 0; j < 6; j++) ctrl[j] = qc[j];
        orc_ee_glue(P, ee_lag, (double)qpos[5], action, t64, &g64);
        ctrl[5] = (real)g64;
    } else {
        double q6[6], c6[6];
        for (int j = 0; j < 6; j++) q6[j] = (double)qpos[j];
        orc_joint_ctrl(P, q6, action, c6);
        for (int j = 0; j < 6; j++) ctrl[j] = (real)c6[j];
    }
    (void)k;
    /* ---- 20 x mj_step reach:276-279 */
    lag_t lag;
    warm_t warm_local, *warmp = &warm_local;
    memset(&lag, 0, sizeof lag);
    lag.want_kkt = io->kkt != NULL;
    memset(&warm_local, 0, sizeof warm_local);
    /* The constraint forces of the last substep warm-start the first substep of the next control step (io->warm, one warm_t per env, zero
     * after reset), as MuJoCo carries mjData.qacc_warmstart across mj_step calls and the reference never resets it between env.step calls.
     * ORC_COMPAT_COLD_SOLVE_EACH_STEP (or io->warm == NULL): every control step starts from zero forces instead. */
    if (io->warm && !(P->compat & ORC_COMPAT_COLD_SOLVE_EACH_STEP)) warmp = (warm_t *)((char *)io->warm + e * sizeof(warm_t));
    for (int s = 0; s < P->n_substeps; s++) {
        t_trace_slot = (g_newton_trace && s < g_newton_trace_sub) ? g_newton_trace + ((size_t)e * g_newton_trace_sub + s) * 2 : NULL;
        substep(P, T, qpos, qvel, ctrl, &lag, warmp, e == 0 && s == P->n_substeps - 1, s);
    }
    t_trace_slot = NULL;
    for (int i = 0; i < nq; i++) qpos64[i] = (double)qpos[i];
    for (int i = 0; i < nv; i++) qvel64[i] = (double)qvel[i];
    for (int i = 0; i < 3; i++) ee_lag[i] = (double)lag.ee[i];
    if (io->sim_time) io->sim_time[e] += P->n_substeps * H_STEP; /* data.time advances in mj_step only */
    if (io->kkt) io->kkt[e] = lag.kkt;
    if (io->active_mask) {
        io->active_mask[e] = lag.active_mask; io->active_count[e] = lag.active_count; io->max_sweeps[e] = lag.max_sweeps;
        io->choice[e] = lag.choice + (uint32_t)ik_iters * 0x9E3779B1u;
    }

    /* ---- observation, reward, termination: reach:313-333 (+ lift:322-346 push:330-346 stack:326-348) */
    float *obs = io->obs + 18 * e;
    write_obs(P, T, qpos64, qvel64, target, obs);
    double a3[3], b3[3];
    float r32 = 0; double r64 = 0; uint8_t succ = 0, term = 0;
    double cube[3] = {(double)lag.cube[0][0], (double)lag.cube[0][1], (double)lag.cube[0][2]};
    switch (P->task) {
    case ORC_TASK_REACH:
        for (int i = 0; i < 3; i++) { a3[i] = ee_lag[i]; b3[i] = cube[i]; }
        orc_reward(P, a3, b3, &r32, &r64, &succ); term = succ; break;
    case ORC_TASK_LIFT: { /* lift:341-345 REF-QUIRK-5 */
        double d = sqrt((ee_lag[0] - cube[0]) * (ee_lag[0] - cube[0]) + (ee_lag[1] - cube[1]) * (ee_lag[1] - cube[1]) +
                        (ee_lag[2] - cube[2]) * (ee_lag[2] - cube[2]));
        r64 = (cube[2] - P->height_threshold) + d; r32 = (float)r64; succ = 0; term = 0; break; }
    case ORC_TASK_PUSH: case ORC_TASK_PICK_PLACE:
        for (int i = 0; i < 3; i++) { a3[i] = cube[i]; b3[i] = (double)target[i]; }
        orc_reward(P, a3, b3, &r32, &r64, &succ); term = succ; break;
    case ORC_TASK_PUSH_LOOP: { /* push_cube_loop_env.py:324-331: reward from the FRESH qpos cast to float32, never terminates */
        float cpos[3] = {(float)qpos64[6], (float)qpos64[7], (float)qpos64[8]};
        double ov;
        orc_loop_reward(cpos, io->goal + e, &ov, &r64, &succ);
        r32 = (float)r64; term = 0; break; }
    case ORC_TASK_STACK: /* stack:334-347 */
        for (int i = 0; i < 3; i++) { a3[i] = (double)lag.cube[1][i]; b3[i] = (double)lag.cube[0][i]; }
        b3[2] += 0.03;
        orc_reward(P, a3, b3, &r32, &r64, &succ); term = succ; break;
    }
    /* failure containment (MJ-DOC: mj_checkPos / mj_checkVel reset an unstable simulation): NaN / inf / |x| >= 2^34 */
    int diverged = 0;
    for (int i = 0; i < nq; i++) if (!(fabs(qpos64[i]) < 17179869184.0)) diverged = 1;
    for (int i = 0; i < nv; i++) if (!(fabs(qvel64[i]) < 17179869184.0)) diverged = 1;
    if (diverged) { r32 = -1.0f; r64 = -1.0; succ = 0; term = 0; }
    io->elapsed[e] += 1;
    uint8_t trunc = diverged || (P->max_episode_steps > 0 && io->elapsed[e] >= P->max_episode_steps); /* gymnasium TimeLimit; <=0 disables */
    io->reward[e] = r32; io->reward64[e] = r64; io->terminated[e] = term; io->truncated[e] = trunc; io->is_success[e] = succ;
    memcpy(io->term_obs + 18 * e, obs, sizeof(float) * 18);
    io->did_reset[e] = 0;
    if (diverged || (P->auto_reset && (term || trunc))) { /* SB3 DummyVecEnv.step_wait semantics (examples/gym_manipulation_sb3.py:34-39) */
        reset_one(P, T, qpos64, qvel64, ee_lag, target, io->elapsed + e, io->rng + 4 * e, io->goal ? io->goal[e] : 0);
        if (diverged) for (int d = 0; d < ORC_NV_MAX; d++) qvel64[d] = 0;
        write_obs(P, T, qpos64, qvel64, target, obs);
        io->did_reset[e] = 1;
        if (io->warm) memset((char *)io->warm + e * sizeof(warm_t), 0, sizeof(warm_t)); /* no forces carried into a new episode */
    }
}

void orc_step(const orc_params *p, orc_io *io, int n, const float *action, int threads) {
    task_model T = get_task_model(p->task);
    const int k = orc_action_dim(p);
    ensure_invweight0();
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads) schedule(static)
#endif
    for (int e = 0; e < n; e++) step_one(p, &T, io, (size_t)e, action + (size_t)e * k);
    (void)threads;
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_last_diag(int *n_rows, int *n_contacts, double *pgs_residual) {
    *n_rows = g_diag_rows; *n_contacts = g_diag_contacts; *pgs_residual = g_diag_res;
}

/* the oracle's own L0 tables, flattened, for the test that compares them with tests/golden/model_golden.json:
 * per link i (6): pos3, axis3, ipos3, iquat4, mass, diaginertia3, range2 = 19 doubles; then site3, armature, damping, kp, kv,
 * frcrange, timestep, cube_half; then per task (6): cube mass, cube inertia, mu_tan, mu_tors; then walls (5: inner faces x, y0, y1, top, thickness);
 * then NSPH x (link, pos3, rad) and NLPX x (link, pos3, rad, group).  Returns the number of doubles written. */
int orc_model_table(double *out) {
    int n = 0;
    for (int i = 0; i < 6; i++) {
        for (int k = 0; k < 3; k++) out[n++] = LINK_POS[i][k];
        for (int k = 0; k < 3; k++) out[n++] = LINK_AXIS[i][k];
        for (int k = 0; k < 3; k++) out[n++] = LINK_IPOS[i][k];
        for (int k = 0; k < 4; k++) out[n++] = LINK_IQUAT[i][k];
        out[n++] = LINK_MASS[i];
        for (int k = 0; k < 3; k++) out[n++] = LINK_DIAGI[i][k];
        out[n++] = JNT_LO[i]; out[n++] = JNT_HI[i];
    }
    for (int k = 0; k < 3; k++) out[n++] = SITE_POS[k];
    out[n++] = ARMATURE; out[n++] = DAMPING; out[n++] = KP; out[n++] = KV; out[n++] = FRC_LIM; out[n++] = H_STEP; out[n++] = CUBE_HALF;
    for (int t = 0; t < 6; t++) {
        task_model T = get_task_model(t);
        out[n++] = T.cube_mass; out[n++] = T.cube_inertia; out[n++] = T.mu_cube[0]; out[n++] = T.mu_cube[2];
    }
    out[n++] = WALL_X; out[n++] = WALL_Y0; out[n++] = WALL_Y1; out[n++] = WALL_TOP; out[n++] = WALL_THICK;
    /* finger geom class (follower.xml:15): solimp d0, dwidth-limit, width; friction (tangential, torsional, rolling) */
    out[n++] = SOLIMP_FINGER[0]; out[n++] = SOLIMP_FINGER[1]; out[n++] = SOLIMP_FINGER[2];
    out[n++] = MU_FINGER[0]; out[n++] = MU_FINGER[2]; out[n++] = MU_FINGER[3];
    for (int s = 0; s < NSPH; s++) { out[n++] = SPH_LINK[s]; for (int k = 0; k < 3; k++) out[n++] = SPH_POS[s][k]; out[n++] = SPH_RAD[s]; }
    for (int s = 0; s < NLPX; s++) { out[n++] = LPX_LINK[s]; for (int k = 0; k < 3; k++) out[n++] = LPX_POS[s][k]; out[n++] = LPX_RAD[s]; out[n++] = LPX_CUBE[s]; }
    return n;
}

/* ---- model queries ---- */
/* test access to world_surface (the floor and PushCubeLoop's rail boxes as the arm sees them): depth, normal[3], code */
double orc_world_surface(const double *p3, double r, int walls, double *n3, int *code) {
    real p[3] = {(real)p3[0], (real)p3[1], (real)p3[2]}, n[3];
    const real d = world_surface(p, (real)r, walls, n, code);
    for (int k = 0; k < 3; k++) n3[k] = (double)n[k];
    return (double)d;
}
/* the finger pad boxes (finger_geom = 1): NSPH x (centre3, half3), link frames of SPH_LINK */
void orc_pad_table(double *out) {
    for (int s = 0; s < NSPH; s++) for (int k = 0; k < 3; k++) { out[6 * s + k] = PAD_C[s][k]; out[6 * s + 3 + k] = PAD_H[s][k]; }
}
/* world frames of the six links: rotation matrices (row-major 3 x 3, columns = the link's axes) and origins */
void orc_link_frames(const double *q6, double *R54, double *p18) {
    real q[6];
    kin_t K;
    for (int i = 0; i < 6; i++) q[i] = (real)q6[i];
    K.ncube = 0;
    arm_kinematics(q, &K);
    for (int i = 0; i < 6; i++) {
        for (int k = 0; k < 9; k++) R54[9 * i + k] = (double)K.R[i + 1][k];
        for (int k = 0; k < 3; k++) p18[3 * i + k] = (double)K.p[i + 1][k];
    }
}
void orc_fk(const double *q6, double *link_pos, double *site, double *spheres) {
    real q[6]; kin_t K;
    for (int j = 0; j < 6; j++) q[j] = (real)q6[j];
    arm_kinematics(q, &K);
    for (int i = 0; i < 6; i++) for (int k = 0; k < 3; k++) link_pos[3 * i + k] = (double)K.p[i + 1][k];
    for (int k = 0; k < 3; k++) site[k] = (double)K.site[k];
    if (spheres) for (int s = 0; s < NSPH; s++) for (int k = 0; k < 3; k++) spheres[3 * s + k] = (double)K.sph[s][k];
}
int orc_proxies(const double *q6, double *centres /*[NLPX][3]*/, double *radii /*[NLPX]*/) {
    real q[6]; kin_t K;
    for (int j = 0; j < 6; j++) q[j] = (real)q6[j];
    arm_kinematics(q, &K);
    for (int s = 0; s < NLPX; s++) { for (int k = 0; k < 3; k++) centres[3 * s + k] = (double)K.lpx[s][k]; radii[s] = LPX_RAD[s]; }
    return NLPX;
}
void orc_mass_matrix(const double *q6, int with_armature, double *M) {
    real q[6], Mr[36]; kin_t K;
    for (int j = 0; j < 6; j++) q[j] = (real)q6[j];
    arm_kinematics(q, &K);
    arm_mass(&K, with_armature, Mr);
    for (int i = 0; i < 36; i++) M[i] = (double)Mr[i];
}
void orc_bias(const double *q6, const double *qd6, double *bias) {
    real q[6], qd[6], b[6]; kin_t K;
    for (int j = 0; j < 6; j++) { q[j] = (real)q6[j]; qd[j] = (real)qd6[j]; }
    arm_kinematics(q, &K);
    arm_bias(&K, qd, b);
    for (int j = 0; j < 6; j++) bias[j] = (double)b[j];
}
void orc_site_jac(const double *q6, double *J) {
    real q[6], Jp[18], Jr[18]; kin_t K;
    for (int j = 0; j < 6; j++) q[j] = (real)q6[j];
    arm_kinematics(q, &K);
    jac_point(&K, 6, 4, K.site, Jp, Jr);
    for (int i = 0; i < 18; i++) J[i] = (double)Jp[i];
}
void orc_invweight0(double *body_tran, double *body_rot, double *dof) {
    ensure_invweight0();
    for (int i = 0; i < 6; i++) { body_tran[i] = g_inv_tran[i]; body_rot[i] = g_inv_rot[i]; dof[i] = g_inv_dof[i]; }
}
int orc_ik(const double *q6_in, const double *target3, double *q_ctrl, double *q_state, double *site_last) {
    real qs[6], t[3], qc[6], sl[3];
    for (int j = 0; j < 6; j++) qs[j] = (real)q6_in[j];
    for (int k = 0; k < 3; k++) t[k] = (real)target3[k];
    int it = ik_solve(qs, t, qc, sl);
    for (int j = 0; j < 6; j++) { q_ctrl[j] = (double)qc[j]; q_state[j] = (double)qs[j]; }
    for (int k = 0; k < 3; k++) site_last[k] = (double)sl[k];
    return it;
}
