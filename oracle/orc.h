/*
 * oracle/orc.h -- CPU oracle for the rigid-body step hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product (libode_mi355*.so / the rl-ode-physics_amd package).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
 * and there only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the arithmetic of this path lives in third-party Open
 * Dynamics Engine (libode), which the reference includes as a system header
 * (/root/reference/src/main.c:11, inc/body.h:4) but does not vendor or pin,
 * and which is not installed here.  The reference holds no test or golden
 * vector for the path.  This file is therefore a plain-C
 * restatement of ODE's published QuickStep algorithm as recalled from the
 * public ODE 0.13-0.16 sources (tagged [ODE-recall]), anchored on the
 * reference's own call sites (tagged main.c:LINE), and pinned by the analytic
 * known-answer tests in tests/test_oracle_kat.py -- not by reference outputs.
 * The one compilable slice of the reference, src/rand.c, is pinned by the
 * golden values recorded in SURVEY.md section 8c (tests/golden/rand_golden.json).
 *
 * Precision: compile with -DORC_SINGLE for float (ODE dSINGLE), default double
 * (ODE dDOUBLE).  Compile with -ffp-contract=off so results do not depend on
 * the host's FMA availability.
 */
#ifndef ORC_H
#define ORC_H

#include <stdint.h>
#include <stddef.h>

#ifdef ORC_SINGLE
typedef float real;
#define ORC_CFM_DEFAULT 1e-5f   /* [ODE-recall] dWorldCreate: global_cfm, dSINGLE */
#else
typedef double real;
#define ORC_CFM_DEFAULT 1e-10   /* [ODE-recall] dWorldCreate: global_cfm, dDOUBLE */
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- geometry classes (subset the reference uses: main.c:717,720,743; CONVEX for BASELINE configs[4], which the
 *      reference itself never creates -- SURVEY.md F9) ---- */
enum { ORC_GEOM_SPHERE = 0, ORC_GEOM_BOX = 1, ORC_GEOM_PLANE = 2, ORC_GEOM_CONVEX = 3 };

/* ---- contact surface mode bits [ODE-recall contact.h] ---- */
enum { ORC_CONTACT_BOUNCE = 0x004 };

/* ---- SOR row ordering ---- */
enum {
    ORC_ORDER_FIXED = 0, /* rows solved in creation order, no reshuffle          */
    ORC_ORDER_ODE   = 1  /* [ODE-recall] head-inserted lists + dRandInt shuffle  */
                         /* every 8th iteration (RANDOMLY_REORDER_CONSTRAINTS)   */
};

/* ---- which stepper orc_world_tick runs: the reference calls dWorldStep (main.c:213); BASELINE names dWorldQuickStep ---- */
enum {
    ORC_STEPPER_QUICK = 0, /* dWorldQuickStep: 20 SOR sweeps                                              */
    ORC_STEPPER_EXACT = 1  /* dWorldStep: the island's boxed LCP  A lambda = b + w,  lo <= lambda <= hi,   */
                           /* solved to complementarity [ODE-recall step.cpp + lcp.cpp: same rows, same A] */
};

/* ---- gyroscopic torque form ---- */
enum {
    ORC_GYRO_OFF      = 0,
    ORC_GYRO_EXPLICIT = 1, /* tacc -= w x (I w)            [ODE-recall, <=0.12]   */
    ORC_GYRO_IMPLICIT = 2  /* Lacoursiere pseudo-tensor    [ODE-recall, 0.13+]    */
};

typedef struct orc_world orc_world;

typedef struct {
    real pos[3];
    real normal[3];
    real depth;
    int g1, g2;
} orc_contactgeom;

/* world ------------------------------------------------------------------ */
orc_world *orc_world_create(void);                       /* main.c:95 */
void orc_world_destroy(orc_world *w);                    /* main.c:266 */
void orc_world_set_gravity(orc_world *w, real x, real y, real z); /* main.c:96 */
void orc_world_set_erp(orc_world *w, real erp);
void orc_world_set_cfm(orc_world *w, real cfm);
void orc_world_set_quickstep(orc_world *w, int iters, real sor_w);
void orc_world_set_row_order(orc_world *w, int mode);
void orc_world_set_gyro_mode(orc_world *w, int mode);
void orc_world_set_stepper(orc_world *w, int mode);
int  orc_world_last_lcp_rounds(orc_world *w);
/* diagnostics of the last tick's contact joints, in creation order: count; per joint the bodies (b2 = -1: static geometry),
 * the contact (position, normal into b1, depth) and the normal force the step found */
int  orc_world_joint_count(orc_world *w);
void orc_world_joint_info(orc_world *w, int k, int *b1, int *b2, real pos[3], real normal[3], real *depth, real *lambda_n);
/* contact surface applied by the built-in near callback (main.c:684-687) */
void orc_world_set_surface(orc_world *w, int mode, real mu, real bounce, real bounce_vel);
void orc_world_set_max_contacts(orc_world *w, int n);    /* main.c:675 (8) */
void orc_world_set_broadphase(orc_world *w, int mode);   /* 0 auto, 1 sweep, 2 grid: same pair set either way */
void orc_rand_seed(uint32_t s);                          /* [ODE-recall] dRandSetSeed */

/* bodies ----------------------------------------------------------------- */
int  orc_body_create(orc_world *w);                      /* main.c:703; m=1, I=identity (F7) */
void orc_body_set_position(orc_world *w, int b, real x, real y, real z);   /* main.c:708 */
void orc_body_set_rotation(orc_world *w, int b, const real R[12]);         /* main.c:709 */
void orc_body_set_quaternion(orc_world *w, int b, const real q[4]);
void orc_body_set_linear_vel(orc_world *w, int b, real x, real y, real z);
void orc_body_set_angular_vel(orc_world *w, int b, real x, real y, real z);
void orc_body_set_mass(orc_world *w, int b, real mass, const real I[9]);   /* body-frame, row-major 3x3 */
void orc_body_add_force(orc_world *w, int b, real x, real y, real z);      /* main.c:532 (comment) */
void orc_body_add_torque(orc_world *w, int b, real x, real y, real z);
const real *orc_body_get_position(orc_world *w, int b);  /* main.c:229 */
const real *orc_body_get_rotation(orc_world *w, int b);  /* main.c:230; 3x4 row-major */
const real *orc_body_get_quaternion(orc_world *w, int b);
const real *orc_body_get_linear_vel(orc_world *w, int b);
const real *orc_body_get_angular_vel(orc_world *w, int b);

/* geoms ------------------------------------------------------------------ */
int  orc_geom_create_box(orc_world *w, real lx, real ly, real lz);         /* main.c:720,743 */
int  orc_geom_create_sphere(orc_world *w, real radius);                    /* main.c:717 */
int  orc_geom_create_plane(orc_world *w, real a, real b, real c, real d);
void orc_geom_set_body(orc_world *w, int g, int b);                        /* main.c:726 */
void orc_geom_set_position(orc_world *w, int g, real x, real y, real z);   /* main.c:748 */
void orc_geom_set_rotation(orc_world *w, int g, const real R[12]);         /* main.c:749 */
void orc_geom_set_category_bits(orc_world *w, int g, uint32_t bits);       /* main.c:724,751,752 */
void orc_geom_set_collide_bits(orc_world *w, int g, uint32_t bits);        /* main.c:725 */

/* narrowphase: dCollide(o1,o2,flags,contact,skip) (main.c:678) ------------- */
int orc_collide(orc_world *w, int g1, int g2, int max_contacts, orc_contactgeom *out);
/* test helper: body-less geoms g1, g2 placed at n poses (pos3 + R12 each; optional n x 3 sizes) in turn, collided each time */
void orc_collide_bulk(orc_world *w, int g1, int g2, int n, const real *pose1, const real *size1, const real *pose2, const real *size2,
                      int max_contacts, int *counts, orc_contactgeom *out);

/* one tick = dSpaceCollide + near callback + dWorldQuickStep + dJointGroupEmpty
 * (main.c:211-215 with QuickStep substituted for dWorldStep) */
void orc_world_tick(orc_world *w, real h);
int  orc_world_last_contact_count(orc_world *w);
/* geom pairs with finite AABBs (i.e. not involving a plane) that passed the broadphase in the last tick */
int  orc_world_last_body_pairs(orc_world *w);
/* diagnostic: sum over rows of |delta lambda| in the last SOR sweep of the last tick */
double orc_world_last_sor_residual(orc_world *w);

/* bulk helpers for large scenes (ctypes-friendly, arrays are n x k row-major) */
int  orc_world_body_count(orc_world *w);
void orc_world_add_boxes(orc_world *w, int n, const real *pos, const real *quat,
                         const real *lvel, const real *avel, const real *mass,
                         const real *idiag, const real *sides);
/* convex bodies share one hull: n body-frame points, 3 reals each (dCreateConvex's points array [ODE-recall]) */
void orc_world_set_hull(orc_world *w, int n, const real *points);
/* the hull's faces, nf x 4: unit outward normal and offset in the body frame (dCreateConvex's planes array [ODE-recall]);
 * needed by the box-convex collider's "box corner inside the hull" half */
void orc_world_set_hull_faces(orc_world *w, int nf, const real *planes);
int  orc_geom_create_convex(orc_world *w);
void orc_world_add_convex(orc_world *w, int n, const real *pos, const real *quat,
                          const real *lvel, const real *avel, const real *mass, const real *idiag);
void orc_world_add_spheres(orc_world *w, int n, const real *pos, const real *quat,
                           const real *lvel, const real *avel, const real *mass,
                           const real *idiag, const real *radius);
void orc_world_get_state(orc_world *w, real *pos, real *quat, real *lvel, real *avel);
/* run `steps` ticks, return wall seconds of the tick loop only */
double orc_world_run(orc_world *w, real h, int steps);

/* pose read-back: GetTransformMat (main.c:602-622), column-major 4x4 */
void orc_pack_transform(real out16[16], const real pos[3], const real R[12]);

/* PRNG restated from /root/reference/src/rand.c:7-13,21,29 ------------------ */
void     orc_ref_rand_seed(uint32_t s);
uint32_t orc_ref_rand_next(void);
int32_t  orc_ref_rand_int(int32_t min, int32_t max);
double   orc_ref_rand_double(double min, double max);

int orc_real_size(void);

#ifdef __cplusplus
}
#endif
#endif
