/* oracle/orc_internal.h -- private structs of the CPU oracle (TEST INFRASTRUCTURE, see orc.h) */
#ifndef ORC_INTERNAL_H
#define ORC_INTERNAL_H

#include "orc.h"
#include "orc_math.h"

enum { ORC_BODY_KINEMATIC = 1, ORC_BODY_NOGRAVITY = 2 };

typedef struct {
    real pos[4];
    real q[4];
    real R[12];
    real lvel[4], avel[4];
    real facc[4], tacc[4];
    real mass, invMass;
    real I[12], invI[12];     /* body frame */
    int flags;
    int tag;
} orc_body;

typedef struct {
    int type;
    int body;                 /* -1 = static geom (dGeomGetBody == 0, main.c:691) */
    real side[3];             /* box: full side lengths; sphere: side[0] = radius */
    real plane[4];
    real pos[4];              /* used when body < 0 */
    real R[12];
    uint32_t cat, col;
} orc_geom;

typedef struct {
    orc_contactgeom geom;
    int mode;
    real mu, bounce, bounce_vel;
    int b1, b2;               /* b1 >= 0 always; b2 may be -1 */
    int reverse;              /* dJOINT_REVERSE: bodies were swapped at attach */
    real lambda_n;            /* diagnostics: the normal row's multiplier after the last step (a force: lambda) */
    int tag;
} orc_joint;

struct orc_world {
    real gravity[4];
    real erp, cfm, sor_w;
    int iters;
    int row_order, gyro_mode;
    int stepper;              /* ORC_STEPPER_QUICK (SOR, dWorldQuickStep) or ORC_STEPPER_EXACT (dWorldStep: the LCP solved exactly) */
    int lcp_rounds;           /* diagnostics: pivoting rounds of the last exact solve (largest island) */
    int surf_mode; real surf_mu, surf_bounce, surf_bounce_vel;
    int max_contacts;
    int bp_mode;              /* broadphase: 0 auto, 1 sweep along x, 2 uniform (x,z) grid */

    orc_body *bodies; int nb, cap_b;
    orc_geom *geoms;  int ng, cap_g;
    orc_joint *joints; int nj, cap_j;

    real *hull; int hull_n;   /* body-frame points of the hull every ORC_GEOM_CONVEX geom uses */
    real *hull_planes; int hull_nf;   /* its faces: unit outward normal + offset (n.x <= d inside), body frame */

    int last_contacts;
    int last_body_pairs;      /* finite-AABB pairs that reached the near callback in the last tick */
    double last_residual;
};

/* geometry pose, whether body-attached or static */
const real *orc_geom_pos(const orc_world *w, const orc_geom *g);
const real *orc_geom_R(const orc_world *w, const orc_geom *g);

void orc_collide_all(orc_world *w);           /* dSpaceCollide + NearCallback */
void orc_quickstep(orc_world *w, real h);     /* dWorldQuickStep */
uint32_t orc_ode_rand(void);
int orc_ode_rand_int(int n);

#endif
