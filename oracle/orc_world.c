/*
 * oracle/orc_world.c -- world / body / geom bookkeeping of the CPU oracle.
 * TEST INFRASTRUCTURE (see orc.h).  Follows the reference's call sites:
 * world setup main.c:94-98, AddBody main.c:695-733, AddBodyMap main.c:735-761,
 * tick main.c:211-215, pose read-back main.c:221-237 + GetTransformMat 602-622.
 */
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <time.h>
#include "orc_internal.h"

static void set_identity12(real *m)
{
    memset(m, 0, 12 * sizeof(real));
    m[0] = m[5] = m[10] = 1;
}

orc_world *orc_world_create(void)
{
    orc_world *w = (orc_world *)calloc(1, sizeof(*w));
    /* [ODE-recall] dWorldCreate defaults: ERP 0.2, CFM 1e-5f/1e-10, QuickStep 20 iters, w 1.3 */
    w->erp = R(0.2);
    w->cfm = ORC_CFM_DEFAULT;
    w->iters = 20;
    w->sor_w = R(1.3);
    w->row_order = ORC_ORDER_FIXED;
    w->gyro_mode = ORC_GYRO_IMPLICIT;
    /* NearCallback surface, main.c:684-687 */
    w->surf_mode = ORC_CONTACT_BOUNCE;
    w->surf_mu = ORC_INF;
    w->surf_bounce = R(0.2);
    w->surf_bounce_vel = R(0.1);
    w->max_contacts = 8;      /* main.c:675 */
    return w;
}

void orc_world_destroy(orc_world *w)
{
    if (!w) return;
    free(w->bodies); free(w->geoms); free(w->joints); free(w->hull); free(w->hull_planes);
    free(w);
}

void orc_world_set_gravity(orc_world *w, real x, real y, real z)
{ w->gravity[0] = x; w->gravity[1] = y; w->gravity[2] = z; }
void orc_world_set_erp(orc_world *w, real erp) { w->erp = erp; }
void orc_world_set_cfm(orc_world *w, real cfm) { w->cfm = cfm; }
void orc_world_set_quickstep(orc_world *w, int iters, real sor_w) { w->iters = iters; w->sor_w = sor_w; }
void orc_world_set_row_order(orc_world *w, int mode) { w->row_order = mode; }
void orc_world_set_stepper(orc_world *w, int mode) { w->stepper = mode; }
int orc_world_last_lcp_rounds(orc_world *w) { return w->lcp_rounds; }
int orc_world_joint_count(orc_world *w) { return w->last_contacts; }   /* (the array outlives dJointGroupEmpty until the next tick) */
void orc_world_joint_info(orc_world *w, int k, int *b1, int *b2, real pos[3], real normal[3], real *depth, real *lambda_n)
{
    const orc_joint *j = &w->joints[k];
    *b1 = j->b1; *b2 = j->b2;
    for (int i = 0; i < 3; i++) { pos[i] = j->geom.pos[i]; normal[i] = j->reverse ? -j->geom.normal[i] : j->geom.normal[i]; }
    *depth = j->geom.depth; *lambda_n = j->lambda_n;
}
void orc_world_set_gyro_mode(orc_world *w, int mode) { w->gyro_mode = mode; }
void orc_world_set_surface(orc_world *w, int mode, real mu, real bounce, real bounce_vel)
{ w->surf_mode = mode; w->surf_mu = mu; w->surf_bounce = bounce; w->surf_bounce_vel = bounce_vel; }
void orc_world_set_max_contacts(orc_world *w, int n) { w->max_contacts = n; }
void orc_world_set_broadphase(orc_world *w, int mode) { w->bp_mode = mode; }

/* ---- bodies ------------------------------------------------------------ */
int orc_body_create(orc_world *w)
{
    if (w->nb == w->cap_b) {
        w->cap_b = w->cap_b ? 2 * w->cap_b : 64;
        w->bodies = (orc_body *)realloc(w->bodies, (size_t)w->cap_b * sizeof(orc_body));
    }
    orc_body *b = &w->bodies[w->nb];
    memset(b, 0, sizeof(*b));
    b->q[0] = 1;
    set_identity12(b->R);
    /* [ODE-recall] dBodyCreate: dMassSetParameters(m=1, cg=0, I=identity); the
       reference never calls dBodySetMass (main.c:695-733) */
    b->mass = 1; b->invMass = 1;
    set_identity12(b->I);
    set_identity12(b->invI);
    return w->nb++;
}

void orc_body_set_position(orc_world *w, int b, real x, real y, real z)
{ real *p = w->bodies[b].pos; p[0] = x; p[1] = y; p[2] = z; }

void orc_body_set_rotation(orc_world *w, int b, const real Rm[12])
{
    /* [ODE-recall] dBodySetRotation: copy R, q = dQfromR(R), normalise q.
       (ODE also re-orthogonalises R; callers here pass orthonormal R.) */
    orc_body *bd = &w->bodies[b];
    memcpy(bd->R, Rm, 12 * sizeof(real));
    bd->R[3] = bd->R[7] = bd->R[11] = 0;
    orc_R_to_q(bd->R, bd->q);
    orc_normalize4(bd->q);
}

void orc_body_set_quaternion(orc_world *w, int b, const real q[4])
{
    /* [ODE-recall] dBodySetQuaternion: copy, normalise, R = dQtoR(q) */
    orc_body *bd = &w->bodies[b];
    memcpy(bd->q, q, 4 * sizeof(real));
    orc_normalize4(bd->q);
    orc_q_to_R(bd->q, bd->R);
}

void orc_body_set_linear_vel(orc_world *w, int b, real x, real y, real z)
{ real *p = w->bodies[b].lvel; p[0] = x; p[1] = y; p[2] = z; }
void orc_body_set_angular_vel(orc_world *w, int b, real x, real y, real z)
{ real *p = w->bodies[b].avel; p[0] = x; p[1] = y; p[2] = z; }

void orc_body_set_mass(orc_world *w, int b, real mass, const real I[9])
{
    orc_body *bd = &w->bodies[b];
    bd->mass = mass;
    bd->invMass = R(1.0) / mass;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) bd->I[4 * i + j] = I[3 * i + j];
        bd->I[4 * i + 3] = 0;
    }
    if (bd->I[1] == 0 && bd->I[2] == 0 && bd->I[4] == 0 && bd->I[6] == 0 &&
        bd->I[8] == 0 && bd->I[9] == 0) {
        set_identity12(bd->invI);
        bd->invI[0] = R(1.0) / bd->I[0];
        bd->invI[5] = R(1.0) / bd->I[5];
        bd->invI[10] = R(1.0) / bd->I[10];
    } else if (!orc_invert3(bd->invI, bd->I)) {
        fprintf(stderr, "orc_body_set_mass: singular inertia\n");
        abort();
    }
}

void orc_body_add_force(orc_world *w, int b, real x, real y, real z)
{ real *p = w->bodies[b].facc; p[0] += x; p[1] += y; p[2] += z; }
void orc_body_add_torque(orc_world *w, int b, real x, real y, real z)
{ real *p = w->bodies[b].tacc; p[0] += x; p[1] += y; p[2] += z; }

const real *orc_body_get_position(orc_world *w, int b) { return w->bodies[b].pos; }
const real *orc_body_get_rotation(orc_world *w, int b) { return w->bodies[b].R; }
const real *orc_body_get_quaternion(orc_world *w, int b) { return w->bodies[b].q; }
const real *orc_body_get_linear_vel(orc_world *w, int b) { return w->bodies[b].lvel; }
const real *orc_body_get_angular_vel(orc_world *w, int b) { return w->bodies[b].avel; }

/* ---- geoms ------------------------------------------------------------- */
static int geom_new(orc_world *w, int type)
{
    if (w->ng == w->cap_g) {
        w->cap_g = w->cap_g ? 2 * w->cap_g : 64;
        w->geoms = (orc_geom *)realloc(w->geoms, (size_t)w->cap_g * sizeof(orc_geom));
    }
    orc_geom *g = &w->geoms[w->ng];
    memset(g, 0, sizeof(*g));
    g->type = type;
    g->body = -1;
    set_identity12(g->R);
    g->cat = ~0u; g->col = ~0u;   /* [ODE-recall] dxGeom ctor: both masks all ones */
    return w->ng++;
}

int orc_geom_create_box(orc_world *w, real lx, real ly, real lz)
{
    int g = geom_new(w, ORC_GEOM_BOX);
    w->geoms[g].side[0] = lx; w->geoms[g].side[1] = ly; w->geoms[g].side[2] = lz;
    return g;
}

int orc_geom_create_sphere(orc_world *w, real radius)
{
    int g = geom_new(w, ORC_GEOM_SPHERE);
    w->geoms[g].side[0] = radius;
    return g;
}

void orc_world_set_hull(orc_world *w, int n, const real *points)
{
    free(w->hull);
    w->hull = (real *)malloc((size_t)(n > 0 ? n : 1) * 3 * sizeof(real));
    memcpy(w->hull, points, (size_t)n * 3 * sizeof(real));
    w->hull_n = n;
}

void orc_world_set_hull_faces(orc_world *w, int nf, const real *planes)
{
    free(w->hull_planes);
    w->hull_planes = (real *)malloc((size_t)(nf > 0 ? nf : 1) * 4 * sizeof(real));
    memcpy(w->hull_planes, planes, (size_t)nf * 4 * sizeof(real));
    w->hull_nf = nf;
}

int orc_geom_create_convex(orc_world *w)
{
    return geom_new(w, ORC_GEOM_CONVEX);
}

int orc_geom_create_plane(orc_world *w, real a, real b, real c, real d)
{
    /* [ODE-recall] dCreatePlane normalises (a,b,c,d) by |(a,b,c)| */
    int g = geom_new(w, ORC_GEOM_PLANE);
    real l = a * a + b * b + c * c;
    if (l > 0) {
        l = R(1.0) / orc_sqrt(l);
        a *= l; b *= l; c *= l; d *= l;
    } else { a = 1; b = 0; c = 0; d = 0; }
    real *p = w->geoms[g].plane;
    p[0] = a; p[1] = b; p[2] = c; p[3] = d;
    return g;
}

void orc_geom_set_body(orc_world *w, int g, int b) { w->geoms[g].body = b; }
void orc_geom_set_position(orc_world *w, int g, real x, real y, real z)
{ real *p = w->geoms[g].pos; p[0] = x; p[1] = y; p[2] = z; }
void orc_geom_set_rotation(orc_world *w, int g, const real Rm[12])
{ memcpy(w->geoms[g].R, Rm, 12 * sizeof(real)); }
void orc_geom_set_category_bits(orc_world *w, int g, uint32_t bits) { w->geoms[g].cat = bits; }
void orc_geom_set_collide_bits(orc_world *w, int g, uint32_t bits) { w->geoms[g].col = bits; }

const real *orc_geom_pos(const orc_world *w, const orc_geom *g)
{ return g->body >= 0 ? w->bodies[g->body].pos : g->pos; }
const real *orc_geom_R(const orc_world *w, const orc_geom *g)
{ return g->body >= 0 ? w->bodies[g->body].R : g->R; }

/* ---- tick: main.c:211-215 ---------------------------------------------- */
void orc_world_tick(orc_world *w, real h)
{
    orc_collide_all(w);        /* dSpaceCollide(space, NULL, NearCallback)  main.c:212 */
    w->last_contacts = w->nj;
    orc_quickstep(w, h);       /* dWorldStep / dWorldQuickStep (orc_world_set_stepper)  main.c:213 */
    w->nj = 0;                 /* dJointGroupEmpty(contactGroup)            main.c:214 */
}

int orc_world_last_contact_count(orc_world *w) { return w->last_contacts; }
int orc_world_last_body_pairs(orc_world *w) { return w->last_body_pairs; }
double orc_world_last_sor_residual(orc_world *w) { return w->last_residual; }

/* ---- bulk helpers ------------------------------------------------------- */
int orc_world_body_count(orc_world *w) { return w->nb; }

static void add_bulk(orc_world *w, int type, int n, const real *pos, const real *quat,
                     const real *lvel, const real *avel, const real *mass,
                     const real *idiag, const real *dims)
{
    for (int i = 0; i < n; i++) {
        int b = orc_body_create(w);
        orc_body_set_position(w, b, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        if (quat) orc_body_set_quaternion(w, b, quat + 4 * i);
        if (lvel) orc_body_set_linear_vel(w, b, lvel[3 * i], lvel[3 * i + 1], lvel[3 * i + 2]);
        if (avel) orc_body_set_angular_vel(w, b, avel[3 * i], avel[3 * i + 1], avel[3 * i + 2]);
        if (mass) {
            real I[9] = { idiag[3 * i], 0, 0, 0, idiag[3 * i + 1], 0, 0, 0, idiag[3 * i + 2] };
            orc_body_set_mass(w, b, mass[i], I);
        }
        int g = (type == ORC_GEOM_BOX)
                    ? orc_geom_create_box(w, dims[3 * i], dims[3 * i + 1], dims[3 * i + 2])
                    : (type == ORC_GEOM_CONVEX) ? orc_geom_create_convex(w)
                    : orc_geom_create_sphere(w, dims[i]);
        /* AddBody: category CMASK_OBJ=2, collide CMASK_OBJ|CMASK_MAP=3  (main.c:181,724-725) */
        orc_geom_set_category_bits(w, g, 2u);
        orc_geom_set_collide_bits(w, g, 3u);
        orc_geom_set_body(w, g, b);
    }
}

void orc_world_add_boxes(orc_world *w, int n, const real *pos, const real *quat,
                         const real *lvel, const real *avel, const real *mass,
                         const real *idiag, const real *sides)
{ add_bulk(w, ORC_GEOM_BOX, n, pos, quat, lvel, avel, mass, idiag, sides); }

void orc_world_add_convex(orc_world *w, int n, const real *pos, const real *quat,
                          const real *lvel, const real *avel, const real *mass, const real *idiag)
{ add_bulk(w, ORC_GEOM_CONVEX, n, pos, quat, lvel, avel, mass, idiag, NULL); }

void orc_world_add_spheres(orc_world *w, int n, const real *pos, const real *quat,
                           const real *lvel, const real *avel, const real *mass,
                           const real *idiag, const real *radius)
{ add_bulk(w, ORC_GEOM_SPHERE, n, pos, quat, lvel, avel, mass, idiag, radius); }

void orc_world_get_state(orc_world *w, real *pos, real *quat, real *lvel, real *avel)
{
    for (int i = 0; i < w->nb; i++) {
        const orc_body *b = &w->bodies[i];
        if (pos)  { pos[3 * i] = b->pos[0]; pos[3 * i + 1] = b->pos[1]; pos[3 * i + 2] = b->pos[2]; }
        if (quat) { for (int k = 0; k < 4; k++) quat[4 * i + k] = b->q[k]; }
        if (lvel) { lvel[3 * i] = b->lvel[0]; lvel[3 * i + 1] = b->lvel[1]; lvel[3 * i + 2] = b->lvel[2]; }
        if (avel) { avel[3 * i] = b->avel[0]; avel[3 * i + 1] = b->avel[1]; avel[3 * i + 2] = b->avel[2]; }
    }
}

double orc_world_run(orc_world *w, real h, int steps)
{
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int s = 0; s < steps; s++) orc_world_tick(w, h);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}

/* GetTransformMat, main.c:602-622: column-major 4x4 from ODE's 3x4 row-major R */
void orc_pack_transform(real o[16], const real pos[3], const real Rm[12])
{
    o[0] = Rm[0];  o[1] = Rm[4];  o[2] = Rm[8];   o[3] = 0;
    o[4] = Rm[1];  o[5] = Rm[5];  o[6] = Rm[9];   o[7] = 0;
    o[8] = Rm[2];  o[9] = Rm[6];  o[10] = Rm[10]; o[11] = 0;
    o[12] = pos[0]; o[13] = pos[1]; o[14] = pos[2]; o[15] = 1;
}

int orc_real_size(void) { return (int)sizeof(real); }
