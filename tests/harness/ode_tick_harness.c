/*
 * ode_tick_harness.c -- a headless stand-in for the physics half of the reference's StartServer()
 * (/root/reference/src/main.c:59-270), written against include/ode/ode.h only.  It makes the same
 * ODE calls in the same order as the reference does -- world setup (main.c:94-98), static map boxes
 * (AddBodyMap, main.c:735-761, including its double dGeomSetCategoryBits), dynamic bodies (AddBody,
 * main.c:695-733), the tick loop (main.c:211-215) with the reference's contact policy (NearCallback,
 * main.c:674-693) and the pose read-back (main.c:221-237 + GetTransformMat, main.c:602-622) -- with
 * raylib / ENet / the window loop left out.  The scene comes from stdin so tests can feed the CPU
 * oracle the same numbers.
 *
 * stdin:  dt steps use_plane
 *         n_static  then per static box : sx sy sz  px py pz  R[12]
 *         n_body    then per body       : type(1 sphere, 2 box) sx sy sz  px py pz  R[12]
 * stdout: per body 16 numbers (column-major 4x4 transform), %.17g
 */
#define _POSIX_C_SOURCE 199309L
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <ode/ode.h>

#define MAX_CONTACTS 8              /* main.c:675 */
enum { CMASK_MAP = 1, CMASK_OBJ = 2, CMASK_ALL = ~0 };   /* inc/body.h:8-12 */

static dWorldID world;
static dSpaceID space;
static dJointGroupID contactGroup;

static void near_callback(void *data, dGeomID o1, dGeomID o2)
{
    dContact contacts[MAX_CONTACTS];
    int i, nc;
    (void)data;
    nc = dCollide(o1, o2, MAX_CONTACTS, &contacts[0].geom, sizeof(dContact));
    if (nc <= 0) return;
    for (i = 0; i < nc; i++) {
        dJointID c;
        contacts[i].surface.mode = dContactBounce;
        contacts[i].surface.bounce = 0.2;
        contacts[i].surface.bounce_vel = 0.1;
        contacts[i].surface.mu = dInfinity;
        c = dJointCreateContact(world, contactGroup, &contacts[i]);
        dJointAttach(c, dGeomGetBody(o1), dGeomGetBody(o2));
    }
}

static void pack_transform(dReal res[16], const dReal *pos, const dReal *rot)
{
    res[0] = rot[0]; res[1] = rot[4]; res[2] = rot[8];  res[3] = 0;
    res[4] = rot[1]; res[5] = rot[5]; res[6] = rot[9];  res[7] = 0;
    res[8] = rot[2]; res[9] = rot[6]; res[10] = rot[10]; res[11] = 0;
    res[12] = pos[0]; res[13] = pos[1]; res[14] = pos[2]; res[15] = 1;
}

static double now_ms(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static double rd(void)
{
    double v;
    if (scanf("%lf", &v) != 1) { fprintf(stderr, "harness: bad scene input\n"); exit(2); }
    return v;
}

int main(void)
{
    int steps, use_plane, n_static, n_body, i, k, s, up_front, every, readback, created, quick, exact_after;
    double dt, sink = 0, t_collide = 0, t_step = 0, t0, t1, t2;
    int time_from, timed = 0;
    struct spawn { int type; dReal size[3], pos[3]; dMatrix3 rm; } *spawn;
    dBodyID *bodies;
    dGeomID *geoms;

    dt = rd(); steps = (int)rd(); use_plane = (int)rd();

    dInitODE();
    world = dWorldCreate();
    dWorldSetGravity(world, 0.0, -9.8, 0.0);
    space = dHashSpaceCreate(0);
    contactGroup = dJointGroupCreate(0);
    if (use_plane) dCreatePlane(space, 0, 1, 0, 0);

    n_static = (int)rd();
    for (i = 0; i < n_static; i++) {
        dReal sx = (dReal)rd(), sy = (dReal)rd(), sz = (dReal)rd();
        dReal px = (dReal)rd(), py = (dReal)rd(), pz = (dReal)rd();
        dMatrix3 rm;
        dGeomID g;
        for (k = 0; k < 12; k++) rm[k] = (dReal)rd();
        g = dCreateBox(space, sx, sy, sz);
        dGeomSetPosition(g, px, py, pz);
        dGeomSetRotation(g, rm);
        dGeomSetCategoryBits(g, CMASK_MAP);
        dGeomSetCategoryBits(g, CMASK_ALL & ~CMASK_MAP);      /* sic: main.c:751-752 */
    }

    n_body = (int)rd();
    bodies = (dBodyID *)calloc((size_t)n_body + 1, sizeof(dBodyID));
    geoms = (dGeomID *)calloc((size_t)n_body + 1, sizeof(dGeomID));
    spawn = (struct spawn *)calloc((size_t)n_body + 1, sizeof(struct spawn));
    for (i = 0; i < n_body; i++) {
        spawn[i].type = (int)rd();
        for (k = 0; k < 3; k++) spawn[i].size[k] = (dReal)rd();
        for (k = 0; k < 3; k++) spawn[i].pos[k] = (dReal)rd();
        for (k = 0; k < 12; k++) spawn[i].rm[k] = (dReal)rd();
    }
    /* HARNESS_SPAWN="m k": the first m bodies exist from the start, then one more every k ticks (the reference spawns
     * between ticks on a key press, main.c:502-521); HARNESS_READBACK=r: poses of the live bodies are read every r
     * ticks (the 60 Hz broadcast loop, main.c:221-237).  Unset: everything up front, poses read at the end only. */
    up_front = n_body; every = 0; readback = 0;
    if (getenv("HARNESS_SPAWN") && sscanf(getenv("HARNESS_SPAWN"), "%d %d", &up_front, &every) != 2) { up_front = n_body; every = 0; }
    if (getenv("HARNESS_READBACK")) readback = atoi(getenv("HARNESS_READBACK"));
    quick = getenv("HARNESS_STEPPER") && getenv("HARNESS_STEPPER")[0] == 'q';
    /* HARNESS_EXACT_AFTER=n: dWorldQuickStep for the first n ticks (a pile settles cheaply), dWorldStep from then on */
    exact_after = getenv("HARNESS_EXACT_AFTER") ? atoi(getenv("HARNESS_EXACT_AFTER")) : 0;
    /* HARNESS_TIME_FROM=n: wall time of dSpaceCollide and of the step call, summed over ticks n.., reported on stderr */
    time_from = getenv("HARNESS_TIME_FROM") ? atoi(getenv("HARNESS_TIME_FROM")) : -1;
    created = 0;
    for (s = 0; s <= steps; s++) {
        while (created < n_body && (created < up_front || s == steps || (every > 0 && s >= (created - up_front + 1) * every))) {
            i = created++;
            bodies[i] = dBodyCreate(world);
            dBodySetPosition(bodies[i], spawn[i].pos[0], spawn[i].pos[1], spawn[i].pos[2]);
            dBodySetRotation(bodies[i], spawn[i].rm);
            geoms[i] = (spawn[i].type == 1) ? dCreateSphere(space, spawn[i].size[0])
                                            : dCreateBox(space, spawn[i].size[0], spawn[i].size[1], spawn[i].size[2]);
            dGeomSetCategoryBits(geoms[i], CMASK_OBJ);
            dGeomSetCollideBits(geoms[i], CMASK_OBJ | CMASK_MAP);
            dGeomSetBody(geoms[i], bodies[i]);
        }
        if (s == steps) break;
        t0 = now_ms();
        dSpaceCollide(space, NULL, near_callback);
        t1 = now_ms();
        if (quick || s < exact_after) dWorldQuickStep(world, (dReal)dt);          /* HARNESS_STEPPER=quick: BASELINE's configs name dWorldQuickStep */
        else dWorldStep(world, (dReal)dt);                     /* the reference's call, main.c:213 */
        dJointGroupEmpty(contactGroup);
        if (time_from >= 0 && s >= time_from) {
            /* the step is asynchronous until somebody reads a pose: read one, as the reference's broadcast loop does */
            if (created > 0) sink += (double)dBodyGetPosition(bodies[0])[1];
            t2 = now_ms();
            t_collide += t1 - t0; t_step += t2 - t1; timed++;
        }
        if (readback > 0 && (s + 1) % readback == 0)
            for (i = 0; i < created; i++) {
                dReal t[16];
                pack_transform(t, dBodyGetPosition(bodies[i]), dBodyGetRotation(bodies[i]));
                sink += (double)t[13];
            }
    }

    if (timed > 0)
        fprintf(stderr, "harness: timed_ticks=%d ms_per_tick=%.4f collide_ms=%.4f step_ms=%.4f\n", timed, (t_collide + t_step) / timed,
                t_collide / timed, t_step / timed);
    for (i = 0; i < n_body; i++) {
        dReal t[16];
        pack_transform(t, dBodyGetPosition(bodies[i]), dBodyGetRotation(bodies[i]));
        for (k = 0; k < 16; k++) printf("%.17g%c", (double)t[k], k == 15 ? '\n' : ' ');
    }

    for (i = 0; i < n_body; i++) {
        dBodyDestroy(bodies[i]);
        dGeomDestroy(geoms[i]);
    }
    dJointGroupDestroy(contactGroup);
    dWorldDestroy(world);
    dCloseODE();
    free(bodies); free(geoms); free(spawn);
    return sink == 12345.678 ? 1 : 0;
}
