/*
 * include/ode/ode.h -- the ODE C API subset that /root/reference/src/main.c
 * links against (33 symbols), plus the few neighbours the
 * BASELINE.json scenes need (dWorldQuickStep, dCreatePlane, dBodySetMass,
 * dMassSetBox ...).  Served by libode_mi355.so: object bookkeeping, the
 * broadphase callback loop and dCollide run on the host (the API is a
 * synchronous per-pair host callback, main.c:212, 674-693); the step itself --
 * contact rows, SOR, integration -- runs on the MI355X.  Each declaration
 * cites the reference call site it serves.
 *
 * dWorldStep (main.c:213) solves every island's boxed LCP to the end -- the
 * solution ODE's Dantzig solver reaches, by block principal pivoting on the
 * device -- and dWorldQuickStep runs QuickStep's 20 SOR sweeps: the two names
 * behave as ODE's two steppers do.
 * Errors: no call site checks a return value (main.c:94-98, 212-214); failures
 * print to stderr and return 0 / NULL, HIP failures abort like ODE's dError.
 */
#ifndef DMX_ODE_H
#define DMX_ODE_H

#include <stddef.h>
#include "common.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- contact structures (main.c:676-687) ---------------------------------
 * Layout and flag values are those of the ODE 0.13 - 0.16 line's ode/contact.h [ODE-recall; SURVEY.md cites 0.16.x]: rolling
 * friction (rho, rho2, rhoN; dContactRolling, dContactApprox1_N) sits between mu2 and bounce, so an object compiled against
 * stock 0.16 headers writes `bounce` where this library reads it (tests/test_ode_compat.py checks every offset).  ODE <= 0.12
 * had no rho fields: objects built against those headers must be recompiled against these.  The rolling-friction fields are
 * accepted and ignored (the reference never sets them, main.c:684-687). */
enum {
    dContactMu2      = 0x001,
    dContactAxisDep  = 0x001,
    dContactFDir1    = 0x002,
    dContactBounce   = 0x004,   /* main.c:684 */
    dContactSoftERP  = 0x008,
    dContactSoftCFM  = 0x010,
    dContactMotion1  = 0x020,
    dContactMotion2  = 0x040,
    dContactMotionN  = 0x080,
    dContactSlip1    = 0x100,
    dContactSlip2    = 0x200,
    dContactRolling  = 0x400,
    dContactApprox0  = 0x0000,
    dContactApprox1_1 = 0x1000,
    dContactApprox1_2 = 0x2000,
    dContactApprox1_N = 0x4000,
    dContactApprox1  = 0x7000
};

typedef struct dSurfaceParameters {
    int mode;                 /* main.c:684 */
    dReal mu;                 /* main.c:687 */
    dReal mu2;
    dReal rho, rho2, rhoN;    /* rolling / spinning friction (0.13+): not used by the reference, ignored here */
    dReal bounce;             /* main.c:685 */
    dReal bounce_vel;         /* main.c:686 */
    dReal soft_erp;
    dReal soft_cfm;
    dReal motion1, motion2, motionN;
    dReal slip1, slip2;
} dSurfaceParameters;

typedef struct dContactGeom {
    dVector3 pos;
    dVector3 normal;          /* points into body 1 */
    dReal depth;
    dGeomID g1, g2;
    int side1, side2;
} dContactGeom;

typedef struct dContact {
    dSurfaceParameters surface;
    dContactGeom geom;        /* &contacts[0].geom with stride sizeof(dContact): main.c:678 */
    dVector3 fdir1;
} dContact;

typedef void dNearCallback(void *data, dGeomID o1, dGeomID o2);   /* main.c:48, 674 */

typedef struct dMass {
    dReal mass;
    dVector3 c;
    dMatrix3 I;
} dMass;

enum { dSphereClass = 0, dBoxClass = 1, dPlaneClass = 4 };

/* ---- library lifecycle ---------------------------------------------------- */
void dInitODE(void);                                             /* main.c:94  */
int  dInitODE2(unsigned int flags);
void dCloseODE(void);                                            /* main.c:267 */

/* ---- world ---------------------------------------------------------------- */
dWorldID dWorldCreate(void);                                     /* main.c:95  */
void dWorldDestroy(dWorldID);                                    /* main.c:266 */
void dWorldSetGravity(dWorldID, dReal x, dReal y, dReal z);      /* main.c:96  */
void dWorldGetGravity(dWorldID, dVector3 gravity);
void dWorldSetERP(dWorldID, dReal erp);
dReal dWorldGetERP(dWorldID);
void dWorldSetCFM(dWorldID, dReal cfm);
dReal dWorldGetCFM(dWorldID);
void dWorldSetQuickStepNumIterations(dWorldID, int num);
int  dWorldGetQuickStepNumIterations(dWorldID);
void dWorldSetQuickStepW(dWorldID, dReal over_relaxation);
dReal dWorldGetQuickStepW(dWorldID);
int  dWorldStep(dWorldID, dReal stepsize);                       /* main.c:213 */
int  dWorldQuickStep(dWorldID, dReal stepsize);

/* ---- bodies --------------------------------------------------------------- */
dBodyID dBodyCreate(dWorldID);                                   /* main.c:703 */
void dBodyDestroy(dBodyID);                                      /* main.c:261 */
void dBodySetPosition(dBodyID, dReal x, dReal y, dReal z);       /* main.c:708 */
void dBodySetRotation(dBodyID, const dMatrix3 R);                /* main.c:709 */
void dBodySetQuaternion(dBodyID, const dQuaternion q);
void dBodySetLinearVel(dBodyID, dReal x, dReal y, dReal z);
void dBodySetAngularVel(dBodyID, dReal x, dReal y, dReal z);
const dReal *dBodyGetPosition(dBodyID);                          /* main.c:229 */
const dReal *dBodyGetRotation(dBodyID);                          /* main.c:230 */
const dReal *dBodyGetQuaternion(dBodyID);
const dReal *dBodyGetLinearVel(dBodyID);
const dReal *dBodyGetAngularVel(dBodyID);
void dBodySetKinematic(dBodyID);                                 /* main.c:712 */
void dBodySetDynamic(dBodyID);
int  dBodyIsKinematic(dBodyID);
void dBodyAddForce(dBodyID, dReal fx, dReal fy, dReal fz);       /* main.c:532 (commented out there) */
void dBodyAddTorque(dBodyID, dReal fx, dReal fy, dReal fz);
void dBodySetMass(dBodyID, const dMass *mass);
void dBodyGetMass(dBodyID, dMass *mass);
void dBodySetGyroscopicMode(dBodyID, int enabled);
int  dBodyGetGyroscopicMode(dBodyID);
dWorldID dBodyGetWorld(dBodyID);

void dMassSetZero(dMass *);
void dMassSetParameters(dMass *, dReal themass, dReal cgx, dReal cgy, dReal cgz,
                        dReal I11, dReal I22, dReal I33, dReal I12, dReal I13, dReal I23);
void dMassSetBox(dMass *, dReal density, dReal lx, dReal ly, dReal lz);
void dMassSetBoxTotal(dMass *, dReal total_mass, dReal lx, dReal ly, dReal lz);
void dMassSetSphere(dMass *, dReal density, dReal radius);
void dMassSetSphereTotal(dMass *, dReal total_mass, dReal radius);

/* ---- spaces and geoms ----------------------------------------------------- */
dSpaceID dHashSpaceCreate(dSpaceID space);                       /* main.c:97  */
dSpaceID dSimpleSpaceCreate(dSpaceID space);
void dSpaceDestroy(dSpaceID);
void dSpaceCollide(dSpaceID space, void *data, dNearCallback *callback);   /* main.c:212 */
int  dSpaceGetNumGeoms(dSpaceID);

dGeomID dCreateBox(dSpaceID space, dReal lx, dReal ly, dReal lz);          /* main.c:720, 743 */
dGeomID dCreateSphere(dSpaceID space, dReal radius);                       /* main.c:717 */
dGeomID dCreatePlane(dSpaceID space, dReal a, dReal b, dReal c, dReal d);
void dGeomDestroy(dGeomID);                                      /* main.c:263 */
void dGeomSetBody(dGeomID, dBodyID);                             /* main.c:726 */
dBodyID dGeomGetBody(dGeomID);                                   /* main.c:691 */
void dGeomSetPosition(dGeomID, dReal x, dReal y, dReal z);       /* main.c:748 */
void dGeomSetRotation(dGeomID, const dMatrix3 R);                /* main.c:749 */
const dReal *dGeomGetPosition(dGeomID);                          /* main.c:232 */
const dReal *dGeomGetRotation(dGeomID);                          /* main.c:233 */
void dGeomSetCategoryBits(dGeomID, unsigned long bits);          /* main.c:724, 751, 752 */
void dGeomSetCollideBits(dGeomID, unsigned long bits);           /* main.c:725 */
unsigned long dGeomGetCategoryBits(dGeomID);
unsigned long dGeomGetCollideBits(dGeomID);
int  dGeomGetClass(dGeomID);
void dGeomBoxGetLengths(dGeomID box, dVector3 result);
dReal dGeomSphereGetRadius(dGeomID sphere);
void dGeomPlaneGetParams(dGeomID plane, dVector4 result);

/* flags: low 16 bits = max contacts; skip = byte stride between dContactGeoms (main.c:678) */
int dCollide(dGeomID o1, dGeomID o2, int flags, dContactGeom *contact, int skip);

/* ---- contact joints ------------------------------------------------------- */
dJointGroupID dJointGroupCreate(int max_size);                   /* main.c:98  */
void dJointGroupEmpty(dJointGroupID);                            /* main.c:214 */
void dJointGroupDestroy(dJointGroupID);                          /* main.c:265 */
dJointID dJointCreateContact(dWorldID, dJointGroupID, const dContact *);   /* main.c:690 */
void dJointAttach(dJointID, dBodyID body1, dBodyID body2);       /* main.c:691 */

/* ---- rotation helpers used when filling dBodySetRotation's argument ------- */
void dRSetIdentity(dMatrix3 R);
void dRFromAxisAndAngle(dMatrix3 R, dReal ax, dReal ay, dReal az, dReal angle);
void dQtoR(const dQuaternion q, dMatrix3 R);
void dRtoQ(const dMatrix3 R, dQuaternion q);

/* ---- MI355X extension: bulk pose snapshot for the 60 Hz broadcast loop ------
 * (main.c:221-237 + GetTransformMat main.c:602-622): fills out[i*16 .. i*16+15]
 * (column-major 4x4) for bodies[i], i < n, in one device pass + one copy. */
int dmxWorldSnapshotTransforms(dWorldID, const dBodyID *bodies, int n, dReal *out);
/* The same over the reference's own arrays, in place of the loop main.c:221-237:
 *   dmxWorldSnapshotBodyStates(world, &bodies[0].body, sizeof(Body), MAX_BODIES,
 *                              bodyStates[0].transform, sizeof(BodyState));
 * handles that are 0 (empty slots and static geoms, main.c:228) are skipped, their states left as they are.
 * Returns the number of transforms written, -1 on error. */
int dmxWorldSnapshotBodyStates(dWorldID, const void *first_body, size_t body_stride, int n,
                               void *first_transform, size_t state_stride);

#ifdef __cplusplus
}
#endif
#endif
