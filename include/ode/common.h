/*
 * include/ode/common.h -- basic types of the ODE-compatible C API served by
 * libode_mi355.so.  Clean-room declarations of the public ODE names the
 * reference compiles against: /root/reference/inc/body.h:4 includes this
 * header for dReal (body.h:28), dBodyID (body.h:21) and dGeomID (body.h:22);
 * /root/reference/src/main.c uses the rest through ode/ode.h (main.c:11).
 *
 * Precision follows ODE's convention: define dSINGLE or dDOUBLE before
 * including (default dDOUBLE) and link the matching library
 * (libode_mi355.so = dDOUBLE, libode_mi355_single.so = dSINGLE).
 */
#ifndef DMX_ODE_COMMON_H
#define DMX_ODE_COMMON_H

#include <math.h>

#if !defined(dSINGLE) && !defined(dDOUBLE)
#define dDOUBLE 1
#endif
#if defined(dSINGLE) && defined(dDOUBLE)
#error "define only one of dSINGLE / dDOUBLE"
#endif

#ifdef __cplusplus
extern "C" {
#endif

#ifdef dSINGLE
typedef float dReal;
#define dInfinity ((float)INFINITY)          /* main.c:687 */
#else
typedef double dReal;
#define dInfinity ((double)INFINITY)
#endif

/* layouts are ABI: 3-vectors are padded to 4, matrices are 3x4 row-major
   (indexed so at main.c:603-616), quaternions are (w,x,y,z) */
typedef dReal dVector3[4];
typedef dReal dVector4[4];
typedef dReal dMatrix3[4 * 3];
typedef dReal dQuaternion[4];

/* opaque object handles; a null handle means "none" (main.c:228, 260, 691) */
struct dxWorld;
struct dxSpace;
struct dxBody;
struct dxGeom;
struct dxJoint;
struct dxJointGroup;
typedef struct dxWorld *dWorldID;            /* main.c:35  */
typedef struct dxSpace *dSpaceID;            /* main.c:36  */
typedef struct dxBody *dBodyID;              /* body.h:21  */
typedef struct dxGeom *dGeomID;              /* body.h:22  */
typedef struct dxJoint *dJointID;            /* main.c:690 */
typedef struct dxJointGroup *dJointGroupID;  /* main.c:37  */

#ifdef __cplusplus
}
#endif
#endif
