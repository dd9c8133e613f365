/*
 * include/dmx_batch.h -- batch extension of the ODE-compatible C ABI.
 *
 * The reference drives ODE one geom pair at a time through a host callback
 * (dSpaceCollide -> NearCallback -> dCollide -> dJointCreateContact,
 * /root/reference/src/main.c:212, 674-693) and one body at a time through
 * dBodyCreate/dBodySet* (main.c:695-733) and dBodyGet* (main.c:221-237).
 * That shape cannot feed 10^6 bodies to a GPU, so the scenes of
 * BASELINE.json are driven through this batch form of the SAME calls: each
 * entry point below names the reference call (file:line) whose per-object
 * loop it replaces.  Plain C, plain pointers and sizes, no framework types.
 *
 * All state lives in HBM in one tiled slab (tiles of 64 bodies, see dmxBatchDevicePtr); `real` is float (DMX_F32,
 * ODE dSINGLE) or double (DMX_F64, ODE dDOUBLE) per batch.  Host arrays are
 * array-of-structs, row-major n x k, in the batch's precision.
 *
 * Every function returns 0 on success and a negative DMX_E* code on failure
 * (and prints the HIP error to stderr); nothing falls back to the CPU.
 */
#ifndef DMX_BATCH_H
#define DMX_BATCH_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmxBatch *dmxBatchID;

enum { DMX_F32 = 0, DMX_F64 = 1 };

enum {
    DMX_OK = 0,
    DMX_ENODEVICE = -1,   /* no usable HIP device */
    DMX_EHIP = -2,        /* a HIP call failed */
    DMX_EINVAL = -3,      /* bad argument */
    DMX_ENOMEM = -4,
    DMX_ECAPACITY = -5,   /* a fixed-size device table overflowed (broadphase bucket / pair buffer) */
    DMX_ECROSS = -6       /* two bodies owned by different ranks touch: the island has to be migrated to one owner */
};

/* per-body fields (k = components per body) */
enum {
    DMX_POS = 0,      /* k=3  dBodySetPosition / dBodyGetPosition   main.c:708, 229 */
    DMX_QUAT = 1,     /* k=4  (w,x,y,z); dBodySetRotation's R as q  main.c:709, 230 */
    DMX_LVEL = 2,     /* k=3  linear velocity  (0 at creation)      main.c:703      */
    DMX_AVEL = 3,     /* k=3  angular velocity (0 at creation)      main.c:703      */
    DMX_MASS = 4,     /* k=1  dBodySetMass mass  (default 1)             */
    DMX_INERTIA = 5,  /* k=3  body-frame principal inertia (default 1,1,1)          */
    DMX_SIDES = 6,    /* k=3  dCreateBox side lengths / (r,-,-) for a sphere  main.c:717,720 */
    DMX_FORCE = 7,    /* k=3  dBodyAddForce accumulator, cleared each step    main.c:532 */
    DMX_TORQUE = 8,   /* k=3  dBodyAddTorque accumulator                                  */
    DMX_QUAT_RAW = 9, /* k=4  like DMX_QUAT but stored as given (DMX_QUAT normalises on upload,
                              as dBodySetQuaternion does); for state that came from the device */
    DMX_STATE = 10,   /* k=13 pos3 quat4 lvel3 avel3 in one piece (the row layout of GatherBodies and of the boundary
                              pack): the whole read-back of main.c:221-237 in one transfer; the quaternion is stored
                              as given, like DMX_QUAT_RAW */
    DMX_NFIELDS = 11
};

/* geometry class per body (uint8 array) */
enum { DMX_GEOM_NONE = 0, DMX_GEOM_SPHERE = 1, DMX_GEOM_BOX = 2,     /* BodyType, inc/body.h:14-18 */
       DMX_GEOM_CONVEX = 3 };  /* a convex hull (BASELINE configs[4]); the reference itself creates none */

enum { DMX_GYRO_OFF = 0, DMX_GYRO_EXPLICIT = 1, DMX_GYRO_IMPLICIT = 2 };

/* dContactBounce etc: surface mode bits accepted by dmxBatchSetSurface */
enum { DMX_CONTACT_BOUNCE = 0x004 };

/* ---- lifecycle: dInitODE/dWorldCreate/dHashSpaceCreate/dJointGroupCreate (main.c:94-98) */
int dmxDeviceCount(void);                 /* >=1 or DMX_ENODEVICE; never touches a CPU path */
/* (Creation also asks the HIP runtime for the attributes of every kernel a tick of this precision may launch: code objects and
 * per-kernel set-up load here, ~15 ms once per process and precision, rather than as a hitch at the first tick that needs the
 * exact pipeline -- DMX_PRELOAD=0 in the environment leaves that lazy.) */
int dmxBatchCreate(dmxBatchID *out, int64_t n_bodies, int precision, int device);
int dmxBatchDestroy(dmxBatchID b);        /* dWorldDestroy / dCloseODE  main.c:258-268 */
int64_t dmxBatchBodyCount(dmxBatchID b);
int dmxBatchPrecision(dmxBatchID b);

/* ---- world parameters */
int dmxBatchSetGravity(dmxBatchID b, double gx, double gy, double gz);      /* dWorldSetGravity main.c:96 */
int dmxBatchSetERP(dmxBatchID b, double erp);                                /* dWorldSetERP  (default 0.2) */
int dmxBatchSetCFM(dmxBatchID b, double cfm);                                /* dWorldSetCFM  (default 1e-5f / 1e-10) */
int dmxBatchSetQuickStep(dmxBatchID b, int iterations, double sor_w);        /* dWorldSetQuickStepNumIterations / W */
int dmxBatchSetGyroMode(dmxBatchID b, int mode);                             /* dBodySetGyroscopicMode + form */
/* NearCallback's per-contact surface, applied to every generated contact (main.c:684-687) */
int dmxBatchSetSurface(dmxBatchID b, int mode, double mu, double bounce, double bounce_vel);
int dmxBatchSetMaxContacts(dmxBatchID b, int max_contacts);                  /* dCollide flags (main.c:675,678) */
/* dCreatePlane(space,a,b,c,d): the one static half-space bodies collide with; enable=0 removes it */
int dmxBatchSetPlane(dmxBatchID b, double a, double bb, double c, double d, int enable);

/* AddBodyMap (main.c:735-761): n static, body-less box geoms -- the reference's floor and walls (main.c:115-121) --
 * created before every body, as the reference creates them: side lengths (n x 3), positions (n x 3) and rotations
 * (n x 12: the 3x4 row-major matrix dGeomSetRotation takes, main.c:749), doubles, rounded to the batch's precision.
 * Every body collides with every static box (the category / collide bits the reference ends up with, main.c:724-725,
 * 751-752); static boxes do not link dynamics islands and static-static pairs are ignored.  Contacts are dCollide(static
 * geom, body geom) in creation order: ground plane, static boxes in order, then body pairs.  n = 0 removes them;
 * at most DMX_MAX_STATIC_BOXES.
 * A body touching static geometry only (the ground plane, static boxes) is a dynamics island of its own; such bodies are
 * stepped by a fused path -- narrowphase against the plane and the static boxes, then rows + 20 sweeps + integration in
 * registers, one lane per body -- inside the same collision-proof chunks as free bodies (dmxBatchStep).  The path holds
 * 8 contacts per body (the reference's MAX_CONTACTS for ONE geom pair, main.c:675): a body with more (wedged between the
 * floor and two walls, say) sends its chunk through the exact path (pair search, narrowphase, island solve), which has no
 * such limit; with the collision proof switched off (dmxBatchSetBodyCollisions(b, 0)) nobody can, and such a body is
 * stepped with its first 8 contacts.  Same results as the exact path, bit for bit. */
#define DMX_MAX_STATIC_BOXES 64
int dmxBatchSetStaticBoxes(dmxBatchID b, int32_t n, const double *sides, const double *pos, const double *rot3x4);
/* dGeomSetCategoryBits / dGeomSetCollideBits (main.c:724-725, 751-752) at the granularity of geometry classes: whether bodies of
 * class_a and class_b (DMX_GEOM_SPHERE / BOX / CONVEX; symmetric) collide with one another.  Default: every class with every
 * class, as the reference's bits have it (CMASK_OBJ bodies collide with CMASK_OBJ | CMASK_MAP).  Pairs switched off are not kept
 * apart by the safe zones, not enumerated by the pair search and never reach a collider -- what ODE does with geoms whose bits do
 * not match.  (Static boxes and the ground plane collide with every body.) */
int dmxBatchSetClassPairs(dmxBatchID b, int class_a, int class_b, int enable);
/* DMX_STATIC_FUSED (default; DMX_STATIC_FAST=0 in the environment picks the other): as described above.  DMX_STATIC_EXACT:
 * every body whose bounding sphere reaches a static box goes through the exact path every tick (how round 2 did it; kept
 * for A/B runs and for the tests that hold the two against each other). */
enum { DMX_STATIC_EXACT = 0, DMX_STATIC_FUSED = 1 };
int dmxBatchSetStaticPath(dmxBatchID b, int mode);

/* ---- body data: replaces the AddBody loop (main.c:695-733) and the read-back loop (main.c:221-237) */
int dmxBatchUpload(dmxBatchID b, int field, const void *host_aos, int64_t first, int64_t count);
int dmxBatchDownload(dmxBatchID b, int field, void *host_aos, int64_t first, int64_t count);
int dmxBatchUploadGeomType(dmxBatchID b, const uint8_t *types, int64_t first, int64_t count);
/* dCreateConvex's point set for every DMX_GEOM_CONVEX body of the batch: n_points body-frame points (3 doubles each;
 * origin = centre of mass, e.g. from dmxHullBuild in dmx_hull.h).  *radius_out (may be NULL) = the hull's bounding
 * radius: upload it as sides[0] of every convex body (the broadphase reads it there, as it does a sphere's radius).
 * Contacts: convex against the ground plane as ODE's dCollideConvexPlane makes them (the hull's points in array order,
 * the first max_contacts <= 8 on or below the plane); against boxes, spheres and other convex bodies: this library's own
 * colliders, which need the hull's faces (dmxBatchSetConvexHullFaces below). */
int dmxBatchSetConvexHull(dmxBatchID b, int32_t n_points, const double *points_xyz, double *radius_out);
/* dCreateConvex's plane set for the same hull: n_faces x 4 doubles (unit outward normal, offset; body frame), e.g. from
 * dmxHullPlanes.  Contacts of a convex body with BOXES -- static boxes (dmxBatchSetStaticBoxes: the floor of BASELINE
 * configs[4]) and box bodies -- are this library's own collider (ODE's dCollideConvexBox is an empty stub): hull vertices
 * inside the box in array order, each along the box face it is nearest to, then -- with the faces given here -- box
 * corners inside the hull, each along the hull face it is nearest to; the first max_contacts <= 8 are kept (edge-edge
 * penetrations are not detected).  Convex against CONVEX (both bodies carry the batch's one hull): the same two primitives --
 * the second body's vertices inside the first, in array order, each along the first's face it is nearest to, then the first
 * body's vertices inside the second; the first max_contacts are kept.  SPHERE against convex: the hull's face planes' largest
 * signed distance to the sphere's centre (first face on ties); within the radius, one contact along that face -- exact over a
 * face's interior, met early by up to (1 - cos) of the radius over an edge or a vertex.  Without the faces a convex body
 * collides with the ground plane only. */
int dmxBatchSetConvexHullFaces(dmxBatchID b, int32_t n_faces, const double *planes);
/* device address of component c of a field for body 0.  The slab is tiled: bodies are stored in tiles of
 * DMX_SLAB_TILE; inside a tile each of the DMX_SLAB_COMPONENTS components holds DMX_SLAB_TILE consecutive
 * reals, so body i's value sits (i / DMX_SLAB_TILE) * DMX_SLAB_COMPONENTS * DMX_SLAB_TILE + i % DMX_SLAB_TILE
 * reals after the returned address.  dmxBatchStride = bodies the slab is allocated for (a multiple of 256).
 * The batch keeps TWO slabs of this layout and the state alternates between them (see dmxBatchSetSnapshotMode): the
 * address is that of the slab holding the current state and is valid until the next step / chunk call. */
#define DMX_SLAB_TILE       64
#define DMX_SLAB_COMPONENTS 30
void *dmxBatchDevicePtr(dmxBatchID b, int field, int component);
int64_t dmxBatchStride(dmxBatchID b);

/* ---- stepping: the tick loop of main.c:211-215 (collide -> step -> clear contacts), nsteps times.
 * Asynchronous on the batch's stream.  QuickStep semantics. */
int dmxBatchStep(dmxBatchID b, double h, int nsteps);
int dmxBatchSynchronize(dmxBatchID b);
/* Island sharding across GPUs: slots [0, n_active) are this rank's own bodies and are stepped;
 * slots [n_active, n) hold read-only ghost copies of neighbouring ranks' boundary bodies.  n_active must be
 * a multiple of 4 (or n).  dmxBatchStepRange steps one tick over [first, first+count) only, so boundary
 * rows can be stepped (and sent) before the interior; first/count must be multiples of 16 B / sizeof(real). */
int dmxBatchSetActiveCount(dmxBatchID b, int64_t n_active);
/* boundary-row pack fused into the step kernels: every whole-slab tick also writes the new 13-real state of bodies
 * [0, lo_count) and [hi_first, n_active) to out_dev (AoS, lower rows first), ready for the all-gather; NULL = off */
int dmxBatchSetBoundaryPack(dmxBatchID b, void *out_dev, int64_t lo_count, int64_t hi_first);
int dmxBatchStepRange(dmxBatchID b, double h, int64_t first, int64_t count, int reset_diag);
/* run on a caller-owned hipStream_t (e.g. the framework's current stream); NULL restores the batch's own */
int dmxBatchSetStream(dmxBatchID b, void *hip_stream);
int dmxBatchGetStream(dmxBatchID b, void **hip_stream);
/* step nsteps times bracketed by HIP events on the batch's stream; *ms = elapsed device milliseconds */
int dmxBatchStepTimed(dmxBatchID b, double h, int nsteps, float *ms);

/* ---- body-body collisions inside dmxBatchStep (dSpaceCollide for body pairs, main.c:212).
 * enable = 1 (default): every tick proves the body-pair set empty through per-body broadphase safe zones, or,
 * when a body leaves its zone / bodies are crowded, runs the exact pair search + narrowphase + island solve
 * for the bodies involved; results equal the sequential oracle's either way.  enable = 0: the caller asserts
 * that bodies never touch one another (every island is a single body) and the check is skipped.
 * stats: [0] ticks in fast mode, [1] ticks in exact mode, [2] safe-zone rebuilds, [3] ticks that had body
 * pairs, [4] body pairs in the last tick, [5] crowded bodies at the last rebuild. */
int dmxBatchSetBodyCollisions(dmxBatchID b, int enable);
/* How a collision-proof chunk keeps its start state for a rollback.
 * DMX_SNAPSHOT_PINGPONG (default): the batch owns two slabs; the chunk's first launch reads one and writes the new state
 * to the other, later launches run in place there, a rollback swaps back -- no copy, no extra HBM traffic.
 * DMX_SNAPSHOT_COPY: the 13 state components are copied aside at the chunk's start and the state never changes slab --
 * for callers that replay captured HIP graphs of ticks, which bake the slab's address in.  Not inside a chunk. */
enum { DMX_SNAPSHOT_PINGPONG = 0, DMX_SNAPSHOT_COPY = 1 };
int dmxBatchSetSnapshotMode(dmxBatchID b, int mode);
/* How an exact tick (bodies in pairs / at static boxes) runs its bookkeeping -- pair list, islands, joints by island, level
 * schedules.  STAGED: a launch per stage over as many compute units as the stage fills.  ONE_WORKGROUP: the same stages, in
 * the same order, inside two one-workgroup kernels around the narrowphase (barriers instead of ~25 launches), whenever the
 * tick's arrays fit one workgroup (<= 8192 slots and entries); AUTO (default; DMX_SMALL_EXACT=0/2 in the environment picks
 * one of the others): ONE_WORKGROUP for scenes of up to 2048 slots whose last tick had at most 512 pairs.  Same results,
 * bit for bit, either way. */
enum { DMX_EXACT_AUTO = 0, DMX_EXACT_STAGED = 1, DMX_EXACT_ONE_WORKGROUP = 2 };
int dmxBatchSetExactPipeline(dmxBatchID b, int mode);
/* Contact-free ticks (no ground plane) may be taken `ticks` at a time inside one kernel launch, the bodies' state held
 * in registers between them: same arithmetic per tick, same results bit for bit, one read and one write of the state
 * per launch instead of per tick.  Default 1 (one launch per tick); 1..64. */
int dmxBatchSetTicksPerLaunch(dmxBatchID b, int ticks);
int dmxBatchCollisionStats(dmxBatchID b, int64_t out[6]);
/* the same six numbers and [6] AABB pairs met so far that had no collider (always 0 since every class pair has one), [7] exact ticks
 * of the one-workgroup pipeline whose island solve and fused step were enqueued before the host had the tick's counts, and stood
 * (the device's record gates them; DMX_SPECULATE=0 in the environment makes every tick wait for its counts first) */
int dmxBatchCollisionStatsEx(dmxBatchID b, int64_t out[8]);

/* ---- the collision-checked tick loop in pieces.  dmxBatchStep(b, h, n) with body collisions enabled runs, inside
 * the library: [zones, snapshot] -> k checked ticks -> one flag read -> commit, or roll back and replay exactly.  A
 * caller that has work of its own between ticks -- the multi-GPU boundary exchange, which refreshes the ghost slots
 * [active count, body count) every tick -- drives the same steps itself:
 *   ChunkBegin   rebuild stale safe zones (ghost slots included), arm the rollback snapshot (dmxBatchSetSnapshotMode),
 *                clear the violation flag;
 *                *exact_only = 1 when the fast path may not be used (crowded bodies, pending external forces),
 *                *ballistic = 1 when bodies move on straight horizontal lines, so checking the chunk's first
 *                and last tick proves the ticks between
 *   ChunkTick    one fused tick of the active bodies, with or without the safe-zone check.  A checked tick of a scene with
 *                a ground plane does nothing once the violation flag is up (the chunk will be rolled back whole: the
 *                only thing a caller may do after a violation)
 *   CheckZonesOnStream  the check alone for slots [first, first+count), on the caller's stream (ghost slots after
 *                their refresh)
 *   RefreshGhostsOnStream  the per-tick ghost refresh in one launch on the caller's stream: the lower neighbour's rows
 *                (13 reals per body, as GatherBodies / the boundary pack lay them out) into slots [first,
 *                first+count_lo), the upper neighbour's into the count_hi slots behind; a NULL source leaves its
 *                range untouched; check = 1 also tests the new positions against the slots' zones
 *   ChunkEnd     wait for the batch stream and report the flags; the caller combines them over ranks
 *   ChunkCommit  account `ticks` fast ticks; refresh_zones = 1 schedules a zone rebuild (the warn flag was up)
 *   ChunkRollback  restore the state the chunk's first tick started from, ghost slots included (zones are rebuilt at the
 *                next ChunkBegin).  Whatever writes state between ticks (the ghost refresh) does so after that tick
 *   ExactTick    one tick with the exact pair search / narrowphase / island solve; DMX_ECROSS if a pair involves
 *                a ghost slot */
int dmxBatchChunkBegin(dmxBatchID b, int *exact_only, int *ballistic);
int dmxBatchChunkTick(dmxBatchID b, double h, int check);
/* n ticks of a ballistic chunk: the test at the run's first / last tick only, as asked */
int dmxBatchChunkTicks(dmxBatchID b, double h, int nticks, int check_first, int check_last);
int dmxBatchCheckZonesOnStream(dmxBatchID b, void *hip_stream, int64_t first, int64_t count);
int dmxBatchRefreshGhostsOnStream(dmxBatchID b, void *hip_stream, int64_t first, int64_t count_lo, const void *src_lo,
                                  int64_t count_hi, const void *src_hi, int check);
int dmxBatchChunkEnd(dmxBatchID b, int *violated, int *warn);
int dmxBatchChunkCommit(dmxBatchID b, int ticks, int refresh_zones);
int dmxBatchChunkRollback(dmxBatchID b);
int dmxBatchExactTick(dmxBatchID b, double h);

/* ---- dSpaceCollide's pair search alone (main.c:212), for the callback form of the tick: the device finds every pair of
 * bodies (i < j, ascending i then j) whose geoms' AABBs overlap, and the bodies "involved" -- in such a pair, or with
 * their AABB overlapping a static box's (dmxBatchSetStaticBoxes) -- ascending.  The arrays are host memory owned by the
 * batch, valid until its next call.  The ODE API face (dSpaceCollide in libode_mi355) feeds the user's near callback from
 * this list when the world is large enough for the device search to pay. */
int dmxBatchFindPairs(dmxBatchID b, const int32_t **pairs, int64_t *n_pairs, const int32_t **involved, int64_t *n_involved);
/* Island sharding: the (own body, ghost slot) pairs the last dmxBatchFindPairs met -- bodies of this rank whose
 * AABB overlaps a neighbouring rank's boundary body.  Their island spans two ranks; the host loop migrates it to one owner
 * (rl-ode-physics_amd/shard.py) before the exact tick, which would otherwise return DMX_ECROSS.  At most 256 are listed. */
int dmxBatchCrossPairs(dmxBatchID b, const int32_t **pairs, int64_t *n_pairs);

/* ---- explicit contact joints: the callback form of the tick.  The reference's near callback makes one
 * dJointCreateContact + dJointAttach per contact (main.c:683-692) and then calls dWorldStep (main.c:213);
 * this entry takes the whole tick's contact joints at once, groups them into dynamics islands, and steps
 * every live body (bodies without joints integrate freely).  body2 = -1 (or body1 = -1) means static
 * geometry (dGeomGetBody == 0, main.c:691); the normal points into body1 as dCollide returns it. */
typedef struct dmxContactJoint {
    double pos[3], normal[3], depth;      /* dContactGeom */
    int32_t body1, body2;                 /* body slots, -1 = none */
    int32_t mode;                         /* dContactBounce | dContactSoftERP | dContactSoftCFM */
    double mu, bounce, bounce_vel, soft_erp, soft_cfm;   /* dSurfaceParameters (main.c:684-687) */
} dmxContactJoint;
int dmxBatchStepJoints(dmxBatchID b, double h, int64_t n_joints, const dmxContactJoint *joints);
/* Which of ODE's two steppers dmxBatchStepJoints is.  DMX_STEPPER_QUICK (default): dWorldQuickStep, QuickStep's SOR sweeps.
 * DMX_STEPPER_EXACT: dWorldStep, the reference's own call (main.c:213): every island's system
 *   A lambda = b + w,  A = J M^-1 J^T + cfm / h,  lo <= lambda <= hi,  w complementary to lambda
 * -- the same contact rows -- is solved exactly (block principal pivoting over a Cholesky of the free block; A is positive
 * definite, so the solution is the one ODE's Dantzig solver reaches): small islands one workgroup each, islands of
 * DMX_LCP_GRID_ROWS (192) rows or more -- the reference's pen holds up to 512 bodies in one island of 2 000 - 2 600 rows
 * (main.c:208,213, inc/body.h:6) -- by a grid-wide blocked factorisation on the matrix cores with the rows that can never
 * clamp eliminated once per tick and the active set carried from tick to tick (csrc/dmx_lcp.hip).  Cost grows with the cube
 * of an island's rows, as dWorldStep's does; a tick with an island above DMX_MAX_EXACT_ROWS (16 384) rows is stepped with
 * the SOR instead (stderr says so).  dmxBatchStep (the BASELINE configs, which name dWorldQuickStep) is not affected. */
enum { DMX_STEPPER_QUICK = 0, DMX_STEPPER_EXACT = 1 };
int dmxBatchSetStepper(dmxBatchID b, int stepper);
/* Counters of the grid-wide exact solve since the batch was created: out[0] island solves, [1] pivoting rounds in all,
 * [2] most rounds in one solve, [3..5] rows / never-clamping rows / bounded rows of the last island solved, [6] rounds that
 * flipped a single row (Murty's rule), [7] ticks stepped with the SOR because an island exceeded DMX_MAX_EXACT_ROWS. */
int dmxBatchLcpStats(dmxBatchID b, int64_t out[8]);
/* The order QuickStep's SOR sweeps an island's rows in (dmxBatchStepJoints, DMX_STEPPER_QUICK).  DMX_ORDER_CREATION (default):
 * the order the contact joints were created in, every sweep -- deterministic, and what lets islands be solved by workgroups
 * under a level schedule.  DMX_ORDER_ODE: what stock ODE does [ODE-recall]: rows numbered in the order its island builder
 * discovers the joints (depth-first from the newest body, each body's joints newest first) and re-shuffled before sweeps
 * 0, 8, 16, ... with ODE's linear congruential generator (RANDOMLY_REORDER_CONSTRAINTS), seeded here per batch (`seed` =
 * dRandSetSeed; ODE's generator is process-global).  Islands are then swept sequentially, one lane each: an option for
 * comparing with ODE's own sequence, not a fast path.  "Newest body" is the highest slot. */
enum { DMX_ORDER_CREATION = 0, DMX_ORDER_ODE = 1 };
int dmxBatchSetRowOrder(dmxBatchID b, int order, uint32_t seed);

/* per-body flags for the island path: dBodyDestroy'ed slots, dBodySetKinematic (main.c:712), gravity / gyro modes */
enum { DMX_BODY_ALIVE = 1, DMX_BODY_KINEMATIC = 2, DMX_BODY_NOGRAVITY = 4, DMX_BODY_NOGYRO = 8 };
int dmxBatchUploadBodyFlags(dmxBatchID b, const uint8_t *flags, int64_t first, int64_t count);

/* ---- diagnostics of the last step */
int dmxBatchLastContactCount(dmxBatchID b, int64_t *n);       /* contact joints created in the last tick */
int dmxBatchLastResidual(dmxBatchID b, double *r);            /* sum |delta lambda| of the last SOR sweep */

/* ---- pose read-back for the 60 Hz snapshot (main.c:221-240): GetTransformMat (main.c:602-622) on
 * device; out_dev/out_host receive count x 16 reals, column-major 4x4 per body */
int dmxBatchPackTransforms(dmxBatchID b, void *out_dev, int64_t first, int64_t count);
int dmxBatchDownloadTransforms(dmxBatchID b, void *out_host, int64_t first, int64_t count);

/* ---- boundary-body exchange for island sharding across GPUs: gather pos+quat+lvel+avel
 * (13 reals) of the listed bodies into a contiguous device buffer (count x 13, AoS) and back */
int dmxBatchGatherBodies(dmxBatchID b, const int32_t *idx_dev, int64_t count, void *out_dev);
int dmxBatchScatterBodies(dmxBatchID b, const int32_t *idx_dev, int64_t count, const void *in_dev);
/* scatter on a caller-chosen hipStream_t (the exchange's side stream): ghost slots are never touched by the step
 * kernels, so they can be refreshed while the batch's own stream integrates the next tick */
int dmxBatchScatterBodiesOnStream(dmxBatchID b, const int32_t *idx_dev, int64_t count, const void *in_dev, void *hip_stream);

const char *dmxVersion(void);

#ifdef __cplusplus
}
#endif
#endif
