// ode_compat.cpp -- the ODE C API subset of include/ode/ode.h on top of the device batch.
//
// What runs where (the split the reference's call pattern forces):
//   host   : object bookkeeping, dSpaceCollide's pair loop and the user's near callback, dCollide (the
//            callback needs its contacts synchronously, /root/reference/src/main.c:678), grouping the
//            tick's contact joints into islands;
//   device : everything dWorldStep / dWorldQuickStep computes -- contact rows, SOR sweeps, velocity
//            update, integration (dmxBatchStepJoints -> solve_islands) -- and the pose snapshot.
// Host mirrors of body state are refreshed lazily: the first getter (or dSpaceCollide) after a step does
// one bulk device->host copy; setters mark the world dirty and the next step uploads.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "../../include/ode/ode.h"
#include "../../include/dmx_batch.h"
#include "dmx_collide.hpp"
#include "dmx_math.hpp"

using dmx::M3;
using dmx::Q4;
using dmx::V3;

namespace {

#ifdef dSINGLE
constexpr int kPrecision = DMX_F32;
constexpr dReal kDefaultCFM = 1e-5f;
#else
constexpr int kPrecision = DMX_F64;
constexpr dReal kDefaultCFM = 1e-10;
#endif

[[noreturn]] void fatal(const char *what, int rc)
{
    fprintf(stderr, "libode_mi355: %s failed (code %d); the step has no CPU fallback\n", what, rc);
    abort();
}
#define DMX_MUST(call) do { int rc_ = (call); if (rc_ != DMX_OK) fatal(#call, rc_); } while (0)

M3<dReal> to_m3(const dReal *R) { return { { { R[0], R[1], R[2] }, { R[4], R[5], R[6] }, { R[8], R[9], R[10] } } }; }
void from_m3(const M3<dReal> &M, dReal *R)
{
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) R[4 * i + j] = M.m[i][j]; R[4 * i + 3] = 0; }
}
void set_identity(dReal *R) { memset(R, 0, 12 * sizeof(dReal)); R[0] = R[5] = R[10] = 1; }

}  // namespace

struct dxGeom;
struct dxJoint;

struct dxBody {
    dxWorld *world = nullptr;
    int slot = -1;
    dVector3 pos = { 0, 0, 0, 0 };
    dMatrix3 R;
    dQuaternion q = { 1, 0, 0, 0 };
    dVector3 lvel = { 0, 0, 0, 0 }, avel = { 0, 0, 0, 0 };
    dVector3 facc = { 0, 0, 0, 0 }, tacc = { 0, 0, 0, 0 };
    dMass mass;
    int flags = DMX_BODY_ALIVE;
    std::vector<dxGeom *> geoms;
};

struct dxGeom {
    int cls = dBoxClass;
    int id = 0;                       // creation order: fixes the pair order of dSpaceCollide
    dxBody *body = nullptr;
    dxSpace *space = nullptr;
    dReal side[3] = { 0, 0, 0 };      // box sides / sphere radius in side[0]
    dReal plane[4] = { 0, 0, 1, 0 };
    dVector3 pos = { 0, 0, 0, 0 };    // pose of a body-less (static) geom, main.c:748-749
    dMatrix3 R;
    unsigned long cat = ~0ul, col = ~0ul;
};

struct dxSpace {
    std::vector<dxGeom *> geoms;
};

struct dxJoint {
    dxWorld *world = nullptr;
    dxJointGroup *group = nullptr;
    dContact contact;
    dxBody *b1 = nullptr, *b2 = nullptr;
};

struct dxJointGroup {
    std::vector<dxJoint *> joints;
};

struct dxWorld {
    dmxBatchID batch = nullptr;
    int cap = 0;
    std::vector<dxBody *> slots;       // slot -> body (nullptr = free)
    std::vector<dxJoint *> joints;     // contact joints of the current tick, creation order
    dReal g[3] = { 0, 0, 0 };
    dReal erp = (dReal)0.2, cfm = kDefaultCFM, sor_w = (dReal)1.3;
    int iters = 20;
    bool dev_newer = false;            // device holds newer body state than the host mirrors
    bool host_dirty = true;            // host mirrors hold changes the device has not seen
    uint32_t order_seed = 0; bool order_seeded = false;     // DMX_ROW_ORDER=ode:<seed>
    bool geom_dirty = true;            // the bodies' geoms (class, extents) changed since the device last saw them
    std::vector<double> dev_statics;   // the static boxes the device holds (18 doubles each), for the device pair search
    std::vector<dReal> buf;
    std::vector<dmxContactJoint> cj;

    void create_batch(int capacity)
    {
        DMX_MUST(dmxBatchCreate(&batch, capacity, kPrecision, 0));
        if (const char *e = getenv("DMX_ROW_ORDER")) {          // "ode" or "ode:<seed>": stock ODE's row order and shuffle for dWorldQuickStep
            if (strncmp(e, "ode", 3) == 0) {
                if (!order_seeded) { order_seed = e[3] == ':' ? (uint32_t)strtoul(e + 4, nullptr, 10) : 0u; order_seeded = true; }
                DMX_MUST(dmxBatchSetRowOrder(batch, DMX_ORDER_ODE, order_seed));
            }
        }
        cap = capacity;
        slots.resize((size_t)cap, nullptr);
        host_dirty = true;
    }
    void push_params()
    {
        DMX_MUST(dmxBatchSetGravity(batch, g[0], g[1], g[2]));
        DMX_MUST(dmxBatchSetERP(batch, erp));
        DMX_MUST(dmxBatchSetCFM(batch, cfm));
        DMX_MUST(dmxBatchSetQuickStep(batch, iters, sor_w));
    }
    void to_host()
    {
        if (!dev_newer) return;
        // one transfer for the whole read-back (13 reals per slot: pos3 quat4 lvel3 avel3), up to the highest live slot
        int hi = 0;
        for (int s = 0; s < cap; s++) if (slots[(size_t)s]) hi = s + 1;
        buf.resize((size_t)hi * 13);
        if (hi > 0) DMX_MUST(dmxBatchDownload(batch, DMX_STATE, buf.data(), 0, hi));
        for (int s = 0; s < hi; s++) {
            dxBody *b = slots[(size_t)s];
            if (!b) continue;
            const dReal *r = &buf[(size_t)s * 13];
            for (int k = 0; k < 3; k++) { b->pos[k] = r[k]; b->lvel[k] = r[7 + k]; b->avel[k] = r[10 + k]; }
            for (int k = 0; k < 4; k++) b->q[k] = r[3 + k];
        }
        for (dxBody *b : slots) {
            if (!b) continue;
            const Q4<dReal> q = { b->q[0], b->q[1], b->q[2], b->q[3] };
            from_m3(dmx::quat_to_R(q), b->R);          // dxStepBody: R = R(q)
            b->facc[0] = b->facc[1] = b->facc[2] = 0;  // the step cleared the accumulators
            b->tacc[0] = b->tacc[1] = b->tacc[2] = 0;
        }
        dev_newer = false;
    }
    void to_device()
    {
        if (!host_dirty) return;
        to_host();
        auto up = [&](int field, int k, auto get) {
            buf.assign((size_t)cap * k, 0);
            for (int s = 0; s < cap; s++) {
                const dxBody *b = slots[(size_t)s];
                for (int j = 0; j < k; j++) buf[(size_t)s * k + j] = b ? get(b, j) : (dReal)(field == DMX_STATE ? (j == 3) : (field == DMX_MASS || field == DMX_INERTIA));   // empty slot: unit quaternion, unit mass
            }
            DMX_MUST(dmxBatchUpload(batch, field, buf.data(), 0, cap));
        };
        // pos3 quat4 lvel3 avel3 in one transfer; the quaternion is already normalised and is stored as is
        up(DMX_STATE, 13, [](const dxBody *b, int j) { return j < 3 ? b->pos[j] : j < 7 ? b->q[j - 3] : j < 10 ? b->lvel[j - 7] : b->avel[j - 10]; });
        up(DMX_MASS, 1, [](const dxBody *b, int) { return b->mass.mass; });
        up(DMX_INERTIA, 3, [](const dxBody *b, int j) { return b->mass.I[5 * j]; });
        up(DMX_FORCE, 3, [](const dxBody *b, int j) { return b->facc[j]; });
        up(DMX_TORQUE, 3, [](const dxBody *b, int j) { return b->tacc[j]; });
        std::vector<uint8_t> fl((size_t)cap, 0);
        for (int s = 0; s < cap; s++) if (slots[(size_t)s]) fl[(size_t)s] = (uint8_t)slots[(size_t)s]->flags;
        DMX_MUST(dmxBatchUploadBodyFlags(batch, fl.data(), 0, cap));
        host_dirty = false;
    }
    // extents and classes of the bodies' geoms, for the device pair search (dSpaceCollide); false when some body's
    // geometry is not one box / sphere geom (the host search handles those worlds)
    bool geometry_to_device();
    void grow()
    {
        to_host();
        std::vector<dxBody *> keep = slots;
        DMX_MUST(dmxBatchDestroy(batch));
        batch = nullptr;
        create_batch(cap * 2);
        std::copy(keep.begin(), keep.end(), slots.begin());
        geom_dirty = true;
        dev_statics.clear();
    }
};

namespace {

int g_next_geom_id = 0;

const dReal *geom_pos(const dxGeom *g) { return g->body ? g->body->pos : g->pos; }
const dReal *geom_R(const dxGeom *g) { return g->body ? g->body->R : g->R; }
void touch(dxBody *b) { b->world->to_host(); b->world->host_dirty = true; }
void sync_geom(const dxGeom *g) { if (g->body) g->body->world->to_host(); }

}  // namespace

bool dxWorld::geometry_to_device()
{
    if (!geom_dirty) return true;
    std::vector<dReal> sides((size_t)cap * 3, 0);
    std::vector<uint8_t> gt((size_t)cap, DMX_GEOM_NONE);
    for (int s = 0; s < cap; s++) {
        const dxBody *b = slots[(size_t)s];
        if (!b || b->geoms.empty()) continue;                  // a body without geoms collides with nothing
        if (b->geoms.size() != 1) return false;
        const dxGeom *g = b->geoms[0];
        if (g->cls == dBoxClass) { gt[(size_t)s] = DMX_GEOM_BOX; for (int k = 0; k < 3; k++) sides[(size_t)s * 3 + k] = g->side[k]; }
        else if (g->cls == dSphereClass) { gt[(size_t)s] = DMX_GEOM_SPHERE; sides[(size_t)s * 3] = g->side[0]; }
        else return false;
    }
    DMX_MUST(dmxBatchUpload(batch, DMX_SIDES, sides.data(), 0, cap));
    DMX_MUST(dmxBatchUploadGeomType(batch, gt.data(), 0, cap));
    geom_dirty = false;
    return true;
}

// ================================================================================ lifecycle
extern "C" void dInitODE(void) { (void)dInitODE2(0); }
extern "C" int dInitODE2(unsigned int)
{
    if (dmxDeviceCount() < 1) {
        fprintf(stderr, "libode_mi355: dInitODE: no MI355X / HIP device visible; this library has no CPU path\n");
        return 0;
    }
    return 1;
}
extern "C" void dCloseODE(void) {}

// ================================================================================ world
extern "C" dWorldID dWorldCreate(void)
{
    dxWorld *w = new dxWorld();
    w->create_batch(512);              // MAX_BODIES, inc/body.h:6; doubles on demand
    return w;
}
extern "C" void dWorldDestroy(dWorldID w)
{
    if (!w) return;
    for (dxBody *b : w->slots) {
        if (!b) continue;
        for (dxGeom *g : b->geoms) g->body = nullptr;
        delete b;
    }
    for (dxJoint *j : w->joints) j->world = nullptr;
    if (w->batch) dmxBatchDestroy(w->batch);
    delete w;
}
extern "C" void dWorldSetGravity(dWorldID w, dReal x, dReal y, dReal z) { w->g[0] = x; w->g[1] = y; w->g[2] = z; }
extern "C" void dWorldGetGravity(dWorldID w, dVector3 g) { g[0] = w->g[0]; g[1] = w->g[1]; g[2] = w->g[2]; }
extern "C" void dWorldSetERP(dWorldID w, dReal erp) { w->erp = erp; }
extern "C" dReal dWorldGetERP(dWorldID w) { return w->erp; }
extern "C" void dWorldSetCFM(dWorldID w, dReal cfm) { w->cfm = cfm; }
extern "C" dReal dWorldGetCFM(dWorldID w) { return w->cfm; }
extern "C" void dWorldSetQuickStepNumIterations(dWorldID w, int n) { w->iters = n; }
extern "C" int dWorldGetQuickStepNumIterations(dWorldID w) { return w->iters; }
extern "C" void dWorldSetQuickStepW(dWorldID w, dReal v) { w->sor_w = v; }
extern "C" dReal dWorldGetQuickStepW(dWorldID w) { return w->sor_w; }

static int world_step(dWorldID w, dReal h, int stepper)
{
    if (!w || !(h > 0)) return 0;
    w->to_device();
    w->push_params();
    DMX_MUST(dmxBatchSetStepper(w->batch, stepper));
    w->cj.clear();
    for (const dxJoint *j : w->joints) {
        dmxContactJoint c;
        const dContact &ct = j->contact;
        for (int k = 0; k < 3; k++) { c.pos[k] = ct.geom.pos[k]; c.normal[k] = ct.geom.normal[k]; }
        c.depth = ct.geom.depth;
        c.body1 = j->b1 ? j->b1->slot : -1;
        c.body2 = j->b2 ? j->b2->slot : -1;
        c.mode = ct.surface.mode;
        c.mu = ct.surface.mu; c.bounce = ct.surface.bounce; c.bounce_vel = ct.surface.bounce_vel;
        c.soft_erp = ct.surface.soft_erp; c.soft_cfm = ct.surface.soft_cfm;
        w->cj.push_back(c);
    }
    DMX_MUST(dmxBatchStepJoints(w->batch, h, (int64_t)w->cj.size(), w->cj.data()));
    w->dev_newer = true;
    return 1;
}
// dWorldQuickStep: QuickStep's SOR sweeps.  dWorldStep -- what the reference calls, main.c:213 -- the same rows with
// every island's LCP solved exactly (include/dmx_batch.h, dmxBatchSetStepper).
extern "C" int dWorldQuickStep(dWorldID w, dReal h) { return world_step(w, h, DMX_STEPPER_QUICK); }
extern "C" int dWorldStep(dWorldID w, dReal h) { return world_step(w, h, DMX_STEPPER_EXACT); }

// ================================================================================ mass
extern "C" void dMassSetZero(dMass *m) { memset(m, 0, sizeof(*m)); }
extern "C" void dMassSetParameters(dMass *m, dReal themass, dReal cgx, dReal cgy, dReal cgz, dReal I11, dReal I22,
                                   dReal I33, dReal I12, dReal I13, dReal I23)
{
    dMassSetZero(m);
    m->mass = themass;
    m->c[0] = cgx; m->c[1] = cgy; m->c[2] = cgz;
    m->I[0] = I11; m->I[5] = I22; m->I[10] = I33;
    m->I[1] = m->I[4] = I12; m->I[2] = m->I[8] = I13; m->I[6] = m->I[9] = I23;
}
extern "C" void dMassSetBoxTotal(dMass *m, dReal total, dReal lx, dReal ly, dReal lz)
{
    dMassSetZero(m);
    m->mass = total;
    m->I[0] = total / (dReal)12.0 * (ly * ly + lz * lz);
    m->I[5] = total / (dReal)12.0 * (lx * lx + lz * lz);
    m->I[10] = total / (dReal)12.0 * (lx * lx + ly * ly);
}
extern "C" void dMassSetBox(dMass *m, dReal density, dReal lx, dReal ly, dReal lz)
{ dMassSetBoxTotal(m, lx * ly * lz * density, lx, ly, lz); }
extern "C" void dMassSetSphereTotal(dMass *m, dReal total, dReal r)
{
    dMassSetZero(m);
    m->mass = total;
    const dReal II = (dReal)0.4 * total * r * r;
    m->I[0] = m->I[5] = m->I[10] = II;
}
extern "C" void dMassSetSphere(dMass *m, dReal density, dReal r)
{ dMassSetSphereTotal(m, (dReal)(4.0 / 3.0 * 3.14159265358979323846) * r * r * r * density, r); }

// ================================================================================ bodies
extern "C" dBodyID dBodyCreate(dWorldID w)
{
    if (!w) return nullptr;
    int slot = -1;
    for (int s = 0; s < w->cap; s++) if (!w->slots[(size_t)s]) { slot = s; break; }
    if (slot < 0) { slot = w->cap; w->grow(); }
    w->to_host();
    dxBody *b = new dxBody();
    b->world = w;
    b->slot = slot;
    set_identity(b->R);
    dMassSetParameters(&b->mass, 1, 0, 0, 0, 1, 1, 1, 0, 0, 0);   // ODE default; the reference never sets mass (F7)
    w->slots[(size_t)slot] = b;
    w->host_dirty = true;
    return b;
}
extern "C" void dBodyDestroy(dBodyID b)
{
    if (!b) return;
    dxWorld *w = b->world;
    w->to_host();
    for (dxGeom *g : b->geoms) g->body = nullptr;
    for (dxJoint *j : w->joints) { if (j->b1 == b) j->b1 = nullptr; if (j->b2 == b) j->b2 = nullptr; }
    w->slots[(size_t)b->slot] = nullptr;
    w->host_dirty = true;
    delete b;
}
extern "C" dWorldID dBodyGetWorld(dBodyID b) { return b->world; }
extern "C" void dBodySetPosition(dBodyID b, dReal x, dReal y, dReal z) { touch(b); b->pos[0] = x; b->pos[1] = y; b->pos[2] = z; }
extern "C" void dBodySetRotation(dBodyID b, const dMatrix3 R)
{
    touch(b);
    memcpy(b->R, R, sizeof(dMatrix3));
    b->R[3] = b->R[7] = b->R[11] = 0;
    Q4<dReal> q = dmx::R_to_quat(to_m3(b->R));
    dmx::normalize(q);
    b->q[0] = q.w; b->q[1] = q.x; b->q[2] = q.y; b->q[3] = q.z;
}
extern "C" void dBodySetQuaternion(dBodyID b, const dQuaternion qq)
{
    touch(b);
    Q4<dReal> q = { qq[0], qq[1], qq[2], qq[3] };
    dmx::normalize(q);
    b->q[0] = q.w; b->q[1] = q.x; b->q[2] = q.y; b->q[3] = q.z;
    from_m3(dmx::quat_to_R(q), b->R);
}
extern "C" void dBodySetLinearVel(dBodyID b, dReal x, dReal y, dReal z) { touch(b); b->lvel[0] = x; b->lvel[1] = y; b->lvel[2] = z; }
extern "C" void dBodySetAngularVel(dBodyID b, dReal x, dReal y, dReal z) { touch(b); b->avel[0] = x; b->avel[1] = y; b->avel[2] = z; }
extern "C" const dReal *dBodyGetPosition(dBodyID b) { b->world->to_host(); return b->pos; }
extern "C" const dReal *dBodyGetRotation(dBodyID b) { b->world->to_host(); return b->R; }
extern "C" const dReal *dBodyGetQuaternion(dBodyID b) { b->world->to_host(); return b->q; }
extern "C" const dReal *dBodyGetLinearVel(dBodyID b) { b->world->to_host(); return b->lvel; }
extern "C" const dReal *dBodyGetAngularVel(dBodyID b) { b->world->to_host(); return b->avel; }
extern "C" void dBodySetKinematic(dBodyID b) { touch(b); b->flags |= DMX_BODY_KINEMATIC; }
extern "C" void dBodySetDynamic(dBodyID b) { touch(b); b->flags &= ~DMX_BODY_KINEMATIC; }
extern "C" int dBodyIsKinematic(dBodyID b) { return (b->flags & DMX_BODY_KINEMATIC) != 0; }
extern "C" void dBodySetGyroscopicMode(dBodyID b, int on) { touch(b); if (on) b->flags &= ~DMX_BODY_NOGYRO; else b->flags |= DMX_BODY_NOGYRO; }
extern "C" int dBodyGetGyroscopicMode(dBodyID b) { return (b->flags & DMX_BODY_NOGYRO) == 0; }
extern "C" void dBodyAddForce(dBodyID b, dReal x, dReal y, dReal z) { touch(b); b->facc[0] += x; b->facc[1] += y; b->facc[2] += z; }
extern "C" void dBodyAddTorque(dBodyID b, dReal x, dReal y, dReal z) { touch(b); b->tacc[0] += x; b->tacc[1] += y; b->tacc[2] += z; }
extern "C" void dBodySetMass(dBodyID b, const dMass *m)
{
    touch(b);
    if (!(m->mass > 0) || !(m->I[0] > 0) || !(m->I[5] > 0) || !(m->I[10] > 0)) {
        fprintf(stderr, "libode_mi355: dBodySetMass: mass and principal inertia must be positive; ignored\n");
        return;
    }
    if (m->I[1] != 0 || m->I[2] != 0 || m->I[6] != 0 || m->c[0] != 0 || m->c[1] != 0 || m->c[2] != 0)
        fprintf(stderr, "libode_mi355: dBodySetMass: off-diagonal inertia / offset centre of mass are not "
                        "supported on the device path; using the diagonal about the body origin\n");
    b->mass = *m;
}
extern "C" void dBodyGetMass(dBodyID b, dMass *m) { *m = b->mass; }

// ================================================================================ spaces / geoms
extern "C" dSpaceID dSimpleSpaceCreate(dSpaceID) { return new dxSpace(); }
extern "C" dSpaceID dHashSpaceCreate(dSpaceID) { return new dxSpace(); }
extern "C" void dSpaceDestroy(dSpaceID s)
{
    if (!s) return;
    for (dxGeom *g : s->geoms) g->space = nullptr;
    delete s;
}
extern "C" int dSpaceGetNumGeoms(dSpaceID s) { return (int)s->geoms.size(); }

static dxGeom *new_geom(dSpaceID space, int cls)
{
    dxGeom *g = new dxGeom();
    g->cls = cls;
    g->id = g_next_geom_id++;
    set_identity(g->R);
    g->space = space;
    if (space) space->geoms.push_back(g);
    return g;
}
extern "C" dGeomID dCreateBox(dSpaceID space, dReal lx, dReal ly, dReal lz)
{
    dxGeom *g = new_geom(space, dBoxClass);
    g->side[0] = lx; g->side[1] = ly; g->side[2] = lz;
    return g;
}
extern "C" dGeomID dCreateSphere(dSpaceID space, dReal radius)
{
    dxGeom *g = new_geom(space, dSphereClass);
    g->side[0] = radius;
    return g;
}
extern "C" dGeomID dCreatePlane(dSpaceID space, dReal a, dReal b, dReal c, dReal d)
{
    dxGeom *g = new_geom(space, dPlaneClass);
    dReal l = a * a + b * b + c * c;
    if (l > 0) { l = (dReal)1 / dmx::tsqrt<dReal>(l); a *= l; b *= l; c *= l; d *= l; }
    else { a = 1; b = 0; c = 0; d = 0; }
    g->plane[0] = a; g->plane[1] = b; g->plane[2] = c; g->plane[3] = d;
    return g;
}
extern "C" void dGeomDestroy(dGeomID g)
{
    if (!g) return;
    if (g->body) { auto &v = g->body->geoms; v.erase(std::remove(v.begin(), v.end(), g), v.end()); g->body->world->geom_dirty = true; }
    if (g->space) { auto &v = g->space->geoms; v.erase(std::remove(v.begin(), v.end(), g), v.end()); }
    delete g;
}
extern "C" void dGeomSetBody(dGeomID g, dBodyID b)
{
    if (g->body == b) return;
    if (g->body) { auto &v = g->body->geoms; v.erase(std::remove(v.begin(), v.end(), g), v.end()); g->body->world->geom_dirty = true; }
    g->body = b;
    if (b) { b->geoms.push_back(g); b->world->geom_dirty = true; }
}
extern "C" dBodyID dGeomGetBody(dGeomID g) { return g->body; }
extern "C" void dGeomSetPosition(dGeomID g, dReal x, dReal y, dReal z)
{
    if (g->body) { dBodySetPosition(g->body, x, y, z); return; }
    g->pos[0] = x; g->pos[1] = y; g->pos[2] = z;
}
extern "C" void dGeomSetRotation(dGeomID g, const dMatrix3 R)
{
    if (g->body) { dBodySetRotation(g->body, R); return; }
    memcpy(g->R, R, sizeof(dMatrix3));
    g->R[3] = g->R[7] = g->R[11] = 0;
}
extern "C" const dReal *dGeomGetPosition(dGeomID g) { sync_geom(g); return geom_pos(g); }
extern "C" const dReal *dGeomGetRotation(dGeomID g) { sync_geom(g); return geom_R(g); }
extern "C" void dGeomSetCategoryBits(dGeomID g, unsigned long bits) { g->cat = bits; }
extern "C" void dGeomSetCollideBits(dGeomID g, unsigned long bits) { g->col = bits; }
extern "C" unsigned long dGeomGetCategoryBits(dGeomID g) { return g->cat; }
extern "C" unsigned long dGeomGetCollideBits(dGeomID g) { return g->col; }
extern "C" int dGeomGetClass(dGeomID g) { return g->cls; }
extern "C" void dGeomBoxGetLengths(dGeomID g, dVector3 r) { r[0] = g->side[0]; r[1] = g->side[1]; r[2] = g->side[2]; }
extern "C" dReal dGeomSphereGetRadius(dGeomID g) { return g->side[0]; }
extern "C" void dGeomPlaneGetParams(dGeomID g, dVector4 r) { for (int i = 0; i < 4; i++) r[i] = g->plane[i]; }

// ---- dCollide ------------------------------------------------------------------------------------
namespace {

struct Hit { V3<dReal> pos, normal; dReal depth; };

// contacts of the ordered class pair (a,b); false when no collider exists for that order
bool collide_ordered(const dxGeom *a, const dxGeom *b, int maxc, Hit *out, int *n)
{
    const dReal *pa = geom_pos(a), *pb = geom_pos(b);
    const V3<dReal> xa = { pa[0], pa[1], pa[2] }, xb = { pb[0], pb[1], pb[2] };
    *n = 0;
    if (a->cls == dBoxClass && b->cls == dPlaneClass) {
        V3<dReal> cp[4]; dReal cd[4];
        const V3<dReal> pn = { b->plane[0], b->plane[1], b->plane[2] };
        *n = dmx::box_plane(xa, to_m3(geom_R(a)), a->side, pn, b->plane[3], maxc, cp, cd);
        for (int i = 0; i < *n; i++) out[i] = { cp[i], pn, cd[i] };
        return true;
    }
    if (a->cls == dSphereClass && b->cls == dPlaneClass) {
        V3<dReal> cp[4]; dReal cd[4];
        const V3<dReal> pn = { b->plane[0], b->plane[1], b->plane[2] };
        *n = dmx::sphere_plane(xa, a->side[0], pn, b->plane[3], cp, cd);
        for (int i = 0; i < *n; i++) out[i] = { cp[i], pn, cd[i] };
        return true;
    }
    if (a->cls == dSphereClass && b->cls == dSphereClass) {
        dmx::ContactPoint<dReal> c;
        *n = dmx::sphere_sphere(xa, a->side[0], xb, b->side[0], &c);
        if (*n) out[0] = { c.pos, c.normal, c.depth };
        return true;
    }
    if (a->cls == dSphereClass && b->cls == dBoxClass) {
        dmx::ContactPoint<dReal> c;
        *n = dmx::sphere_box(xa, a->side[0], xb, to_m3(geom_R(b)), b->side, &c);
        if (*n) out[0] = { c.pos, c.normal, c.depth };
        return true;
    }
    if (a->cls == dBoxClass && b->cls == dBoxClass) {
        dmx::ContactPoint<dReal> c[8];
        *n = dmx::box_box(xa, to_m3(geom_R(a)), a->side, xb, to_m3(geom_R(b)), b->side, maxc, c);
        for (int i = 0; i < *n; i++) out[i] = { c[i].pos, c[i].normal, c[i].depth };
        return true;
    }
    return false;
}

}  // namespace

extern "C" int dCollide(dGeomID o1, dGeomID o2, int flags, dContactGeom *contact, int skip)
{
    if (!o1 || !o2 || !contact || o1 == o2) return 0;
    if (o1->body && o1->body == o2->body) return 0;
    int maxc = flags & 0xffff;
    if (maxc < 1) maxc = 1;
    sync_geom(o1); sync_geom(o2);
    Hit hits[8];
    int n = 0;
    bool flip = false;
    if (!collide_ordered(o1, o2, maxc > 8 ? 8 : maxc, hits, &n)) {
        if (!collide_ordered(o2, o1, maxc > 8 ? 8 : maxc, hits, &n)) return 0;   // e.g. plane-plane
        flip = true;
    }
    if (n > maxc) n = maxc;
    for (int i = 0; i < n; i++) {
        dContactGeom *c = (dContactGeom *)((char *)contact + (size_t)i * (size_t)skip);   // byte stride, main.c:678
        c->pos[0] = hits[i].pos.x; c->pos[1] = hits[i].pos.y; c->pos[2] = hits[i].pos.z; c->pos[3] = 0;
        const dReal sg = flip ? (dReal)-1 : (dReal)1;
        c->normal[0] = flip ? -hits[i].normal.x : hits[i].normal.x;
        c->normal[1] = flip ? -hits[i].normal.y : hits[i].normal.y;
        c->normal[2] = flip ? -hits[i].normal.z : hits[i].normal.z;
        c->normal[3] = 0; (void)sg;
        c->depth = hits[i].depth;
        c->g1 = o1; c->g2 = o2;
        c->side1 = -1; c->side2 = -1;
    }
    return n;
}

// ---- dSpaceCollide -----------------------------------------------------------------------------------
namespace {

struct GeomBox { dReal lo[3], hi[3]; };
GeomBox aabb_of(const dxGeom *g)
{
    const dReal *p = geom_pos(g), *R = geom_R(g);
    GeomBox b;
    for (int i = 0; i < 3; i++) {
        const dReal r = g->cls == dSphereClass
                            ? g->side[0]
                            : (dReal)0.5 * (dmx::tabs(R[4 * i] * g->side[0]) + dmx::tabs(R[4 * i + 1] * g->side[1]) +
                                            dmx::tabs(R[4 * i + 2] * g->side[2]));
        b.lo[i] = p[i] - r; b.hi[i] = p[i] + r;
    }
    return b;
}
bool boxes_overlap(const GeomBox &a, const GeomBox &b)
{
    return !(b.lo[0] > a.hi[0] || a.lo[0] > b.hi[0] || b.lo[1] > a.hi[1] || a.lo[1] > b.hi[1] || b.lo[2] > a.hi[2] || a.lo[2] > b.hi[2]);
}
bool pair_passes(const dxGeom *a, const dxGeom *b)
{
    if (a->body && a->body == b->body) return false;
    return ((a->cat & b->col) || (b->cat & a->col)) != 0;
}

// worlds of at least this many bodies take their body pairs from the device (DMX_COMPAT_DEVICE_PAIRS: 1 = always, 0 = never;
// a number >= 2 = that many bodies).  Below it the host sweep is quicker than a device round trip.
int device_pairs_threshold()
{
    static const int v = [] {
        const char *e = getenv("DMX_COMPAT_DEVICE_PAIRS");
        if (!e) return 2048;
        const int x = atoi(e);
        return x == 0 ? 0x7fffffff : x;
    }();
    return v;
}

// The pair search of dSpaceCollide on the device: body-body AABB pairs and the bodies near static boxes come from
// dmxBatchFindPairs (hashed-grid search, dmx_exact.hip); planes, the static boxes' own pairs and the category / collide
// filter are a few host loops over that short list.  false = this space is not of the shape the device search covers
// (bodies of several worlds, bodies with several geoms, static geoms other than boxes and planes): the host search runs.
bool device_pairs(dxSpace *space, std::vector<std::pair<dxGeom *, dxGeom *>> &pairs)
{
    dxWorld *w = nullptr;
    std::vector<dxGeom *> planes, statics;
    int nbodies = 0;
    for (dxGeom *g : space->geoms) {
        if (g->cls == dPlaneClass) { planes.push_back(g); continue; }
        if (!g->body) { if (g->cls != dBoxClass) return false; statics.push_back(g); continue; }
        if (w && g->body->world != w) return false;
        w = g->body->world;
        nbodies++;
    }
    if (!w || nbodies < device_pairs_threshold() || (int)statics.size() > DMX_MAX_STATIC_BOXES) return false;
    for (const dxBody *b : w->slots) if (b) for (const dxGeom *g : b->geoms) if (g->space != space) return false;
    w->to_device();
    if (!w->geometry_to_device()) return false;
    std::vector<double> st;
    for (const dxGeom *g : statics) {
        for (int k = 0; k < 3; k++) st.push_back(g->side[k]);
        for (int k = 0; k < 3; k++) st.push_back(g->pos[k]);
        for (int k = 0; k < 12; k++) st.push_back(g->R[k]);
    }
    if (st != w->dev_statics) {
        std::vector<double> sides, pos, rot;
        for (size_t s = 0; s < statics.size(); s++) {
            sides.insert(sides.end(), st.begin() + 18 * s, st.begin() + 18 * s + 3);
            pos.insert(pos.end(), st.begin() + 18 * s + 3, st.begin() + 18 * s + 6);
            rot.insert(rot.end(), st.begin() + 18 * s + 6, st.begin() + 18 * s + 18);
        }
        DMX_MUST(dmxBatchSetStaticBoxes(w->batch, (int32_t)statics.size(), sides.data(), pos.data(), rot.data()));
        w->dev_statics = st;
    }
    const int32_t *bp = nullptr, *inv = nullptr;
    int64_t nbp = 0, ninv = 0;
    DMX_MUST(dmxBatchFindPairs(w->batch, &bp, &nbp, &inv, &ninv));
    w->to_host();                                 // dCollide reads poses from the host mirrors: one bulk copy per tick
    auto push = [&](dxGeom *a, dxGeom *b) { if (a->id < b->id) pairs.emplace_back(a, b); else pairs.emplace_back(b, a); };
    for (int64_t k = 0; k < nbp; k++) {
        dxGeom *a = w->slots[(size_t)bp[2 * k]]->geoms[0], *b = w->slots[(size_t)bp[2 * k + 1]]->geoms[0];
        if (pair_passes(a, b)) push(a, b);
    }
    std::vector<GeomBox> sbox;
    for (const dxGeom *g : statics) sbox.push_back(aabb_of(g));
    for (int64_t k = 0; k < ninv; k++) {
        dxGeom *g = w->slots[(size_t)inv[k]]->geoms[0];
        const GeomBox gb = aabb_of(g);
        for (size_t s = 0; s < statics.size(); s++)
            if (boxes_overlap(gb, sbox[s]) && pair_passes(g, statics[s])) push(g, statics[s]);
    }
    for (size_t s = 0; s < statics.size(); s++)               // static-static pairs reach the callback too (SURVEY a-5)
        for (size_t t = s + 1; t < statics.size(); t++)
            if (boxes_overlap(sbox[s], sbox[t]) && pair_passes(statics[s], statics[t])) push(statics[s], statics[t]);
    for (dxGeom *pl : planes)
        for (dxGeom *g : space->geoms)
            if (g->cls != dPlaneClass && pair_passes(pl, g)) push(pl, g);
    return true;
}

}  // namespace

extern "C" void dSpaceCollide(dSpaceID space, void *data, dNearCallback *callback)
{
    if (!space || !callback) return;
    {
        std::vector<std::pair<dxGeom *, dxGeom *>> dpairs;
        if (device_pairs(space, dpairs)) {
            std::sort(dpairs.begin(), dpairs.end(), [](const auto &a, const auto &b) {
                return a.first->id < b.first->id || (a.first->id == b.first->id && a.second->id < b.second->id);
            });
            for (auto &pr : dpairs) callback(data, pr.first, pr.second);          // NearCallback, main.c:674
            return;
        }
    }
    struct Box { dReal lo[3], hi[3]; dxGeom *g; };
    std::vector<Box> bb;
    std::vector<dxGeom *> planes;
    std::vector<std::pair<dxGeom *, dxGeom *>> pairs;
    for (dxGeom *g : space->geoms) sync_geom(g);
    for (dxGeom *g : space->geoms) {
        if (g->cls == dPlaneClass) { planes.push_back(g); continue; }
        const dReal *p = geom_pos(g), *R = geom_R(g);
        Box b; b.g = g;
        for (int i = 0; i < 3; i++) {
            const dReal r = g->cls == dSphereClass
                                ? g->side[0]
                                : (dReal)0.5 * (dmx::tabs(R[4 * i] * g->side[0]) + dmx::tabs(R[4 * i + 1] * g->side[1]) +
                                                dmx::tabs(R[4 * i + 2] * g->side[2]));
            b.lo[i] = p[i] - r; b.hi[i] = p[i] + r;
        }
        bb.push_back(b);
    }
    auto passes = [](const dxGeom *a, const dxGeom *b) {
        if (a->body && a->body == b->body) return false;
        return ((a->cat & b->col) || (b->cat & a->col)) != 0;
    };
    auto push = [&](dxGeom *a, dxGeom *b) { if (a->id < b->id) pairs.emplace_back(a, b); else pairs.emplace_back(b, a); };
    for (dxGeom *pl : planes)
        for (const Box &b : bb) if (passes(pl, b.g)) push(pl, b.g);
    std::sort(bb.begin(), bb.end(), [](const Box &a, const Box &b) { return a.lo[0] < b.lo[0] || (a.lo[0] == b.lo[0] && a.g->id < b.g->id); });
    for (size_t i = 0; i < bb.size(); i++)
        for (size_t j = i + 1; j < bb.size() && bb[j].lo[0] <= bb[i].hi[0]; j++) {
            if (bb[j].lo[1] > bb[i].hi[1] || bb[i].lo[1] > bb[j].hi[1]) continue;
            if (bb[j].lo[2] > bb[i].hi[2] || bb[i].lo[2] > bb[j].hi[2]) continue;
            if (passes(bb[i].g, bb[j].g)) push(bb[i].g, bb[j].g);
        }
    // pair order is fixed by geom creation order so a tick is reproducible
    std::sort(pairs.begin(), pairs.end(), [](const auto &a, const auto &b) {
        return a.first->id < b.first->id || (a.first->id == b.first->id && a.second->id < b.second->id);
    });
    for (auto &pr : pairs) callback(data, pr.first, pr.second);          // NearCallback, main.c:674
}

// ================================================================================ contact joints
extern "C" dJointGroupID dJointGroupCreate(int) { return new dxJointGroup(); }
extern "C" void dJointGroupEmpty(dJointGroupID g)
{
    if (!g) return;
    for (dxJoint *j : g->joints) {
        if (j->world) { auto &v = j->world->joints; v.erase(std::remove(v.begin(), v.end(), j), v.end()); }
        delete j;
    }
    g->joints.clear();
}
extern "C" void dJointGroupDestroy(dJointGroupID g) { dJointGroupEmpty(g); delete g; }
extern "C" dJointID dJointCreateContact(dWorldID w, dJointGroupID g, const dContact *c)
{
    if (!w || !c) return nullptr;
    dxJoint *j = new dxJoint();
    j->world = w;
    j->group = g;
    j->contact = *c;                   // the caller's dContact lives on its stack (main.c:676)
    w->joints.push_back(j);
    if (g) g->joints.push_back(j);
    return j;
}
extern "C" void dJointAttach(dJointID j, dBodyID b1, dBodyID b2)
{
    if (!j) return;
    j->b1 = b1; j->b2 = b2;
}

// ================================================================================ rotation helpers
extern "C" void dRSetIdentity(dMatrix3 R) { set_identity(R); }
extern "C" void dQtoR(const dQuaternion q, dMatrix3 R) { from_m3(dmx::quat_to_R(Q4<dReal>{ q[0], q[1], q[2], q[3] }), R); }
extern "C" void dRtoQ(const dMatrix3 R, dQuaternion q)
{
    const Q4<dReal> r = dmx::R_to_quat(to_m3(R));
    q[0] = r.w; q[1] = r.x; q[2] = r.y; q[3] = r.z;
}
extern "C" void dRFromAxisAndAngle(dMatrix3 R, dReal ax, dReal ay, dReal az, dReal angle)
{
    dReal l = ax * ax + ay * ay + az * az;
    Q4<dReal> q = { 1, 0, 0, 0 };
    if (l > 0) {
        angle *= (dReal)0.5;
        l = (dReal)sin((double)angle) / dmx::tsqrt<dReal>(l);
        q = { (dReal)cos((double)angle), ax * l, ay * l, az * l };
    }
    from_m3(dmx::quat_to_R(q), R);
}

// ================================================================================ snapshot extension
extern "C" int dmxWorldSnapshotTransforms(dWorldID w, const dBodyID *bodies, int n, dReal *out)
{
    if (!w || !bodies || !out || n < 0) return 0;
    w->to_device();
    w->buf.resize((size_t)w->cap * 16);
    DMX_MUST(dmxBatchDownloadTransforms(w->batch, w->buf.data(), 0, w->cap));
    for (int i = 0; i < n; i++) {
        if (!bodies[i] || bodies[i]->world != w) return 0;
        memcpy(out + (size_t)i * 16, w->buf.data() + (size_t)bodies[i]->slot * 16, 16 * sizeof(dReal));
    }
    return 1;
}

// The whole of main.c:221-237 in one call: walks the caller's array of body handles (stride body_stride bytes, as in
// `Body bodies[MAX_BODIES]`, body.h:20-24) and writes each live body's column-major 4x4 into the caller's array of
// states (stride state_stride bytes, as in `BodyState bodyStates[MAX_BODIES]`, body.h:26-31).  Entries whose handle is
// 0 (empty slots, static geoms: main.c:228) are left untouched.  Returns the number of transforms written, -1 on error.
extern "C" int dmxWorldSnapshotBodyStates(dWorldID w, const void *first_body, size_t body_stride, int n,
                                          void *first_transform, size_t state_stride)
{
    if (!w || !first_body || !first_transform || n < 0 || body_stride < sizeof(dBodyID) || state_stride < 16 * sizeof(dReal))
        return -1;
    w->to_device();
    w->buf.resize((size_t)w->cap * 16);
    DMX_MUST(dmxBatchDownloadTransforms(w->batch, w->buf.data(), 0, w->cap));
    int written = 0;
    for (int i = 0; i < n; i++) {
        dBodyID b;
        memcpy(&b, (const char *)first_body + (size_t)i * body_stride, sizeof b);
        if (!b) continue;
        if (b->world != w) return -1;
        memcpy((char *)first_transform + (size_t)i * state_stride, w->buf.data() + (size_t)b->slot * 16, 16 * sizeof(dReal));
        written++;
    }
    return written;
}
