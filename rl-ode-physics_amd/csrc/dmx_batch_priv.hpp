// dmx_batch_priv.hpp -- the batch object behind dmxBatchID, shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <algorithm>
#include <chrono>
#include <functional>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/dmx_batch.h"
#include "dmx_internal.hpp"

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "libode_mi355: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(e_), \
                    __FILE__, __LINE__);                                                           \
            return DMX_EHIP;                                                                       \
        }                                                                                          \
    } while (0)

using namespace dmx;

// a contact joint in canonical form: body1 a live dynamic slot, normal into it (dmx_joints.cpp)
struct DmxCanonicalJoint { int b1, b2; const dmxContactJoint *j; bool rev; };

struct dmxBatch {
    int64_t n = 0, stride = 0;
    void *pack_out = nullptr; int64_t pack_lo = 0, pack_hi = 0;   // dmxBatchSetBoundaryPack
    int64_t n_active = 0;            // bodies [0, n_active) are stepped; the rest are ghost slots
    int precision = DMX_F32;
    int device = 0;
    size_t rsize = 4;
    void *slab = nullptr;            // C_COUNT x stride reals: the CURRENT slab (state, constants, zones)
    // Second slab of the same layout.  Constants and zones are kept identical in both (every writer of those
    // components writes both); the state lives in `slab`.  The first launch of a collision-proof chunk reads `slab`
    // and writes the new state to `slab_alt`, then the two swap roles: the chunk's start state stays behind,
    // untouched, as the rollback snapshot (rollback = swap back) at no copy and no extra traffic.
    void *slab_alt = nullptr;
    bool flip_armed = false;         // the next fast launch starts a chunk: write out of place and swap
    bool flipped = false;            // this chunk has swapped: its snapshot is `slab_alt`
    int snapshot_mode = DMX_SNAPSHOT_PINGPONG;
    int snap_kind = 0;               // how the chunk in flight keeps its start state (dmx_general.cpp: SNAP_*)
    // a collision-proof chunk left open by dmxBatchStep (dmx_general.cpp "lazy chunks"): its calls so far, to replay
    // them after a rollback; closed by dmx_settle
    struct OpenChunk {
        bool open = false, ballistic = false, last_checked = false;
        int ticks = 0, budget = 0;
        std::vector<std::pair<double, int>> segs;       // (h, ticks) per Step call, in order
    } oc;
    bool lazy_chunks = true;         // DMX_LAZY_CHUNKS=0 reads every chunk's flag before dmxBatchStep returns
    uint8_t *gtype = nullptr;        // stride bytes
    StepDiag *diag = nullptr;        // device, one slot per wave of the fused step
    StepDiag *diag_isl = nullptr;    // device, island path (atomics)
    size_t n_diag = 0;
    bool last_islands = false;       // which diagnostics the last tick wrote
    StepDiag *diag_host = nullptr;   // pinned
    void *stage = nullptr;           // device staging between host-order rows and the tiled slab
    size_t stage_bytes = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    // an exact tick's fused step for the bodies outside the islands runs beside the island solve (disjoint bodies): a second
    // stream forked from `stream` after the tick's record and joined behind the solve (dmx_general.cpp: careful_tick)
    hipStream_t fork_stream = nullptr; hipEvent_t fork_ev = nullptr, join_ev = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // world parameters (defaults: dWorldCreate [ODE], gravity unset = 0)
    double g[3] = { 0, 0, 0 };
    double erp = 0.2, cfm = 1e-5, sor_w = 1.3;
    int iters = 20;
    int gyro = DMX_GYRO_IMPLICIT;
    int plane_on = 0;
    double plane[4] = { 0, 1, 0, 0 };
    int surf_mode = DMX_CONTACT_BOUNCE;                 // main.c:684
    double mu = __builtin_huge_val(), bounce = 0.2, bounce_vel = 0.1;   // main.c:685-687
    int max_contacts = 8;                               // main.c:675
    bool ext_pending = false;
    // host scratch of the island grouping, persistent between ticks (dmx_joints.cpp) and of the exact tick (dmx_general.cpp)
    std::vector<int> sc_parent, sc_island, sc_last, sc_slots;
    std::vector<int> sc_iv[16];                 // work arrays of the island grouping (dmx_joints.cpp)
    std::vector<DmxCanonicalJoint> sc_cj;
    std::vector<uint8_t> sc_include;            // per slot: 1 while the body is in this tick's island subset
    const int32_t *sc_include_list = nullptr;   // that subset as an ascending list (set around dmx_step_joints by the exact tick)
    int64_t sc_include_count = 0;
    // host-side phase timers of the exact tick (seconds), printed at destroy when DMX_HOST_PROFILE is set
    double prof[12] = { 0 }; bool prof_on = false;
    int ticks_per_launch = 1;        // contact-free ticks fused into one integrate_free launch (dmxBatchSetTicksPerLaunch)
    int min_waves = 0;               // DMX_MIN_WAVES launch-tuning override (see StepParams)
    int nt = 0;                      // DMX_NT launch-tuning override (see StepParams)
    int oop = 0;                     // DMX_OOP=1: the plain (unchecked) loop writes every tick out of place, alternating slabs (experiment)
    int vec = 0;                     // DMX_VEC launch-tuning override (bodies per lane in integrate_free), 0 = default (1)
    bool stepped_with_plane = false;
    // general island path (explicit contact joints)
    uint8_t *bflags = nullptr;                 // device, per-slot BF_* flags
    std::vector<uint8_t> h_bflags;             // host mirror
    struct DevBuf { void *p = nullptr; size_t bytes = 0; };
    DevBuf jd_int, jd_real, jd_rows, jd_rowjb, jd_bscr, jd_local;   // device staging / scratch
    DevBuf jd_lcp, jd_lcp_off, jd_lcp_int;     // dWorldStep's exact island solve: A, factor, vectors per island; offsets; pivoting state
    std::vector<long long> sc_lcp_off;
    std::vector<int> sc_nbd_of, sc_grid_list; std::vector<std::pair<uint64_t, int>> sc_pair_ord;
    void *lcp_grid = nullptr;                  // dWorldStep's grid-wide solve of large islands (dmx_lcp.hip): buffers, remembered active sets
    double exs_acc[64] = { 0 }; long exs_ticks = 0;      // DMX_EXS_TIMING: stage times of the small-scene exact tick, summed
    void *ex_counts_dev = nullptr, *bp_flags_dev = nullptr;    // device-visible addresses of ex_counts_host / bp_flags_host
    int exact_pipeline = 0;         // dmxBatchSetExactPipeline (DMX_EXACT_*); the environment's DMX_SMALL_EXACT is the default
    bool bp_fresh = false;          // the safe zones were built at exactly the current poses (no tick since)
    bool snap_fresh = false;        // ... as of the open snapshot
    bool bp_skip_fast = false;      // the next chunk goes the exact way without trying the fast one (a retry would repeat a failure)
    bool stepper_exact = false;                // dmxBatchSetStepper: dmxBatchStepJoints solves every island's LCP exactly
    bool row_order_ode = false;                // dmxBatchSetRowOrder: ODE's joint-discovery row order + periodic LCG shuffle (QuickStep)
    uint32_t ode_rand = 0;                     // that LCG's state (ODE's is process-global; here it belongs to the batch)
    std::vector<int> sc_ode_proc, sc_ode_iv[4], sc_ode_order;
    std::vector<uint8_t> sc_ode_tag_b, sc_ode_tag_j;
    DevBuf jd_order;
    void *jh_int = nullptr, *jh_real = nullptr;                      // pinned host staging
    // body-body broadphase (dmx_broadphase.hip / dmx_general.cpp)
    int bp_enabled = 1;                        // dmxBatchSetBodyCollisions
    bool bp_valid = false;                     // safe zones match the current constant data
    int bp_chunk = 0;                          // current fast-chunk length in ticks (adaptive, dmx_general.cpp)
    uint32_t bp_crowded = 0;                   // bodies whose safe radius is <= 0 at the last build
    double bp_rmax = 0, bp_rcls[4] = { 0, 0, 0, 0 };
    uint32_t bp_mask = 0; int bp_cap = 8; int bp_xbits = -1;      // -1: not chosen yet
    DevBuf bp_count, bp_items, bp_flags, bp_inpair, bp_snapshot;
    // device-resident bookkeeping of the exact tick (dmx_exact.hip): capacity estimates carried from tick to tick, one arena
    // for the pipeline's arrays, per-body scan arrays, the per-slot level scratch, the pinned read-back record
    uint32_t ex_cap_pairs = 0, ex_cap_rows = 0;
    uint32_t ex_prev_inv = 1;                   // bodies the last exact tick found involved (in a pair / at a static box)
    DevBuf ex_arena, ex_body, ex_last, ex_aabb;
    void *ex_counts_host = nullptr;
    std::vector<int32_t> fp_pairs, fp_inv, fp_cross;      // dmxBatchFindPairs' results; (own body, ghost slot) pairs it met
    uint32_t *bp_flags_host = nullptr;         // pinned
    std::vector<double> h_sides;               // host mirror of DMX_SIDES (exact values of the batch precision)
    std::vector<uint8_t> h_gtype;
    // convex bodies: the shared hull's body-frame points, and the per-tick plane contacts of every convex body
    DevBuf hull, cbuf, ccount;
    DevBuf hull_planes; int hull_nf = 0;       // the hull's faces (dmxBatchSetConvexHullFaces): 4 reals each
    bool spec_refused = false;                 // the last exact tick's record refused (or would have refused) the speculative launches
    int64_t stat_spec_ticks = 0;               // exact ticks whose solve + fused step went out before the host had the counts, and stood
    int64_t stat_unsupported = 0;              // AABB pairs met that have no collider (convex-convex, convex-sphere)
    DevBuf sbox; int n_static = 0;             // static box geoms (dmxBatchSetStaticBoxes), SBOX_REALS reals each
    // the fused path of bodies at static geometry (np_static -> step_contacts): per-body contact buffer and counts
    DevBuf sbuf, scount;
    uint32_t class_pairs = CLASS_PAIRS_ALL;     // dmxBatchSetClassPairs: which geometry classes collide with which
    bool has_simple = true;                    // some slot is a box or a sphere (kept by dmxBatchUploadGeomType)
    int64_t n_simple = 0;                      // ... how many (h_gtype starts all GEOM_NONE)
    bool static_fast = true;                   // DMX_STATIC_FAST=0: every body at a static box goes through the exact tick (round 2's way)
    bool static_need8 = false;                 // a body with 5..8 static contacts has been met: the second step_contacts launch rides along
    int nofast_hold = 0, nofast_level = 0;     // chunks to go the exact way after a body overflowed the contact buffer (backs off)
    int hull_n = 0;
    int64_t stat_rollbacks = 0;
    int64_t stat_fast_ticks = 0, stat_careful_ticks = 0, stat_rebuilds = 0, stat_pair_ticks = 0;
    unsigned long long last_pairs = 0;
    bool last_mixed = false;                   // last tick used fused + island kernels together
    size_t jh_int_bytes = 0, jh_real_bytes = 0;
};

int dmx_ensure_dev(dmxBatch::DevBuf &d, size_t bytes);
// do the fused kernels make contacts (and so leave contact diagnostics behind)?  The ground plane, or static boxes on the fused path
inline bool dmx_fused_contacts(const dmxBatch *b) { return b->plane_on != 0 || (b->n_static > 0 && b->static_fast); }
// f(begin, end, thread) over contiguous chunks of [0, n) on up to DMX_HOST_THREADS (default: the host's cores, at most 16)
// threads of a persistent pool (dmx_host_pool.cpp); runs inline when n is below two grains.  The chunks must touch
// disjoint data.
namespace dmx { void host_pool_run(int nt, const std::function<void(int)> &f); }
template <class F> inline void dmx_parallel_for(int64_t n, int64_t grain, F f)
{
    static const int max_threads = [] {
        const char *e = getenv("DMX_HOST_THREADS");
        int t = e ? atoi(e) : (int)std::thread::hardware_concurrency();
        return t < 1 ? 1 : (t > 16 ? 16 : t);
    }();
    const int nt = (int)std::min<int64_t>(max_threads, n / (grain > 0 ? grain : 1));
    if (nt <= 1) { if (n > 0) f((int64_t)0, n, 0); return; }
    const int64_t per = (n + nt - 1) / nt;
    dmx::host_pool_run(nt, [&](int t) {
        const int64_t lo = t * per, hi = std::min(n, lo + per);
        if (lo < hi) f(lo, hi, t);
    });
}

struct DmxPhase {          // adds the scope's wall time to b->prof[k]
    dmxBatch *b; int k; std::chrono::steady_clock::time_point t0;
    DmxPhase(dmxBatch *bb, int kk) : b(bb), k(kk) { if (b->prof_on) t0 = std::chrono::steady_clock::now(); }
    ~DmxPhase() { if (b->prof_on) b->prof[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};
// contact geometry that already lives on the device (device narrowphase): joint k's pos/normal/depth are entry src[k]
struct DevGeometry { const void *pos, *normal, *depth; const int32_t *src; };
int dmx_step_joints(dmxBatch *b, double h, int64_t n_joints, const dmxContactJoint *joints, const uint8_t *include,
                    const DevGeometry *geo);
// body-body collision handling of the batch tick (dmx_general.cpp)
int dmx_step_collide(dmxBatch *b, double h, int nsteps);
// close the chunk dmxBatchStep may have left open (flag read; rollback + replay on a violation).  Every entry point that
// observes or changes the batch calls this first.
int dmx_settle(dmxBatch *b);
// the collision-checked loop in pieces (dmx_general.cpp)
int dmx_chunk_begin(dmxBatch *b, int *exact_only, int *ballistic);
int dmx_chunk_tick(dmxBatch *b, double h, int check);
int dmx_chunk_ticks(dmxBatch *b, double h, int n, int check_first, int check_last);
int dmx_check_zones(dmxBatch *b, hipStream_t st, int64_t first, int64_t count);
int dmx_chunk_end(dmxBatch *b, int *violated, int *warn);
int dmx_chunk_commit(dmxBatch *b, int ticks, int refresh_zones);
int dmx_chunk_rollback(dmxBatch *b);
int dmx_exact_tick(dmxBatch *b, double h);
int dmx_find_pairs(dmxBatch *b);           // -> b->fp_pairs / b->fp_inv


template <class T> inline void dmx_normalize_plane(const double in[4], T out[4])
{
    // dCreatePlane normalises (a,b,c,d) by |(a,b,c)| in the library's precision
    T a = (T)in[0], bb = (T)in[1], c = (T)in[2], d = (T)in[3];
    T l = a * a + bb * bb + c * c;
    if (l > 0) { l = T(1) / tsqrt<T>(l); a *= l; bb *= l; c *= l; d *= l; }
    else { a = 1; bb = 0; c = 0; d = 0; }
    out[0] = a; out[1] = bb; out[2] = c; out[3] = d;
}


template <class T> inline StepParams<T> dmx_make_params(dmxBatch *b, double h)
{
    StepParams<T> P;
    P.g = { (T)b->g[0], (T)b->g[1], (T)b->g[2] };
    P.h = (T)h;
    P.erp = (T)b->erp; P.cfm = (T)b->cfm; P.sor_w = (T)b->sor_w;
    P.iters = b->iters;
    P.gyro = b->gyro;
    P.plane_on = b->plane_on;
    T pl[4];
    dmx_normalize_plane<T>(b->plane, pl);
    P.pn = { pl[0], pl[1], pl[2] }; P.pd = pl[3];
    P.surf_mode = b->surf_mode;
    P.mu = (T)b->mu; P.bounce = (T)b->bounce; P.bounce_vel = (T)b->bounce_vel;
    P.max_contacts = b->max_contacts;
    P.vec = b->vec;
    P.min_waves = b->min_waves;
    P.nt = b->nt;
    { static const int nf = [] { const char *e = getenv("DMX_HULL_FILTER"); return !e ? 0 : atoi(e) == 0 ? 1 : atoi(e) == 2 ? 2 : 0; }(); P.hull_nofilter = nf; }
    P.bp_check = 0;          // set by the collision-aware tick (dmx_general.cpp)
    P.ticks = 1;
    P.bp_flags = nullptr;
    P.skip = nullptr;
    P.pack_out = (T *)b->pack_out; P.pack_lo = b->pack_lo; P.pack_hi = b->pack_hi;
    P.sbox = (const T *)b->sbox.p; P.n_static = b->n_static;
    P.sbuf = b->static_fast ? (T *)b->sbuf.p : nullptr; P.scount = (int *)b->scount.p;
    P.has_simple = b->has_simple ? 1 : 0;
    P.have8 = 1;             // (a collision-checked launch may leave the 5..8-contact launch out: fused_tick, dmx_general.cpp)
    P.hull = (const T *)b->hull.p; P.hull_n = b->hull_n;
    P.hull_planes = (const T *)b->hull_planes.p; P.hull_nf = b->hull_nf;
    P.cbuf = (T *)b->cbuf.p; P.ccount = (int *)b->ccount.p;
    return P;
}

