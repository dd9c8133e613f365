// dmx_internal.hpp -- layout constants and launch interface shared by the kernels
// (dmx_kernels.hip), the batch C ABI (dmx_batch.cpp) and the ODE-compatible API (ode_compat.cpp).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dmx_math.hpp"

namespace dmx {

// Body slab layout: tiles of SLAB_TILE consecutive bodies; inside a tile the C_COUNT components are stored one
// after another, SLAB_TILE reals each (array-of-structures-of-arrays).  A wavefront stepping 64 consecutive
// bodies therefore reads and writes ONE contiguous run of memory (17 x 256 B in, 13 x 256 B out in f32)
// instead of 30 streams a whole component array apart: on MI355X that lifts the free-flight pass from
// about 5.6 to about 7.1 TB/s at 1 Mi bodies (scripts/ubench_layout.hip, profiles/r01_layout_ubench.txt).
// The slab is allocated for `stride` bodies = the body count rounded up to 256; pad bodies are valid,
// inert memory (mass 1, unit quaternion).
constexpr int SLAB_TILE = 64, SLAB_TILE_LOG2 = 6;
static_assert((1 << SLAB_TILE_LOG2) == SLAB_TILE, "tile size is a power of two");
enum : int {
    C_POS = 0,       // 3
    C_QUAT = 3,      // 4  (w,x,y,z)
    C_LVEL = 7,      // 3
    C_AVEL = 10,     // 3
    C_MASS = 13,     // 1   -- everything below is read-only during a step
    C_INERTIA = 14,  // 3
    C_SIDES = 17,    // 3
    C_FORCE = 20,    // 3   -- external accumulators, cleared by the step
    C_TORQUE = 23,   // 3
    C_BPX = 26,      // broadphase safe zone: build position x, z and radius (see dmx_broadphase.hip)
    C_BPZ = 27,
    C_BPSAFE = 28,
    C_BPR = 29,      // bounding-sphere radius (refreshed by bp_insert)
    C_COUNT = 30
};

// index of component c of body i in the slab
__host__ __device__ __forceinline__ int64_t slab_ix(int c, int64_t i)
{
    return (i >> SLAB_TILE_LOG2) * (int64_t)(C_COUNT * SLAB_TILE) + (int64_t)c * SLAB_TILE + (i & (SLAB_TILE - 1));
}

// GEOM_CONVEX: the batch's one convex hull (StepParams::hull); its bounding radius sits in sides[0] like a sphere's
enum : int { GEOM_NONE = 0, GEOM_SPHERE = 1, GEOM_BOX = 2, GEOM_CONVEX = 3 };
// can a geom of class a ever produce a contact with one of class b?  Every pair of classes has a collider since round 3 (box /
// sphere / convex hull against one another: dmx_collide.hpp, dmx_collide_wave.hpp); empty slots collide with nobody, and the
// caller may switch class pairs off -- the batch's form of dGeomSetCategoryBits / dGeomSetCollideBits (main.c:724-725), which the
// reference sets per geom: bit (4 a + b) of `mask` says that classes a and b collide (dmxBatchSetClassPairs; symmetric).
constexpr uint32_t CLASS_PAIRS_ALL = 0xffffu;
__host__ __device__ __forceinline__ bool classes_collide(int a, int b, uint32_t mask)
{
    return a != GEOM_NONE && b != GEOM_NONE && ((mask >> (4 * a + b)) & 1u) != 0u;
}
constexpr int CONVEX_MAXC = 8;      // contact slots per convex body per tick (the reference's MAX_CONTACTS, main.c:675)
enum : int { SURF_BOUNCE = 0x004, SURF_SOFT_ERP = 0x008, SURF_SOFT_CFM = 0x010 };
// per-slot body flags (uint8 array)
enum : int { BF_ALIVE = 1, BF_KINEMATIC = 2, BF_NOGRAVITY = 4, BF_NOGYRO = 8 };

// Contact joints grouped by dynamics island, as the general island step consumes them.  Islands list
// their bodies (slot indices, ascending) and their contacts (creation order); body 1 of a contact is
// always a dynamic slot, body 2 is a slot or -1 (static geometry), the normal points into body 1.
template <class T> struct IslandSet {
    int n_islands;
    const int *body_off;   // [n_islands+1] into bodies
    const int *bodies;
    const int *con_off;    // [n_islands+1] into the contact arrays
    const int *row_off;    // [n_islands+1] into rows (3 rows reserved per contact)
    const T *cpos, *cnormal, *cdepth;             // 3, 3, 1 per contact
    const int *csrc;       // optional: contact c's geometry lives at index csrc[c] of gpos/gnormal/gdepth (device narrowphase output)
    const T *gpos, *gnormal, *gdepth;
    const int *cb1, *cb2, *cmode;
    const T *cmu, *cbounce, *cbounce_vel, *csoft_erp, *csoft_cfm;      // all null: every contact carries the batch's surface (StepParams)
    T *rows;               // scratch: ISLAND_ROW_REALS reals per row (29 fields), the base aligned to 256 bytes
    int *rowjb;            // scratch: 2 ints per row
    T *bscr;               // scratch: 28 reals per island body
    int *local;            // scratch: per slot, index of the body inside its island
    // large islands (solve_island_wg): per island -1 or the start of its level offsets in lev_off; per large island its
    // island index and level count; lev_rows = island-relative row indices grouped by level; crow = first row of a contact
    const int *big; int n_big; const int *big_list; const int *lev_count; const int *lev_off; const int *lev_rows;
    const int *crow;
    // launch shape of solve_island_wg: bodies of the largest such island (its accumulators go to LDS when they fit) and
    // the widest level of any schedule (64 lanes per island are enough when no level is wider)
    int big_max_bodies, big_max_width;
    int big_rows_total;    // rows of all large islands together (0: not known): bounds the schedule one of them may bring into LDS
    int big_max_rows;      // rows of the largest of them (0: not known): up to 256 x 8 (f32) a workgroup keeps them all in registers
    const int *row_level;  // level of every scheduled row, laid out like lev_rows (an island's rows start at its lev_off[0])
    const int *order;      // optional (dmxBatchSetRowOrder, DMX_ORDER_ODE): sweep `it` visits row order[(it / 8) * order_stride +
    int order_stride;      //   row_off[island] + i] at its i-th step; null: rows in creation order (solve_islands only)
    int singles;           // 1: islands of one body with 1..8 contacts are left to solve_singles / solve_singles_lds (one lane each);
                           // whoever builds `big` must then keep such islands out of it
};

template <class T> struct StepParams {
    V3<T> g;            // gravity
    T h;                // step size
    T erp, cfm, sor_w;  // world ERP, CFM, SOR over-relaxation
    int iters;          // QuickStep iterations
    int gyro;           // 0 off, 1 explicit, 2 implicit
    int plane_on;       // ground half-space present
    V3<T> pn; T pd;     // plane n.x = d
    int surf_mode; T mu, bounce, bounce_vel;   // contact surface (NearCallback, main.c:684-687)
    int max_contacts;
    int vec;            // launch tuning (env DMX_VEC): bodies per lane in integrate_free (0 = default, one)
    int min_waves;      // launch tuning (env DMX_MIN_WAVES): waves per SIMD the register allocator must leave room for, 0 = default
    int nt;             // launch tuning (env DMX_NT): bit 0 = non-temporal state stores, bit 1 = non-temporal loads (integrate_free)
    int hull_nofilter;  // DMX_HULL_FILTER=0: the hull colliders' conservative filter lets every point through (A/B, diagnosis)
    int bp_check;       // safe-zone test of every body's pre-step position (BPC_* bits; any bit = "this tick" for one-tick kernels)
    int ticks;          // integrate_free: ticks taken by one launch with the state held in registers (>= 1)
    uint32_t *bp_flags; // device flags (BPF_*), written when a body has left its safe zone
    const uint8_t *skip; // per-body: 1 = stepped by the island path this tick, leave untouched (may be null)
    // boundary-row pack for the multi-GPU exchange (null = off): bodies i < pack_lo and i >= pack_hi also write
    // their new 13-real state to pack_out[slot*13 ..], slot = i (lower row) or pack_lo + i - pack_hi (upper row)
    T *pack_out; int64_t pack_lo, pack_hi;
    // convex bodies: the shared hull (hull_n body-frame points, 3 reals each) and the tick's ground-plane contacts of
    // every convex body, written by np_convex_plane and read by step_plane: cbuf[i][k] = (x, y, z, depth), ccount[i]
    const T *hull; int hull_n;
    const T *hull_planes; int hull_nf;      // the hull's faces: unit outward normal + offset, body frame (box-convex collider)
    T *cbuf; int *ccount;
    // static (body-less) box geoms, AddBodyMap main.c:735-761: SBOX_REALS reals each (see SBOX_*); the safe-zone test of the
    // fused kernels also asks that a body's bounding sphere stays clear of every static box's AABB
    const T *sbox; int n_static;
    // the fused path of bodies at static boxes (np_static / np_convex_static -> step_contacts, dmx_kernels.hip): every body's
    // contacts with the ground plane and the static boxes, creation order, canonical form (normal into the body), in tiles
    // like the slab's (sc_ix); scount[i] = their number, SC_MAXC + 1 = more than the buffer holds.  have8: the launch
    // for bodies with 5..8 contacts follows the one for 0..4 (otherwise the latter reports BPF_NEED8 when it meets one)
    T *sbuf; int *scount; int have8;
    int has_simple;     // the batch has box / sphere bodies (0: np_static, which serves those, need not be launched)
    // a launch enqueued before the host knows whether it should run (careful_tick's speculative tick): the fused kernels
    // return at once unless *gate != 0 (ExactCounts::spec_ok on the device).  Null: no question asked.
    const uint32_t *gate = nullptr;
};
// contact buffer of the static fused path: SC_MAXC contacts of SC_REALS reals per body, field f of contact k of body i at
// sbuf[sc_ix(k, f, i)] -- tiles of 64 bodies, so a wavefront's access to one field of one contact is one contiguous run
constexpr int SC_MAXC = 8;
enum : int { SC_POS = 0, SC_NORMAL = 3, SC_DEPTH = 6, SC_REALS = 7 };
__host__ __device__ __forceinline__ int64_t sc_ix(int k, int f, int64_t i)
{
    return ((i >> SLAB_TILE_LOG2) * (int64_t)(SC_MAXC * SC_REALS) + (int64_t)(k * SC_REALS + f)) * SLAB_TILE + (i & (SLAB_TILE - 1));
}
// reals per constraint row in the island kernels' row scratch (IslandSet::rows): 29 fields, padded to one aligned line
constexpr int ISLAND_ROW_REALS = 32;
// islands of up to this many rows: one wavefront of solve_island_wg, rows in registers, only row_level of the schedule read
constexpr int WAVE_ISLAND_ROWS = 256;
// layout of one static box in StepParams::sbox / GridParams::sbox
enum : int { SBOX_POS = 0, SBOX_R = 3, SBOX_SIDE = 12, SBOX_LO = 15, SBOX_HI = 18, SBOX_REALS = 24 };
constexpr int MAX_STATIC_BOXES = 64;

// StepParams::bp_check for a launch of `ticks` ticks: test at every tick, or only at the launch's first / last tick
enum : int { BPC_ALL = 1, BPC_FIRST = 2, BPC_LAST = 4 };
// hashed (x,z)-column grid of the body-body broadphase
// (BPF_VIOLATION .. BPF_NEED8 are the chunk's flags, cleared together when a chunk begins.  BPF_NOFAST: a body has more contacts
// with static geometry than the fused path's buffer holds -- only the exact path can step it; BPF_NEED8: a body has 5..8 and
// the launch for those was not enqueued.  Both come with BPF_VIOLATION.)
enum : int { BPF_OVERFLOW = 0, BPF_CROWDED = 1, BPF_NPAIRS = 2, BPF_VIOLATION = 3, BPF_WARN = 4, BPF_NOFAST = 5, BPF_NEED8 = 6, BPF_COUNT = 7 };
constexpr int BPF_CHUNK_FLAGS = 4;      // VIOLATION, WARN, NOFAST, NEED8
// per body, left by grid_insert for the exact pair search: its column and its AABB in one aligned record (32 B f32, 64 B f64),
// so a candidate costs one access, not eight
template <class T> struct alignas(16) GridRec { T lo[3], hi[3]; int32_t ix, iz; };
template <class T> struct GridParams {
    T cell, inv_cell, r_max;
    T r_cls[4];            // largest bounding radius per geometry class (0: the batch has none of it): what a body of a class can
                           // meet outside its 3x3 block of columns is bounded by the classes it collides with (classes_collide)
    uint32_t mask;         // table size - 1 (power of two)
    int xbits;             // > 0: the table is a 2-D torus of 2^xbits columns per row (neighbouring cells are neighbouring
                           // entries: coalesced lookups); 0: scrambled hash (scenes too elongated for the torus)
    int cap;               // bodies per bucket
    uint32_t *count;       // [mask+1]
    int32_t *items;        // [(mask+1) * cap]
    uint32_t *flags;       // [BPF_COUNT]
    GridRec<T> *rec;       // optional [per slot]: bp_insert leaves every body's column and AABB here for the exact pair search
    const T *sbox; int n_static;   // static boxes (StepParams::sbox)
    // 1: bodies at static boxes take the fused path (step_contacts) unless their contacts might not fit its buffer -- AABB
    // over two or more static boxes, or over one with a ground plane present: only those are "involved" in an exact tick,
    // and being near a static box does not make a body crowded.  0 (DMX_STATIC_FAST=0): every body at a static box is.
    int static_fast, plane_on;
    const T *hull; int hull_n;     // the batch's hull shape (convex bodies' exact AABBs for the pair search: bp_convex_aabb)
    uint32_t class_pairs;          // which geometry classes collide with which (classes_collide)
};

struct StepDiag {
    unsigned long long contacts;
    double residual;
};

// S: the slab the tick reads (state, constants, zones); So: the slab the new state is written to -- S itself for an
// in-place tick, or the batch's other slab (constants and zones are kept identical in both), which leaves S's state
// behind untouched: the zero-cost snapshot of a collision-proof chunk (dmx_general.cpp)
template <class T>
hipError_t launch_step(T *S, T *So, const uint8_t *gtype, int64_t stride, int64_t n, const StepParams<T> &P, bool ext,
                       StepDiag *diag, hipStream_t st);
template <class T>
hipError_t launch_islands(T *S, const uint8_t *bflags, int64_t stride, const IslandSet<T> &I, const StepParams<T> &P,
                          StepDiag *diag, hipStream_t st);
// dWorldStep: every island with rows is solved exactly (boxed LCP, block principal pivoting) by one workgroup;
// scratch = per island 2 m^2 + 3 m reals at scratch_off[k] (k = index in I.big_list), iscratch = 9 ints per reserved row
template <class T>
hipError_t launch_islands_exact(T *S, const uint8_t *bflags, int64_t stride, const IslandSet<T> &I, const StepParams<T> &P,
                                StepDiag *diag, T *scratch, const long long *scratch_off, int *iscratch, int max_rows, hipStream_t st);
// bucket counts and flags to zero, and two optional small records (each at most 256 words) with them
hipError_t launch_bp_clear(uint32_t *count, size_t n_count, uint32_t *flags, void *rec_a, size_t bytes_a, void *rec_b, size_t bytes_b,
                           hipStream_t st);
template <class T>
hipError_t launch_bp_insert(T *S, const uint8_t *gtype, int64_t stride, int64_t n, const GridParams<T> &G, hipStream_t st);
// the exact world AABB of every convex body among [0, n) into G.rec (one wavefront per hull), over the bounding sphere's box
// bp_insert left there
template <class T>
hipError_t launch_bp_convex_aabb(const T *S, const uint8_t *gtype, int64_t n, const GridParams<T> &G, hipStream_t st);
template <class T>
hipError_t launch_bp_safe_zone(T *S, const uint8_t *gtype, int64_t stride, int64_t n_active, const GridParams<T> &G, hipStream_t st);
// ground-plane contacts of the convex bodies among [0, n): one wavefront per body walks the hull (dCollideConvexPlane)
template <class T>
hipError_t launch_np_convex_plane(const T *S, const uint8_t *gtype, int64_t n, const StepParams<T> &P, hipStream_t st);
// contacts of bodies [0, n) with the ground plane and the static boxes -> P.sbuf / P.scount (boxes and spheres one lane each,
// convex hulls one wavefront each); bodies masked by P.skip are left alone
template <class T>
hipError_t launch_np_static(const T *S, const uint8_t *gtype, int64_t n, const StepParams<T> &P, hipStream_t st);
template <class T>
hipError_t launch_pack_transforms(const T *S, int64_t stride, int64_t first, int64_t count, T *out, hipStream_t st);
template <class T>
hipError_t launch_gather(const T *S, int64_t stride, const int32_t *idx, int64_t count, T *out, hipStream_t st);
template <class T>
hipError_t launch_scatter(T *S, int64_t stride, const int32_t *idx, int64_t count, const T *in, hipStream_t st);
// safe-zone test of slots [first, first+count) only (no integration): raises BPF_VIOLATION / BPF_WARN in `flags`
template <class T>
hipError_t launch_check_zones(const T *S, int64_t first, int64_t count, uint32_t *flags, hipStream_t st);
// ghost slots [first, first+count_lo) <- src_lo, the next count_hi <- src_hi (13-real AoS rows; null = leave alone), with
// an optional safe-zone test of the new positions
template <class T>
hipError_t launch_refresh_ghosts(T *S, int64_t first, int64_t count_lo, const T *src_lo, int64_t count_hi, const T *src_hi,
                                 int check, uint32_t *flags, hipStream_t st);
template <class T>
hipError_t launch_fill_component(T *S, int c, T value, int64_t n, hipStream_t st);
// rollback snapshot of the 13 state components of bodies [0, n_bodies): save = slab -> packed, else packed -> slab
template <class T>
hipError_t launch_copy_state(T *S, T *packed, int64_t n_bodies, bool save, hipStream_t st);
// components [c0, c0+k) of bodies [first, first+count) from one slab to the other (same layout)
template <class T>
hipError_t launch_copy_components(const T *from, T *to, int c0, int k, int64_t first, int64_t count, hipStream_t st);
template <class T>
hipError_t launch_aos_to_soa(T *S, int64_t stride, int comp0, int k, int64_t first, int64_t count, const T *aos,
                             hipStream_t st);
template <class T>
hipError_t launch_soa_to_aos(const T *S, int64_t stride, int comp0, int k, int64_t first, int64_t count, T *aos,
                             hipStream_t st);

// one per translation unit with device code: forces the unit's code object to load (see dmx_kernels.hip)
hipError_t dmx_touch_kernels(int real_bytes); hipError_t dmx_touch_islands(int real_bytes); hipError_t dmx_touch_broadphase(int real_bytes);
hipError_t dmx_touch_narrow(int real_bytes); hipError_t dmx_touch_exact(int real_bytes);

}  // namespace dmx
