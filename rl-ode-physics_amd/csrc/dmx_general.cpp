// dmx_general.cpp -- the batch tick with body-body collision handling (dSpaceCollide for body pairs,
// /root/reference/src/main.c:212, then dWorldStep over whatever islands result).
//
// Fast mode (the normal case for the BASELINE scenes): every body carries a broadphase safe zone -- half
// the horizontal gap to its nearest neighbour's bounding sphere at build time.  While every body is inside
// its zone no two bounding spheres can touch, so no collider could return a contact (AABB pairs may exist; they
// would all be empty) and the fused single-body-island kernels are exact; the 3-real check rides inside those
// kernels.  Ticks are enqueued in chunks with no host round trip; the violation flag is read once per chunk, and not
// before the chunk has reached its length or somebody looks (lazy chunks, below).
//
// Careful mode (a body left its zone, bodies are crowded or at a static box): the chunk is rolled back to its snapshot
// (a pointer swap between the batch's two slabs) and replayed tick by tick: the whole bookkeeping of an exact tick --
// pair search, islands, narrowphase, joints by island, level schedules -- runs on the device (dmx_exact.hip), the host
// reads one 64-byte record per tick, then the island kernels; every other body takes the fused kernel with those bodies
// masked out.  Both modes give the same bits as the sequential CPU restatement the tests compare with.
#include <string.h>
#include <algorithm>
#include <memory>
#include <chrono>
#include <atomic>
#include <cmath>

#include "dmx_batch_priv.hpp"
#include "dmx_exact.hpp"

namespace {

constexpr int kChunk = 32;          // ticks between violation-flag reads in fast mode: the starting length ...
constexpr int kChunkMax = 256;      // ... doubled after every clean chunk up to this, back to kChunk after a rollback
constexpr int kBucketCap = 8;
constexpr double kSkin = 1.25;      // cell = kSkin * largest bounding-sphere diameter

template <class T> GridParams<T> grid_of(dmxBatch *b)
{
    GridParams<T> G;
    G.r_max = (T)b->bp_rmax;
    for (int c = 0; c < 4; c++) G.r_cls[c] = (T)b->bp_rcls[c];
    G.cell = (T)(2.0 * kSkin * b->bp_rmax);
    G.inv_cell = T(1) / G.cell;
    G.mask = b->bp_mask;
    G.xbits = b->bp_xbits > 0 ? b->bp_xbits : 0;
    G.cap = b->bp_cap;
    G.count = (uint32_t *)b->bp_count.p;
    G.items = (int32_t *)b->bp_items.p;
    G.flags = (uint32_t *)b->bp_flags.p;
    G.rec = (GridRec<T> *)b->ex_aabb.p;     // null until an exact tick has asked for it
    G.sbox = (const T *)b->sbox.p; G.n_static = b->n_static;
    G.static_fast = b->static_fast ? 1 : 0; G.plane_on = b->plane_on;
    G.hull = (const T *)b->hull.p; G.hull_n = b->hull_n;
    G.class_pairs = b->class_pairs;
    return G;
}

// Pinned, device-visible host records (the chunk flags, the exact tick's count record).  Zeroed: a recycled pinned page may
// hold a previous owner's record, and await_host_record trusts the sequence number it finds there.
int alloc_host_record(void **p, size_t bytes)
{
    HIP_TRY(hipHostMalloc(p, bytes, hipHostMallocCoherent | hipHostMallocMapped));
    memset(*p, 0, bytes);
    return DMX_OK;
}
// Sequence numbers of the small-scene kernels' host records: process-wide and never 0, so no record left in memory by an
// earlier batch -- or by this batch's zeroed allocation -- can be taken for the one a launch is about to write.
uint32_t next_record_seq()
{
    static std::atomic<uint32_t> g{ 0 };
    uint32_t s;
    do { s = g.fetch_add(1u, std::memory_order_relaxed) + 1u; } while (s == 0u);
    return s;
}

int read_flags(dmxBatch *b)
{
    HIP_TRY(hipMemcpyAsync(b->bp_flags_host, b->bp_flags.p, BPF_COUNT * sizeof(uint32_t), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return DMX_OK;
}

int ensure_buffers(dmxBatch *b)
{
    if (b->bp_rmax <= 0) {
        double r = 0, rcls[4] = { 0, 0, 0, 0 };
        for (int64_t i = 0; i < b->n; i++) {
            const double *s = &b->h_sides[(size_t)3 * i];
            const double ri = (b->h_gtype[(size_t)i] == GEOM_SPHERE || b->h_gtype[(size_t)i] == GEOM_CONVEX) ? s[0]     // convex: the hull's bounding radius
                            : b->h_gtype[(size_t)i] == GEOM_BOX ? 0.5 * std::sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]) : 0.0;
            r = std::max(r, ri);
            if (b->h_gtype[(size_t)i] < 4) rcls[b->h_gtype[(size_t)i]] = std::max(rcls[b->h_gtype[(size_t)i]], ri);
        }
        b->bp_rmax = r > 0 ? r : 1.0;
        for (int c = 0; c < 4; c++) b->bp_rcls[c] = rcls[c];
    }
    if (!b->bp_mask) {
        uint32_t h = 1024;
        while ((int64_t)h < 2 * b->n) h <<= 1;
        b->bp_mask = h - 1;
        b->bp_cap = kBucketCap;
        int bits = 0;
        while ((1u << bits) < h) bits++;
        b->bp_xbits = (bits + 1) / 2;          // a square torus to start with; see grow_buckets
    }
    int rc;
    const size_t tbl = (size_t)b->bp_mask + 1;
    if ((rc = dmx_ensure_dev(b->bp_count, tbl * sizeof(uint32_t))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->bp_items, tbl * (size_t)b->bp_cap * sizeof(int32_t))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->bp_flags, 64)) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->bp_inpair, (size_t)b->stride)) != DMX_OK) return rc;
    if (!b->bp_flags_host && (rc = alloc_host_record((void **)&b->bp_flags_host, 64)) != DMX_OK) return rc;
    return DMX_OK;
}

// stacked bodies share an (x,z) column: double the bucket capacity when one overflows
int grow_buckets(dmxBatch *b)
{
    if (b->bp_xbits > 0) {                      // the torus wraps this scene onto itself too often: scramble instead
        b->bp_xbits = 0;
        return DMX_OK;
    }
    if (b->bp_cap >= 1024) {
        fprintf(stderr, "libode_mi355: broadphase bucket overflow (> %d bodies in one (x,z) column)\n", b->bp_cap);
        return DMX_ECAPACITY;
    }
    b->bp_cap *= 2;
    return dmx_ensure_dev(b->bp_items, ((size_t)b->bp_mask + 1) * (size_t)b->bp_cap * sizeof(int32_t));
}

// (rec_a / rec_b: small records of the caller's to zero in the same launch)
template <class T> int fill_grid(dmxBatch *b, void *rec_a = nullptr, size_t bytes_a = 0, void *rec_b = nullptr, size_t bytes_b = 0)
{
    const GridParams<T> G = grid_of<T>(b);
    HIP_TRY(launch_bp_clear((uint32_t *)b->bp_count.p, (size_t)b->bp_mask + 1, (uint32_t *)b->bp_flags.p, rec_a, bytes_a, rec_b, bytes_b,
                            b->stream));
    HIP_TRY(launch_bp_insert<T>((T *)b->slab, b->gtype, b->stride, b->n, G, b->stream));   // ghosts included
    if (G.rec != nullptr && b->hull_n > 0)      // the pair search is coming: hulls get their exact AABBs (the zones keep bounding spheres)
        HIP_TRY(launch_bp_convex_aabb<T>((const T *)b->slab, b->gtype, b->n, G, b->stream));
    return DMX_OK;
}

// (re)build every body's safe zone from the current poses
template <class T> int build_safe_zones(dmxBatch *b)
{
    DmxPhase pz(b, 2);
    int rc = ensure_buffers(b);
    if (rc != DMX_OK) return rc;
    if ((rc = fill_grid<T>(b)) != DMX_OK) return rc;
    // ghost slots [n_active, n) get zones too: the caller that refreshes them checks them (dmxBatchCheckZonesOnStream)
    HIP_TRY(launch_bp_safe_zone<T>((T *)b->slab, b->gtype, b->stride, b->n, grid_of<T>(b), b->stream));
    if ((rc = read_flags(b)) != DMX_OK) return rc;
    if (b->bp_flags_host[BPF_OVERFLOW]) {
        if ((rc = grow_buckets(b)) != DMX_OK) return rc;
        return build_safe_zones<T>(b);
    }
    b->bp_crowded = b->bp_flags_host[BPF_CROWDED];
    // zones (and the bounding radii bp_insert refreshed) are constants of the ticks to come: both slabs hold them
    HIP_TRY(launch_copy_components<T>((const T *)b->slab, (T *)b->slab_alt, C_BPX, C_COUNT - C_BPX, 0, b->n, b->stream));
    b->bp_valid = true;
    b->bp_fresh = true;
    b->stat_rebuilds++;
    return DMX_OK;
}

// ---- the chunk's rollback snapshot ------------------------------------------------------------------------
// Ping-pong (default): nothing is copied.  The chunk's first fast launch reads the current slab and writes the new
// state to the other one, the slabs swap roles, later launches run in place; the start state stays behind untouched
// and a rollback swaps back.  Copy mode (callers that replay captured HIP graphs, which bake the slab address in):
// the 13 state components are copied aside at the chunk's start.
enum : int { SNAP_NONE = 0, SNAP_PINGPONG = 1, SNAP_COPY = 2 };

template <class T> int snapshot_by_copy(dmxBatch *b)
{
    int rc;
    if ((rc = dmx_ensure_dev(b->bp_snapshot, (size_t)C_MASS * b->stride * b->rsize)) != DMX_OK) return rc;
    HIP_TRY(launch_copy_state<T>((T *)b->slab, (T *)b->bp_snapshot.p, b->stride, true, b->stream));
    b->flip_armed = false; b->flipped = false;
    b->snap_kind = SNAP_COPY;
    return DMX_OK;
}

template <class T> int snapshot_begin(dmxBatch *b)
{
    if (b->snapshot_mode == DMX_SNAPSHOT_COPY) return snapshot_by_copy<T>(b);
    b->flip_armed = true; b->flipped = false;
    b->snap_kind = SNAP_NONE;                   // becomes SNAP_PINGPONG at the first fast launch
    b->snap_fresh = b->bp_fresh;
    return DMX_OK;
}

template <class T> int snapshot_restore(dmxBatch *b)
{
    if (b->snap_kind == SNAP_COPY)
        HIP_TRY(launch_copy_state<T>((T *)b->slab, (T *)b->bp_snapshot.p, b->stride, false, b->stream));
    else if (b->snap_kind == SNAP_PINGPONG)
        std::swap(b->slab, b->slab_alt);        // the untouched start state (ghost slots included) is current again
    b->flip_armed = false; b->flipped = false;
    b->snap_kind = SNAP_NONE;
    b->bp_fresh = b->snap_fresh;
    return DMX_OK;
}

inline void snapshot_drop(dmxBatch *b) { b->flip_armed = false; b->flipped = false; b->snap_kind = SNAP_NONE; }

// one launch of the fused kernels over the active bodies; starts the chunk's ping-pong when one is armed
template <class T> int launch_fast(dmxBatch *b, const StepParams<T> &P, bool ext)
{
    T *S = (T *)b->slab, *So = S;
    if (b->flip_armed) {
        if (ext || P.skip != nullptr) {
            // this launch clears accumulators / skips bodies in its input slab: it cannot leave that slab behind
            // as the snapshot.  Take the snapshot by copy for this chunk.
            int rc = snapshot_by_copy<T>(b);
            if (rc != DMX_OK) return rc;
        } else {
            So = (T *)b->slab_alt;
        }
    }
    HIP_TRY(launch_step<T>(S, So, b->gtype, b->stride, b->n_active, P, ext, b->diag, b->stream));
    b->bp_fresh = false;                     // the poses have moved on from the ones the zones were built at
    if (So != S) {
        // ghost slots [n_active, n) are not stepped: their state follows by copy (a few boundary rows)
        HIP_TRY(launch_copy_components<T>(S, So, 0, C_MASS, b->n_active, b->n - b->n_active, b->stream));
        std::swap(b->slab, b->slab_alt);
        b->flip_armed = false; b->flipped = true;
        b->snap_kind = SNAP_PINGPONG;
    }
    return DMX_OK;
}

template <class T> int fused_tick(dmxBatch *b, double h, bool check, const uint8_t *skip, const uint32_t *gate = nullptr)
{
    StepParams<T> P = dmx_make_params<T>(b, h);
    P.bp_check = check ? BPC_ALL : 0;
    P.bp_flags = (uint32_t *)b->bp_flags.p;
    P.skip = skip;
    P.gate = gate;
    // static fused path: the launch for bodies with 5..8 contacts rides along once such a body has been met; a checked tick
    // that meets one without it says so (BPF_NEED8) and its chunk is run again, an unchecked tick cannot be
    P.have8 = (b->static_need8 || !check) ? 1 : 0;
    int rc = launch_fast<T>(b, P, b->ext_pending);
    if (rc != DMX_OK) return rc;
    b->ext_pending = false;
    return DMX_OK;
}

// n ticks of the fast path.  ends_only: the safe-zone test is needed at the run's first (check_first) and last
// (check_last) tick only -- ballistic chunks; otherwise at every tick.  Contact-free scenes take them
// ticks_per_launch at a time inside one integrate_free launch (state in registers between ticks).
template <class T> int fused_run(dmxBatch *b, double h, int n, bool ends_only, bool check_first, bool check_last)
{
    int rc;
    const int per = (b->plane_on || b->n_static > 0 || b->ext_pending) ? 1 : std::max(1, b->ticks_per_launch);
    for (int s = 0; s < n; s += per) {
        const int kk = std::min(per, n - s);
        if (kk == 1) {
            const bool check = !ends_only || (s == 0 && check_first) || (s == n - 1 && check_last);
            if ((rc = fused_tick<T>(b, h, check, nullptr)) != DMX_OK) return rc;
            continue;
        }
        StepParams<T> P = dmx_make_params<T>(b, h);
        P.ticks = kk;
        P.bp_check = !ends_only ? BPC_ALL : ((s == 0 && check_first ? BPC_FIRST : 0) | (s + kk == n && check_last ? BPC_LAST : 0));
        P.bp_flags = (uint32_t *)b->bp_flags.p;
        if ((rc = launch_fast<T>(b, P, false)) != DMX_OK) return rc;
    }
    return DMX_OK;
}

// ---- one exact tick ---------------------------------------------------------------------------------------------
// The whole bookkeeping runs on the device (dmx_exact.hip): canonical pair list, involved bodies, islands, narrowphase,
// joints grouped by island in creation order, level schedules.  The host enqueues the pipeline with capacities
// estimated from earlier ticks, reads ONE 64-byte record back (counts + overflow flags; the stream synchronisation of
// the tick), then launches the island solve with the exact shape and the fused kernel for everyone else.
constexpr int kBigIslandRows = 1;        // multi-body islands with at least this many rows get a workgroup (see dmx_joints.cpp)

constexpr int64_t kSmallExactBodies = 2048, kSmallExactPairs = 512;
// Does this exact tick run its bookkeeping as the two one-workgroup kernels (dmx_exact.hip)?  dmxBatchSetExactPipeline; AUTO:
// scenes of up to kSmallExactBodies slots whose last tick had at most kSmallExactPairs body pairs (one workgroup has one
// compute unit's memory pipeline: past that the stage-per-launch pipeline over many compute units is the faster one again).
int default_exact_pipeline()
{
    static const int v = [] {
        const char *e = getenv("DMX_SMALL_EXACT");
        return !e ? DMX_EXACT_AUTO : atoi(e) == 0 ? DMX_EXACT_STAGED : atoi(e) == 2 ? DMX_EXACT_ONE_WORKGROUP : DMX_EXACT_AUTO;
    }();
    return v;
}
bool use_small_exact(const dmxBatch *b, const ExactCaps &cap, bool pairs_matter)
{
    const int mode = b->exact_pipeline != DMX_EXACT_AUTO ? b->exact_pipeline : default_exact_pipeline();
    if (mode == DMX_EXACT_STAGED || !exact_small_fits(b->n, b->bp_mask, cap)) return false;
    if (mode == DMX_EXACT_ONE_WORKGROUP) return true;
    return b->n <= kSmallExactBodies && (!pairs_matter || b->last_pairs <= (unsigned long long)kSmallExactPairs);
}

// (DMX_EXACT_STAGED forces the stage-per-launch form throughout; DMX_HYBRID_EXACT=0 switches this form off for A/B runs)
bool use_hybrid_exact(const dmxBatch *b, const ExactCaps &cap)
{
    static const bool on = [] { const char *e = getenv("DMX_HYBRID_EXACT"); return !(e && atoi(e) == 0); }();
    const int mode = b->exact_pipeline != DMX_EXACT_AUTO ? b->exact_pipeline : default_exact_pipeline();
    // (up to 4 096 pairs while the entries fit the one-workgroup kernel: the reference's pen with 400 / 512 bodies is 1 250 / 1 700 pairs,
    //  and its nine small rocPRIM launches and their neighbours cost more than that kernel: 388 -> 362 / 428 -> 404 us per tick)
    static const long long maxp = [] { const char *e = getenv("DMX_HYBRID_PAIRS"); return e ? atoll(e) : 4096ll; }();
    return on && mode != DMX_EXACT_STAGED && exact_back_fits(cap) && (long long)b->last_pairs <= maxp;
}

bool exs_timing_enabled()
{
    static const bool v = [] { const char *e = getenv("DMX_EXS_TIMING"); return e && atoi(e) != 0; }();
    return v;
}

// Wait for the small-scene kernels' host record: watch for the sequence number (the record's last word, written after a
// system-scope fence) rather than synchronise the stream -- the device keeps running, the host has its numbers a
// microsecond or two after they were written.  Falls back to the stream after 2 ms (a faulted kernel never writes).
// NOTE for callers: on return the record is complete but the STREAM MAY STILL BE BUSY (the kernel that wrote the record has
// not necessarily retired); whatever follows must be stream-ordered behind it or synchronise itself.
int await_host_record(dmxBatch *b, uint32_t seq)
{
    static const bool spin = [] { const char *e = getenv("DMX_RECORD_SPIN"); return !(e && atoi(e) == 0); }();
    volatile uint32_t *word = &((volatile ExactCounts *)b->ex_counts_host)->seq;
    if (spin) {
        const auto t0 = std::chrono::steady_clock::now();
        for (int it = 0;; it++) {
            if (*word == seq) { std::atomic_thread_fence(std::memory_order_acquire); return DMX_OK; }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#else
            std::this_thread::yield();
#endif
            if ((it & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
        }
    }
    HIP_TRY(hipStreamSynchronize(b->stream));
    return *word == seq ? DMX_OK : DMX_EHIP;
}

// device-visible addresses of the pinned host records the small-scene kernels write themselves
int host_record_pointers(dmxBatch *b, ExactCounts **counts_dev, uint32_t **flags_dev)
{
    if (!b->ex_counts_dev) {
        HIP_TRY(hipHostGetDevicePointer(&b->ex_counts_dev, b->ex_counts_host, 0));
        HIP_TRY(hipHostGetDevicePointer(&b->bp_flags_dev, b->bp_flags_host, 0));
    }
    *counts_dev = (ExactCounts *)b->ex_counts_dev;
    *flags_dev = (uint32_t *)b->bp_flags_dev;
    return DMX_OK;
}

int big_island_rows_general()
{
    static const int v = [] { const char *e = getenv("DMX_BIG_ISLAND_ROWS"); return e ? atoi(e) : kBigIslandRows; }();
    return v;
}

template <class T> int ensure_exact_buffers(dmxBatch *b, const ExactCaps &cap, ExactBuffers<T> &B)
{
    int rc;
    const size_t ne = (size_t)cap.entries(), nslots = cap.slots();
    const size_t n = (size_t)b->stride;
    if ((rc = dmx_ensure_dev(b->ex_body, n * (2 * sizeof(uint64_t)))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->ex_aabb, n * sizeof(GridRec<T>))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->bp_inpair, n)) != DMX_OK) return rc;
    if (!b->ex_last.p) {
        if ((rc = dmx_ensure_dev(b->ex_last, n * sizeof(int32_t))) != DMX_OK) return rc;
        HIP_TRY(exact_init_last((int32_t *)b->ex_last.p, (int64_t)(b->ex_last.bytes / sizeof(int32_t)), b->stream));
    }
    const size_t temp = exact_temp_bytes(cap, b->n_active);
    // one arena, carved up in 256-byte steps
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += (bytes + 255) / 256 * 256; return at; };
    const size_t o_counts = take(sizeof(ExactCounts)), o_stamps = take(64 * sizeof(uint64_t)), o_cross = take(2 * EX_CROSS_CAP * sizeof(int32_t)), o_temp = take(temp), o_pairs = take(2 * (size_t)cap.pairs * 4),
                 o_inv = take((size_t)cap.inv * 4), o_parent = take((size_t)cap.inv * 4), o_root = take((size_t)cap.inv * 4),
                 o_rf = take((size_t)cap.inv * 4), o_rinc = take((size_t)cap.inv * 4),
                 o_gpos = take(nslots * 3 * sizeof(T)), o_gnormal = take(nslots * 3 * sizeof(T)), o_gdepth = take(nslots * sizeof(T)),
                 o_cc = take(ne * 4), o_keys = take(ne * 4), o_vals = take(ne * 4), o_keys_s = take(ne * 4), o_vals_s = take(ne * 4),
                 o_sc = take(ne * 8), o_sinc = take(ne * 8),
                 o_body_off = take(((size_t)cap.inv + 1) * 4), o_con_off = take(((size_t)cap.inv + 1) * 4), o_row_off = take(((size_t)cap.inv + 1) * 4),
                 o_bodies = take((size_t)cap.inv * 4),
                 o_cb1 = take(nslots * 4), o_cb2 = take(nslots * 4), o_csrc = take(nslots * 4), o_crow = take(nslots * 4),
                 o_bg = take((size_t)cap.inv * 8), o_binc = take((size_t)cap.inv * 8),
                 o_big = take((size_t)cap.inv * 4), o_big_list = take((size_t)cap.inv * 4), o_lev_count = take((size_t)cap.inv * 4),
                 o_lev_off = take(((size_t)cap.rows + cap.inv + 1) * 4), o_lev_rows = take((size_t)cap.rows * 4), o_row_level = take((size_t)cap.rows * 4);
    // partners found by the pair search's count pass, EX_STAGE_PARTNERS per body, so that the write pass need not walk the
    // grid again (scenes of up to 65 536 slots: where the walk is the tick -- a crowded pen -- not memory for a million bodies)
    const bool stage = n <= 65536;
    const size_t o_stage = stage ? take(n * EX_STAGE_PARTNERS * sizeof(int32_t)) : 0;
    if (off > ((size_t)128 << 30)) {         // (involved bodies x (1 + static boxes) entries, 8 contact slots each: say so rather than fail in hipMalloc)
        fprintf(stderr, "libode_mi355: the exact tick would need %.1f GB of work arrays (%u involved bodies x %u static boxes)\n",
                (double)off / 1e9, cap.inv, cap.nstatic);
        return DMX_ECAPACITY;
    }
    if ((rc = dmx_ensure_dev(b->ex_arena, off)) != DMX_OK) return rc;
    char *A = (char *)b->ex_arena.p;
    B.counts = (ExactCounts *)(A + o_counts);
    B.cross_list = (int32_t *)(A + o_cross);
    B.stamps = (uint64_t *)(A + o_stamps);
    B.temp = A + o_temp; B.temp_bytes = temp;
    B.pc = (uint64_t *)b->ex_body.p; B.inc = B.pc + n;
    B.inpair = (uint8_t *)b->bp_inpair.p;
    B.pairs = (int32_t *)(A + o_pairs);
    B.inv = (int32_t *)(A + o_inv); B.parent = (int32_t *)(A + o_parent); B.root = (int32_t *)(A + o_root);
    B.rf = (uint32_t *)(A + o_rf); B.rinc = (uint32_t *)(A + o_rinc);
    B.gpos = (T *)(A + o_gpos); B.gnormal = (T *)(A + o_gnormal); B.gdepth = (T *)(A + o_gdepth);
    B.cc = (uint32_t *)(A + o_cc);
    B.keys = (uint32_t *)(A + o_keys); B.vals = (uint32_t *)(A + o_vals); B.keys_s = (uint32_t *)(A + o_keys_s); B.vals_s = (uint32_t *)(A + o_vals_s);
    B.sc = (uint64_t *)(A + o_sc); B.sinc = (uint64_t *)(A + o_sinc);
    B.body_off = (int *)(A + o_body_off); B.con_off = (int *)(A + o_con_off); B.row_off = (int *)(A + o_row_off);
    B.bodies = (int *)(A + o_bodies);
    B.cb1 = (int *)(A + o_cb1); B.cb2 = (int *)(A + o_cb2); B.csrc = (int *)(A + o_csrc); B.crow = (int *)(A + o_crow);
    B.bg = (uint64_t *)(A + o_bg); B.binc = (uint64_t *)(A + o_binc);
    B.big = (int *)(A + o_big); B.big_list = (int *)(A + o_big_list); B.lev_count = (int *)(A + o_lev_count);
    B.lev_off = (int *)(A + o_lev_off); B.lev_rows = (int *)(A + o_lev_rows); B.row_level = (int *)(A + o_row_level);
    B.stage = stage ? (int32_t *)(A + o_stage) : nullptr;
    B.last = (int32_t *)b->ex_last.p;
    return DMX_OK;
}

// the device pipeline's output as the island kernels' input; C: the tick's counts (null: not known yet -- a speculative launch,
// whose workgroups read them from the device's record)
template <class T> IslandSet<T> island_set_of(dmxBatch *b, const ExactBuffers<T> &B, const ExactCounts *C)
{
    IslandSet<T> I;
    memset(&I, 0, sizeof(I));
    I.body_off = B.body_off; I.bodies = B.bodies; I.con_off = B.con_off; I.row_off = B.row_off;
    I.cb1 = B.cb1; I.cb2 = B.cb2; I.csrc = B.csrc; I.crow = B.crow;
    I.gpos = B.gpos; I.gnormal = B.gnormal; I.gdepth = B.gdepth;
    I.big = B.big; I.big_list = B.big_list; I.lev_count = B.lev_count;
    I.lev_off = B.lev_off; I.lev_rows = B.lev_rows; I.row_level = B.row_level;
    if (C != nullptr) {
        I.n_islands = (int)C->ni; I.n_big = (int)C->nbig;
        I.big_max_bodies = (int)C->big_max_bodies; I.big_max_width = (int)C->big_max_width;
        I.big_rows_total = (int)C->big_rows; I.big_max_rows = (int)C->big_max_rows;
    }
    I.rows = (T *)b->jd_rows.p; I.rowjb = (int *)b->jd_rowjb.p; I.bscr = (T *)b->jd_bscr.p; I.local = (int *)b->jd_local.p;
    I.singles = 1;
    // (cmode / cmu / cbounce ... stay null: every contact carries the batch's surface, NearCallback's policy, main.c:684-687)
    return I;
}

// DMX_SPECULATE=0: the small-scene exact tick waits for its record before it launches the solve (A/B runs)
bool speculate_small_exact()
{
    static const bool on = [] { const char *e = getenv("DMX_SPECULATE"); return !(e && atoi(e) == 0); }();
    return on;
}

// DMX_FORK_FUSED=0: an exact tick's fused step runs behind the island solve on the batch's stream, not beside it on a second one
bool fork_fused_enabled()
{
    static const bool on = [] { const char *e = getenv("DMX_FORK_FUSED"); return !(e && atoi(e) == 0); }();
    return on;
}

// DMX_SPEC_FUSE=0: the speculative tick's fused step is a launch of its own behind the solve (A/B runs)
bool fuse_spec_tail()
{
    static const bool on = [] { const char *e = getenv("DMX_SPEC_FUSE"); return !(e && atoi(e) == 0); }();
    return on;
}

template <class T> int careful_tick(dmxBatch *b, double h)
{
    int rc;
    snapshot_drop(b);           // exact ticks run in place and are never rolled back
    std::unique_ptr<DmxPhase> ph(new DmxPhase(b, 0));
    if (!b->ex_counts_host && (rc = alloc_host_record(&b->ex_counts_host, sizeof(ExactCounts))) != DMX_OK) return rc;
    const StepParams<T> P = dmx_make_params<T>(b, h);
    const int rpc = b->mu > 0 ? 3 : 1;
    ExactBuffers<T> B;
    ExactCounts &C = *(ExactCounts *)b->ex_counts_host;
    if (b->ex_cap_pairs == 0) { b->ex_cap_pairs = 1024; b->ex_cap_rows = 16384; }
    if (!b->bp_flags_host && (rc = alloc_host_record((void **)&b->bp_flags_host, 64)) != DMX_OK) return rc;
    bool small = false, speculated = false;
    for (int attempt = 0;; attempt++) {
        if (attempt > 40) return DMX_ECAPACITY;
        ExactCaps cap;
        cap.pairs = b->ex_cap_pairs;
        cap.inv = (uint32_t)std::min<int64_t>(2 * (int64_t)cap.pairs, b->n_active);
        cap.rows = b->ex_cap_rows;
        cap.nstatic = (uint32_t)b->n_static;
        if ((rc = ensure_exact_buffers<T>(b, cap, B)) != DMX_OK) return rc;
        small = use_small_exact(b, cap, true);
        speculated = false;
        if (small) {
            // everything between here and the island solve in three launches; the last one leaves the counts and the
            // flags in host memory: the tick's one wait is all the host does
            ExactCounts *hc; uint32_t *hf;
            if ((rc = host_record_pointers(b, &hc, &hf)) != DMX_OK) return rc;
            const GridParams<T> G = grid_of<T>(b);
            const uint32_t seq = next_record_seq();
            HIP_TRY(launch_exact_small_front<T>((T *)b->slab, b->gtype, b->n, b->n_active, G, B, cap, nullptr, nullptr, 0u, b->stream));
            HIP_TRY(launch_exact_small_group<T>((const T *)b->slab, b->gtype, G, P, B, cap, rpc, big_island_rows_general(), b->diag_isl,
                                                hc, hf, seq, b->n, b->stream));
            if (speculate_small_exact() && !b->spec_refused && !b->ext_pending && (size_t)3 * cap.slots() * ISLAND_ROW_REALS * sizeof(T) <= ((size_t)64 << 20)) {
                // The rest of the tick goes out BEHIND those, before the host has seen a count: the island solve over the capacity
                // (workgroups ask the device's record whether they exist) and the fused step for everyone else, both gated on
                // the record's spec_ok -- the last kernel above clears it when anything overflowed, an island spans two ranks,
                // or an island is not solve_island_wg<64>'s kind; then neither does anything and the host, which still reads
                // the record below, launches what the counts call for as it always did.  In the common case the device goes
                // from the bookkeeping straight into the solve (10 us of host round trip per tick gone), and the host is back
                // enqueueing the next tick while it runs.  Scratch sized for the capacity, not the counts.
                const size_t max_rows = (size_t)3 * cap.slots();
                if ((rc = dmx_ensure_dev(b->jd_rows, (max_rows + 1) * ISLAND_ROW_REALS * sizeof(T))) != DMX_OK) return rc;
                if ((rc = dmx_ensure_dev(b->jd_rowjb, (max_rows + 1) * 2 * sizeof(int))) != DMX_OK) return rc;
                if ((rc = dmx_ensure_dev(b->jd_bscr, ((size_t)cap.inv + 1) * 28 * sizeof(T))) != DMX_OK) return rc;
                if ((rc = dmx_ensure_dev(b->jd_local, (size_t)b->stride * sizeof(int))) != DMX_OK) return rc;
                const IslandSet<T> I = island_set_of<T>(b, B, nullptr);
                if (b->plane_on && b->n_static == 0 && b->hull_n == 0 && fuse_spec_tail()) {
                    // boxes and spheres on the ground plane: the fused step for everyone else rides in the solve's launch (the two
                    // touch disjoint bodies): what fused_tick would set up, with the skip mask and the gate
                    StepParams<T> Pf = dmx_make_params<T>(b, h);
                    Pf.bp_check = 0;
                    Pf.bp_flags = (uint32_t *)b->bp_flags.p;
                    Pf.skip = (const uint8_t *)b->bp_inpair.p;
                    Pf.gate = &B.counts->spec_ok;
                    HIP_TRY(launch_islands_and_step_speculative<T>((T *)b->slab, b->bflags, b->gtype, b->stride, b->n_active, I, P, Pf, b->diag_isl,
                                                                   b->diag, B.counts, cap.inv, b->stream));
                    b->bp_fresh = false;
                } else {
                    HIP_TRY(launch_islands_speculative<T>((T *)b->slab, b->bflags, b->stride, I, P, b->diag_isl, B.counts, cap.inv, b->stream));
                    if ((rc = fused_tick<T>(b, h, false, (const uint8_t *)b->bp_inpair.p, &B.counts->spec_ok)) != DMX_OK) return rc;
                }
                speculated = true;
            }
            { DmxPhase pw(b, 1); if ((rc = await_host_record(b, seq)) != DMX_OK) return rc; }
            if (exs_timing_enabled()) {
                uint64_t st[64];
                // (on the batch's own stream, and drained: the record's arrival does not mean the stream's later kernels have
                //  written their stamps, and the null stream is not ordered with a non-blocking stream)
                HIP_TRY(hipMemcpyAsync(st, B.stamps, sizeof(st), hipMemcpyDeviceToHost, b->stream));
                HIP_TRY(hipStreamSynchronize(b->stream));
                for (int k = 1; k < 9; k++) { b->exs_acc[k] += (double)(st[k] - st[k - 1]); b->exs_acc[32 + k] += (double)(st[32 + k] - st[32 + k - 1]); }
                b->exs_ticks++;
            }
        } else if (use_hybrid_exact(b, cap)) {
            // many bodies, few involved (a handful of teapots leaning on one another among thousands at rest): the body-sized stages
            // over the chip, the entry-sized ones in the one-workgroup kernel; one wait, on the record that kernel writes
            ExactCounts *hc; uint32_t *hf;
            if ((rc = host_record_pointers(b, &hc, &hf)) != DMX_OK) return rc;
            const uint32_t seq = next_record_seq();
            if ((rc = fill_grid<T>(b, B.counts, sizeof(ExactCounts), b->diag_isl, sizeof(StepDiag))) != DMX_OK) return rc;
            const GridParams<T> G = grid_of<T>(b);
            HIP_TRY(launch_exact_pairs<T>((const T *)b->slab, b->gtype, b->n_active, G, B, cap, b->stream));
            HIP_TRY(launch_exact_roots<T>(B, cap, b->stream));
            HIP_TRY(launch_exact_small_group<T>((const T *)b->slab, b->gtype, G, P, B, cap, rpc, big_island_rows_general(), b->diag_isl,
                                                hc, hf, seq, b->n, b->stream));
            if ((rc = await_host_record(b, seq)) != DMX_OK) return rc;
        } else {
        // (the count record and the island solve's diagnostics are zeroed with the grid: one launch)
        if ((rc = fill_grid<T>(b, B.counts, sizeof(ExactCounts), b->diag_isl, sizeof(StepDiag))) != DMX_OK) return rc;
        HIP_TRY(launch_exact_pairs<T>((const T *)b->slab, b->gtype, b->n_active, grid_of<T>(b), B, cap, b->stream));
        auto read_back = [&]() -> int {          // (the record carries the grid's overflow flag too: one copy)
            HIP_TRY(hipMemcpyAsync(&C, B.counts, sizeof(ExactCounts), hipMemcpyDeviceToHost, b->stream));
            HIP_TRY(hipStreamSynchronize(b->stream));
            b->bp_flags_host[BPF_OVERFLOW] = C.bp_overflow;
            return DMX_OK;
        };
        // The last exact tick found no body involved (crowded bounding spheres, nothing touching): most likely this one
        // will not either, so look at the counts before enqueueing the rest.  Otherwise the tick waits for the device once.
        bool looked = false;
        if (b->ex_prev_inv == 0) {
            if ((rc = read_back()) != DMX_OK) return rc;
            looked = true;
        }
        if (!looked || (!b->bp_flags_host[BPF_OVERFLOW] && !(C.overflow & 1u) && C.ninv > 0 && !C.cross)) {
            HIP_TRY(launch_exact_group<T>((const T *)b->slab, b->gtype, b->n_active, grid_of<T>(b), P, B, cap, rpc,
                                          big_island_rows_general(), b->stream));
            // the record travels by itself (one small launch writes it to pinned memory) and the host watches for it
            ExactCounts *hc; uint32_t *hf;
            if ((rc = host_record_pointers(b, &hc, &hf)) != DMX_OK) return rc;
            const uint32_t seq = next_record_seq();
            HIP_TRY(launch_exact_publish(B.counts, (const uint32_t *)b->bp_flags.p, hc, hf, seq, b->stream));
            { DmxPhase pw(b, 1); if ((rc = await_host_record(b, seq)) != DMX_OK) return rc; }
        }
        }
        if (b->bp_flags_host[BPF_OVERFLOW]) {                  // a column holds more bodies than a bucket: widen and search again
            if ((rc = grow_buckets(b)) != DMX_OK) return rc;
            continue;
        }
        if (C.overflow & 1u) {
            const uint64_t need = std::max<uint64_t>(C.npairs, (uint64_t)(C.ninv + 1) / 2);
            b->ex_cap_pairs = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2ull * cap.pairs, need + need / 4 + 64), 1ull << 28);
            if (need > (1ull << 28)) { fprintf(stderr, "libode_mi355: %llu body pairs exceed the pair capacity\n", (unsigned long long)need); return DMX_ECAPACITY; }
            continue;
        }
        if (C.overflow & 2u) {
            b->ex_cap_rows = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2ull * cap.rows, (uint64_t)C.big_rows + C.big_rows / 4 + 64), 1ull << 30);
            continue;
        }
        break;
    }
    b->last_pairs = C.npairs;
    b->ex_prev_inv = C.ninv;
    b->stat_careful_ticks++;
    if (C.cross) {
        fprintf(stderr, "libode_mi355: bodies %u and %u touch across two ranks' slabs; an island spanning ranks has to be "
                        "migrated to one owner first\n", C.cross_a, C.cross_b);
        return DMX_ECROSS;
    }
    // the speculative launches went ahead (the record says so): the tick is on the device in full, nothing is left to launch
    const bool done = speculated && C.spec_ok != 0u;
    if (done) b->stat_spec_ticks++;
    // the device's verdict on THIS tick is the forecast for the next one (a pile at the pen's walls stays the kind of island it is):
    // launches that would be refused are not enqueued (they cost the narrowphase of the fused path and three dispatches)
    b->spec_refused = C.spec_ok == 0u;
    if (C.ninv == 0) {                      // nobody in a body pair, nobody at a static box
        b->last_mixed = false;
        return done ? DMX_OK : fused_tick<T>(b, h, false, nullptr);
    }
    if (C.npairs > 0) b->stat_pair_ticks++;
    // a quiet scene that has turned busy: leave head room so the next ticks do not run the pipeline twice
    if (2ull * C.npairs > b->ex_cap_pairs) b->ex_cap_pairs = (uint32_t)std::min<uint64_t>(2ull * b->ex_cap_pairs, 1ull << 28);
    if (done) {
        b->last_islands = false;
        b->last_mixed = true;
        b->stepped_with_plane = true;
        return DMX_OK;
    }

    // ---- islands of the bodies in pairs; everyone else through the fused kernel ------------------------------------
    ph.reset(new DmxPhase(b, 7));
    const size_t nrows = (size_t)3 * C.njoints;
    if ((rc = dmx_ensure_dev(b->jd_rows, (nrows + 1) * ISLAND_ROW_REALS * sizeof(T))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->jd_rowjb, (nrows + 1) * 2 * sizeof(int))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->jd_bscr, ((size_t)C.ninv + 1) * 28 * sizeof(T))) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->jd_local, (size_t)b->stride * sizeof(int))) != DMX_OK) return rc;
    const IslandSet<T> I = island_set_of<T>(b, B, &C);
    // (diag_isl was zeroed with the grid / by the small-scene kernel)
    // The fused step for everyone else touches none of the islands' bodies: it goes to a second stream, forked here (the record has
    // been read: nothing can call the tick off any more) and joined behind the solve, so the two run side by side -- in the pen the
    // solve is a few workgroups for 200 us, in a field of hulls the fused step is the longer of the two.  (DMX_FORK_FUSED=0: one
    // after the other.)  The island step consumed the accumulators of ITS bodies only; everyone else's are still pending for the fused kernel.
    bool forked = false;
    if (fork_fused_enabled()) {
        if (!b->fork_stream) {
            HIP_TRY(hipStreamCreateWithFlags(&b->fork_stream, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&b->fork_ev, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&b->join_ev, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(b->fork_ev, b->stream));
        HIP_TRY(hipStreamWaitEvent(b->fork_stream, b->fork_ev, 0));
        hipStream_t main_stream = b->stream;
        b->stream = b->fork_stream;
        rc = fused_tick<T>(b, h, false, (const uint8_t *)b->bp_inpair.p);
        b->stream = main_stream;
        if (rc != DMX_OK) return rc;
        HIP_TRY(hipEventRecord(b->join_ev, b->fork_stream));
        forked = true;
    }
    HIP_TRY(launch_islands<T>((T *)b->slab, b->bflags, b->stride, I, P, b->diag_isl, b->stream));
    ph.reset(new DmxPhase(b, 8));
    if (forked) HIP_TRY(hipStreamWaitEvent(b->stream, b->join_ev, 0));
    else if ((rc = fused_tick<T>(b, h, false, (const uint8_t *)b->bp_inpair.p)) != DMX_OK) return rc;
    ph.reset();
    b->last_islands = false;
    b->last_mixed = true;
    b->stepped_with_plane = true;
    return DMX_OK;
}

// What the static fused path reported about the chunk just rolled back.  BPF_NEED8: a body has 5..8 contacts with static
// geometry and the launch for those was left out -- from now on it rides along, and the chunk can simply run again.
// BPF_NOFAST: a body has more contacts than the fused path's buffer holds -- the exact path steps the chunk, and the next
// 1, 2, 4 .. 64 chunks too (a body wedged in a corner stays there; a clean fast chunk walks the hold back down).
// Returns true if either was up.
bool static_flags_say(dmxBatch *b, bool *go_exact)
{
    const uint32_t *f = b->bp_flags_host;
    if (f[BPF_NOFAST]) {
        b->static_need8 = b->static_need8 || f[BPF_NEED8];
        b->nofast_hold = 1 << b->nofast_level;
        if (b->nofast_level < 6) b->nofast_level++;
        *go_exact = true;
        return true;
    }
    if (f[BPF_NEED8] && !b->static_need8) { b->static_need8 = true; *go_exact = false; return true; }
    return false;
}

// The collision-checked loop, synchronous form: every chunk's flag is read (one stream synchronisation) before the
// call returns.  The lazy form below defers that read; this one is what it falls back to after a violation.
template <class T> int step_sync_t(dmxBatch *b, double h, int nsteps)
{
    int rc;
    int remaining = nsteps;
    while (remaining > 0) {
        if (b->ext_pending) {
            // freshly applied dBodyAddForce / AddTorque accumulators are consumed (and cleared) by exactly one tick:
            // take that tick on the exact path so a rolled-back chunk can never lose them
            if ((rc = ensure_buffers(b)) != DMX_OK) return rc;
            if ((rc = careful_tick<T>(b, h)) != DMX_OK) return rc;
            b->bp_valid = false;
            remaining--;
            continue;
        }
        if (!b->bp_valid && (rc = build_safe_zones<T>(b)) != DMX_OK) return rc;
        if (b->bp_chunk < kChunk) b->bp_chunk = kChunk;
        int k = std::min(remaining, b->bp_chunk);
        bool careful = b->bp_crowded > 0 || b->bp_skip_fast || b->nofast_hold > 0;
        b->bp_skip_fast = false;
        if (b->nofast_hold > 0) b->nofast_hold--;
        // fast chunk: snapshot, k fused ticks with the safe-zone check riding along, one flag read.  If a body
        // left its zone the chunk is rolled back; the first retry only refreshes the zones (a body that has
        // drifted since the last build usually fits again), the second replays the chunk exactly.
        for (int attempt = 0; !careful; attempt++) {
            DmxPhase pf(b, 3);
            if ((rc = snapshot_begin<T>(b)) != DMX_OK) return rc;
            HIP_TRY(hipMemsetAsync((uint32_t *)b->bp_flags.p + BPF_VIOLATION, 0, BPF_CHUNK_FLAGS * sizeof(uint32_t), b->stream));
            // Without a ground plane and with gravity along y nothing acts horizontally: every body's (x,z) moves on a
            // straight line during the chunk, and a disc is convex, so a body inside its zone at the chunk's first and
            // last tick is inside it at every tick between -- two checks per chunk prove all of them.
            const bool ballistic = !b->plane_on && b->n_static == 0 && b->g[0] == 0.0 && b->g[2] == 0.0;
            if ((rc = fused_run<T>(b, h, k, ballistic, true, true)) != DMX_OK) return rc;
            if ((rc = read_flags(b)) != DMX_OK) return rc;
            if (!b->bp_flags_host[BPF_VIOLATION]) {
                snapshot_drop(b);
                b->stat_fast_ticks += k;
                b->last_pairs = 0;
                b->last_mixed = false;
                if (b->bp_flags_host[BPF_WARN]) b->bp_valid = false;     // zones are getting used up: refresh before the next chunk
                else if (k == b->bp_chunk) b->bp_chunk = std::min(2 * b->bp_chunk, kChunkMax);   // quiet scene: snapshot and read the flag less often
                if (b->nofast_level > 0) b->nofast_level--;
                break;
            }
            if ((rc = snapshot_restore<T>(b)) != DMX_OK) return rc;
            b->stat_rollbacks++;
            if (static_flags_say(b, &careful)) {
                if (careful) { b->bp_chunk = kChunk; k = std::min(k, kChunk); }
                else attempt--;                 // the same chunk again, now with the launch for 5..8-contact bodies
                continue;
            }
            b->bp_chunk = kChunk;
            const bool same_again = b->bp_fresh && k <= kChunk;    // zones built at these very poses, chunk no longer than the retry's:
            k = std::min(k, kChunk);            // the retry (and an exact replay, if it comes to that) covers a short chunk
            if (attempt == 0 && same_again) {
                careful = true;                 // ... the retry would run the same ticks against the same zones
            } else if (attempt == 0) {
                if (!b->bp_fresh && (rc = build_safe_zones<T>(b)) != DMX_OK) return rc;
                careful = b->bp_crowded > 0;
            } else {
                careful = true;
            }
        }
        if (!careful) { remaining -= k; continue; }
        for (int s = 0; s < k; s++)
            if ((rc = careful_tick<T>(b, h)) != DMX_OK) return rc;
        b->bp_valid = false;           // poses moved: new safe zones before the next fast chunk
        remaining -= k;
    }
    b->stepped_with_plane = dmx_fused_contacts(b) || b->last_mixed;
    b->last_islands = false;
    return DMX_OK;
}

// ---- lazy chunks ----------------------------------------------------------------------------------------------
// dmxBatchStep returns without reading the chunk's violation flag: the chunk stays OPEN across calls (the reference's
// loop issues a tick or two per frame, main.c:211) and is closed -- one zone test of the current poses for a ballistic
// chunk, one flag read -- when it reaches its length or when anything observes or changes the batch (every other entry
// point settles it first).  A violation found at the close rolls the whole chunk back to its snapshot and replays its
// calls through the synchronous loop above, so what a caller can observe is unchanged; a clean scene never waits for
// the host inside a run of Step calls.
template <class T> int close_chunk_t(dmxBatch *b)
{
    int rc;
    dmxBatch::OpenChunk &oc = b->oc;
    if (!oc.open) return DMX_OK;
    oc.open = false;
    if (oc.ballistic && !oc.last_checked)
        // the poses after the chunk's last tick inside their zones (and the first tick's before it: tested in that
        // launch) prove every tick between: each body's (x,z) moved on a straight line and a disc is convex
        HIP_TRY(launch_check_zones<T>((const T *)b->slab, 0, b->n_active, (uint32_t *)b->bp_flags.p, b->stream));
    if ((rc = read_flags(b)) != DMX_OK) return rc;
    if (!b->bp_flags_host[BPF_VIOLATION]) {
        snapshot_drop(b);
        b->stat_fast_ticks += oc.ticks;
        b->last_pairs = 0;
        b->last_mixed = false;
        if (b->bp_flags_host[BPF_WARN]) b->bp_valid = false;
        else if (oc.ticks >= b->bp_chunk) b->bp_chunk = std::min(2 * b->bp_chunk, kChunkMax);
        if (b->nofast_level > 0) b->nofast_level--;
        oc.segs.clear();
        return DMX_OK;
    }
    if ((rc = snapshot_restore<T>(b)) != DMX_OK) return rc;
    b->stat_rollbacks++;
    bool go_exact = false;
    if (static_flags_say(b, &go_exact)) {
        if (go_exact) b->bp_chunk = kChunk;      // (nofast_hold sends the replay the exact way; otherwise it runs fast again as it was)
    } else {
    b->bp_chunk = kChunk;
    if (b->bp_fresh && oc.ticks <= kChunk) b->bp_skip_fast = true;     // same poses, same zones: the replay's first chunk goes the exact way
    else b->bp_valid = false;            // fresh zones first: a body that has drifted since the last build usually fits again
    }
    std::vector<std::pair<double, int>> segs;
    segs.swap(oc.segs);
    for (auto &sg : segs)
        if ((rc = step_sync_t<T>(b, sg.first, sg.second)) != DMX_OK) return rc;
    return DMX_OK;
}

template <class T> int step_collide_t(dmxBatch *b, double h, int nsteps)
{
    int rc;
    dmxBatch::OpenChunk &oc = b->oc;
    int remaining = nsteps;
    while (remaining > 0) {
        if (!oc.open) {
            if (b->ext_pending || !b->lazy_chunks) return step_sync_t<T>(b, h, remaining);
            if (!b->bp_valid && (rc = build_safe_zones<T>(b)) != DMX_OK) return rc;
            if (b->bp_crowded > 0 || b->nofast_hold > 0) return step_sync_t<T>(b, h, remaining);
            if (b->bp_chunk < kChunk) b->bp_chunk = kChunk;
            if ((rc = snapshot_begin<T>(b)) != DMX_OK) return rc;
            HIP_TRY(hipMemsetAsync((uint32_t *)b->bp_flags.p + BPF_VIOLATION, 0, BPF_CHUNK_FLAGS * sizeof(uint32_t), b->stream));
            oc.open = true; oc.ticks = 0; oc.budget = b->bp_chunk; oc.last_checked = false;
            oc.ballistic = !b->plane_on && b->n_static == 0 && b->g[0] == 0.0 && b->g[2] == 0.0;
            oc.segs.clear();
        }
        const int k = std::min(remaining, oc.budget - oc.ticks);
        const bool closes = oc.ticks + k >= oc.budget;         // the chunk's last tick is in this run: test it there
        if ((rc = fused_run<T>(b, h, k, oc.ballistic, oc.ticks == 0, closes)) != DMX_OK) return rc;
        oc.last_checked = closes;
        b->stepped_with_plane = dmx_fused_contacts(b);
        b->last_islands = false;
        b->last_mixed = false;
        b->last_pairs = 0;
        if (!oc.segs.empty() && oc.segs.back().first == h) oc.segs.back().second += k;
        else oc.segs.push_back({ h, k });
        oc.ticks += k;
        remaining -= k;
        if (closes && (rc = close_chunk_t<T>(b)) != DMX_OK) return rc;      // (a replay after a violation sets them anew)
    }
    return DMX_OK;
}

// ---- the same loop in pieces, for a caller that interleaves its own per-tick work (shard.py's exchange) ----
template <class T> int chunk_begin_t(dmxBatch *b, int *exact_only, int *ballistic)
{
    int rc;
    if ((rc = ensure_buffers(b)) != DMX_OK) return rc;
    if (!b->bp_valid && (rc = build_safe_zones<T>(b)) != DMX_OK) return rc;
    *exact_only = (b->bp_crowded > 0 || b->ext_pending || b->nofast_hold > 0) ? 1 : 0;
    if (b->nofast_hold > 0) b->nofast_hold--;
    *ballistic = (!b->plane_on && b->n_static == 0 && b->g[0] == 0.0 && b->g[2] == 0.0) ? 1 : 0;
    if ((rc = snapshot_begin<T>(b)) != DMX_OK) return rc;
    HIP_TRY(hipMemsetAsync((uint32_t *)b->bp_flags.p + BPF_VIOLATION, 0, BPF_CHUNK_FLAGS * sizeof(uint32_t), b->stream));
    return DMX_OK;
}

template <class T> int chunk_rollback_t(dmxBatch *b)
{
    int rc;
    if ((rc = snapshot_restore<T>(b)) != DMX_OK) return rc;
    b->stat_rollbacks++;
    b->bp_valid = false;
    return DMX_OK;
}

template <class T> int check_zones_t(dmxBatch *b, hipStream_t st, int64_t first, int64_t count)
{
    HIP_TRY(launch_check_zones<T>((const T *)b->slab, first, count, (uint32_t *)b->bp_flags.p, st));
    return DMX_OK;
}

}  // namespace

namespace {
template <class T> int find_pairs_t(dmxBatch *b)
{
    int rc;
    if ((rc = ensure_buffers(b)) != DMX_OK) return rc;
    if (!b->ex_counts_host && (rc = alloc_host_record(&b->ex_counts_host, sizeof(ExactCounts))) != DMX_OK) return rc;
    ExactBuffers<T> B;
    ExactCounts &C = *(ExactCounts *)b->ex_counts_host;
    if (b->ex_cap_pairs == 0) { b->ex_cap_pairs = 1024; b->ex_cap_rows = 16384; }
    for (int attempt = 0;; attempt++) {
        if (attempt > 40) return DMX_ECAPACITY;

        ExactCaps cap;
        cap.pairs = b->ex_cap_pairs;
        cap.inv = (uint32_t)std::min<int64_t>(2 * (int64_t)cap.pairs, b->n_active);
        cap.rows = b->ex_cap_rows;
        cap.nstatic = (uint32_t)b->n_static;
        if ((rc = ensure_exact_buffers<T>(b, cap, B)) != DMX_OK) return rc;
        // dmxBatchFindPairs' contract (include/dmx_batch.h): EVERY body whose AABB overlaps a static box is involved -- the
        // caller (dSpaceCollide of the ODE API) hands those pairs to the user's callback itself.  (After the buffers: the
        // grid's per-body records are one of them.)
        GridParams<T> Gfp = grid_of<T>(b);
        Gfp.static_fast = 0;
        bool staged = false;
        uint32_t small_seq = 0;
        if (use_small_exact(b, cap, false)) {
            ExactCounts *hc; uint32_t *hf;
            if ((rc = host_record_pointers(b, &hc, &hf)) != DMX_OK) return rc;
            small_seq = next_record_seq();
            HIP_TRY(launch_exact_small_front<T>((T *)b->slab, b->gtype, b->n, b->n_active, Gfp, B, cap, hc, hf, small_seq, b->stream));
        } else {
            if ((rc = fill_grid<T>(b, B.counts, sizeof(ExactCounts))) != DMX_OK) return rc;
            HIP_TRY(launch_exact_pairs<T>((const T *)b->slab, b->gtype, b->n_active, Gfp, B, cap, b->stream));
            HIP_TRY(hipMemcpyAsync(&C, B.counts, sizeof(ExactCounts), hipMemcpyDeviceToHost, b->stream));
            staged = true;
        }
        if (staged) {
            HIP_TRY(hipStreamSynchronize(b->stream));
            b->bp_flags_host[BPF_OVERFLOW] = C.bp_overflow;
        } else if ((rc = await_host_record(b, small_seq)) != DMX_OK) return rc;
        if (b->bp_flags_host[BPF_OVERFLOW]) { if ((rc = grow_buckets(b)) != DMX_OK) return rc; continue; }
        if (C.overflow & 1u) {
            const uint64_t need = std::max<uint64_t>(C.npairs, (uint64_t)(C.ninv + 1) / 2);
            b->ex_cap_pairs = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(2ull * cap.pairs, need + need / 4 + 64), 1ull << 28);
            continue;
        }
        break;
    }
    b->fp_pairs.resize((size_t)2 * C.npairs);
    b->fp_inv.resize((size_t)C.ninv);
    b->fp_cross.resize((size_t)2 * std::min<uint32_t>(C.ncross, EX_CROSS_CAP));
    if (!b->fp_cross.empty())
        HIP_TRY(hipMemcpyAsync(b->fp_cross.data(), B.cross_list, b->fp_cross.size() * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    if (C.npairs) HIP_TRY(hipMemcpyAsync(b->fp_pairs.data(), B.pairs, b->fp_pairs.size() * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    if (C.ninv) HIP_TRY(hipMemcpyAsync(b->fp_inv.data(), B.inv, b->fp_inv.size() * sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return DMX_OK;
}
}  // namespace

int dmx_find_pairs(dmxBatch *b)
{
    return b->precision == DMX_F32 ? find_pairs_t<float>(b) : find_pairs_t<double>(b);
}

int dmx_settle(dmxBatch *b)
{
    if (!b->oc.open) return DMX_OK;
    HIP_TRY(hipSetDevice(b->device));
    return b->precision == DMX_F32 ? close_chunk_t<float>(b) : close_chunk_t<double>(b);
}

int dmx_step_collide(dmxBatch *b, double h, int nsteps)
{
    return b->precision == DMX_F32 ? step_collide_t<float>(b, h, nsteps) : step_collide_t<double>(b, h, nsteps);
}

int dmx_chunk_begin(dmxBatch *b, int *exact_only, int *ballistic)
{
    return b->precision == DMX_F32 ? chunk_begin_t<float>(b, exact_only, ballistic) : chunk_begin_t<double>(b, exact_only, ballistic);
}

int dmx_chunk_tick(dmxBatch *b, double h, int check)
{
    if (!b->bp_flags.p) return DMX_EINVAL;          // no chunk begun
    const int rc = b->precision == DMX_F32 ? fused_tick<float>(b, h, check != 0, nullptr) : fused_tick<double>(b, h, check != 0, nullptr);
    b->stepped_with_plane = dmx_fused_contacts(b);
    b->last_islands = false;
    b->last_mixed = false;
    b->last_pairs = 0;
    return rc;
}

int dmx_chunk_ticks(dmxBatch *b, double h, int n, int check_first, int check_last)
{
    if (!b->bp_flags.p) return DMX_EINVAL;          // no chunk begun
    const int rc = b->precision == DMX_F32 ? fused_run<float>(b, h, n, true, check_first != 0, check_last != 0)
                                           : fused_run<double>(b, h, n, true, check_first != 0, check_last != 0);
    b->stepped_with_plane = dmx_fused_contacts(b);
    b->last_islands = false;
    b->last_mixed = false;
    b->last_pairs = 0;
    return rc;
}

int dmx_check_zones(dmxBatch *b, hipStream_t st, int64_t first, int64_t count)
{
    if (!b->bp_flags.p) return DMX_EINVAL;
    return b->precision == DMX_F32 ? check_zones_t<float>(b, st, first, count) : check_zones_t<double>(b, st, first, count);
}

int dmx_chunk_end(dmxBatch *b, int *violated, int *warn)
{
    if (!b->bp_flags.p) return DMX_EINVAL;
    const int rc = read_flags(b);
    if (rc != DMX_OK) return rc;
    *violated = b->bp_flags_host[BPF_VIOLATION] ? 1 : 0;
    *warn = b->bp_flags_host[BPF_WARN] ? 1 : 0;
    if (*violated) { bool go_exact; (void)static_flags_say(b, &go_exact); }     // (the caller's replay finds need8 / the hold set)
    return DMX_OK;
}

int dmx_chunk_commit(dmxBatch *b, int ticks, int refresh_zones)
{
    snapshot_drop(b);
    b->stat_fast_ticks += ticks;
    if (refresh_zones) b->bp_valid = false;
    return DMX_OK;
}

int dmx_chunk_rollback(dmxBatch *b)
{
    if (!b->bp_flags.p) return DMX_EINVAL;           // no chunk begun
    return b->precision == DMX_F32 ? chunk_rollback_t<float>(b) : chunk_rollback_t<double>(b);
}

int dmx_exact_tick(dmxBatch *b, double h)
{
    int rc = ensure_buffers(b);
    if (rc != DMX_OK) return rc;
    rc = b->precision == DMX_F32 ? careful_tick<float>(b, h) : careful_tick<double>(b, h);
    b->bp_valid = false;
    b->stepped_with_plane = dmx_fused_contacts(b) || b->last_mixed;
    b->last_islands = false;
    return rc;
}
