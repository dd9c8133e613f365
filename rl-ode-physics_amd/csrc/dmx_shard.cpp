// dmx_shard.cpp -- one rank's tick loop of the island-sharded world behind the C ABI (include/dmx_shard.h).
//
// The reference steps one world on one thread (/root/reference/src/main.c:206-216); this is the multi-GPU form of that loop:
// dynamics islands are independent, a rank owns a slab of them, and the boundary rows' state goes to the neighbours by an
// all-gather (RCCL over xGMI).  Everything here is host choreography over the batch ABI (include/dmx_batch.h) and two
// collectives: which ticks are tested, when the side stream exchanges, what happens when a body leaves its safe zone anywhere.
// Every decision that shapes the sequence of collectives is taken on flags OR-ed over the ranks, so ranks cannot diverge.
#include <dlfcn.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "dmx_batch_priv.hpp"
#include "../../include/dmx_shard.h"

namespace {

constexpr int STATE_REALS = C_MASS;        // pos3 quat4 lvel3 avel3
constexpr int kChunkMin = 32, kChunkMax = 256;
constexpr int kNoticeCap = 64;             // bodies one rank can adopt in one migration round
constexpr int GEO_REALS = 8;               // sides3, class, mass, inertia3: what a neighbour must know about a boundary body

#define SH_TRY(expr)                                   \
    do {                                               \
        const int rc_ = (expr);                        \
        if (rc_ != DMX_OK) return rc_;                 \
    } while (0)

// ---- RCCL, loaded when asked for (the library does not link it: a process that never shards never needs it, and one that
//      already carries a librccl -- PyTorch's -- keeps using that one: same soname) ---------------------------------------
struct RcclApi {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, const void * /* ncclUniqueId by value: 128 bytes, passed in memory */, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
struct NcclId { char bytes[DMX_RCCL_ID_BYTES]; };
typedef int (*comm_init_fn)(void **, int, NcclId, int);        // ncclCommInitRank takes the id BY VALUE

RcclApi *rccl()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.lib ? &api : nullptr;
    tried = true;
    for (const char *name : { "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" }) {
        api.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (api.lib) break;
    }
    if (!api.lib) { fprintf(stderr, "libode_mi355: cannot load librccl (%s)\n", dlerror()); return nullptr; }
    auto sym = [&](const char *n) { return dlsym(api.lib, n); };
    api.GetUniqueId = (int (*)(void *))sym("ncclGetUniqueId");
    api.CommInitRank = (int (*)(void **, int, const void *, int))sym("ncclCommInitRank");
    api.CommDestroy = (int (*)(void *))sym("ncclCommDestroy");
    api.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))sym("ncclAllGather");
    api.AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))sym("ncclAllReduce");
    api.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce) {
        fprintf(stderr, "libode_mi355: librccl lacks an entry point this library needs\n");
        api.lib = nullptr;
        return nullptr;
    }
    return &api;
}
constexpr int kNcclInt8 = 0, kNcclInt32 = 2, kNcclMax = 2;      // ncclDataType_t / ncclRedOp_t values (nccl.h)

struct RcclCtx {
    void *comm = nullptr;
    int32_t *flags_dev = nullptr;
    int32_t *flags_host = nullptr;     // pinned
    hipStream_t stream = nullptr;      // the shard's side stream: the flag all-reduce is enqueued where the all-gathers are (one
                                       // communicator, one stream: the collectives of a rank are ordered without RCCL having to)
};
int rccl_check(int rc, const char *what)
{
    if (rc == 0) return DMX_OK;
    RcclApi *r = rccl();
    fprintf(stderr, "libode_mi355: %s failed: %s\n", what, r && r->GetErrorString ? r->GetErrorString(rc) : "?");
    return DMX_EHIP;
}
int rccl_all_gather(void *ctx, const void *send, void *recv, size_t bytes, void *stream)
{
    RcclCtx *c = (RcclCtx *)ctx;
    return rccl_check(rccl()->AllGather(send, recv, bytes, kNcclInt8, c->comm, (hipStream_t)stream), "ncclAllGather");
}
int rccl_all_reduce_max(void *ctx, int32_t *vals, int n)
{
    RcclCtx *c = (RcclCtx *)ctx;
    if (n > 16) return DMX_EINVAL;
    memcpy(c->flags_host, vals, (size_t)n * sizeof(int32_t));
    HIP_TRY(hipMemcpyAsync(c->flags_dev, c->flags_host, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    SH_TRY(rccl_check(rccl()->AllReduce(c->flags_dev, c->flags_dev, (size_t)n, kNcclInt32, kNcclMax, c->comm, c->stream), "ncclAllReduce"));
    HIP_TRY(hipMemcpyAsync(c->flags_host, c->flags_dev, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    memcpy(vals, c->flags_host, (size_t)n * sizeof(int32_t));
    return DMX_OK;
}

}  // namespace

struct dmxShard {
    dmxBatch *b = nullptr;
    int64_t side = 0, rows = 0, spare = 0, n = 0, n_active = 0, n_total = 0, n_send = 0;
    int rank = 0, world = 1;
    size_t rsize = 4;
    dmxCollectives coll{};
    RcclCtx *own = nullptr;                 // set when the collectives are this library's RCCL binding
    hipStream_t side_stream = nullptr;
    hipEvent_t packed = nullptr, done[2] = { nullptr, nullptr };
    bool have_done[2] = { false, false };
    int last = 0;
    void *send[2] = { nullptr, nullptr }, *recv = nullptr;
    int32_t *send_idx = nullptr;            // device: the lower row's slots, then the upper row's
    void *scratch = nullptr; size_t scratch_bytes = 0;      // device staging of the small set-up / migration all-gathers
    int64_t k = 0;                          // exchanges issued
    int chunk = kChunkMin;
    bool armed = false;
    struct Open { bool open = false, checked = false; int ticks = 0, budget = 0; std::vector<std::pair<double, int>> segs; } oc;
    std::vector<uint8_t> ghost_gtype;       // class of every ghost row body, by ghost slot - n_active
    int64_t spare_used = 0;
    int64_t stat[6] = { 0, 0, 0, 0, 0, 0 };
};

namespace {

hipStream_t main_stream(dmxShard *s) { return s->b->stream; }

int ensure_scratch(dmxShard *s, size_t bytes)
{
    if (bytes <= s->scratch_bytes) return DMX_OK;
    if (s->scratch) HIP_TRY(hipFree(s->scratch));
    s->scratch = nullptr; s->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&s->scratch, bytes));
    s->scratch_bytes = bytes;
    return DMX_OK;
}

// a small host array of doubles, `count` per rank, all-gathered (set-up and migration notices: off the tick path)
int gather_host(dmxShard *s, const std::vector<double> &mine, std::vector<double> &all)
{
    const size_t bytes = mine.size() * sizeof(double);
    SH_TRY(ensure_scratch(s, bytes * (size_t)(s->world + 1)));
    char *dev = (char *)s->scratch;
    HIP_TRY(hipMemcpyAsync(dev, mine.data(), bytes, hipMemcpyHostToDevice, s->side_stream));
    if (s->coll.all_gather(s->coll.ctx, dev, dev + bytes, bytes, s->side_stream) != 0) return DMX_EHIP;
    all.resize(mine.size() * (size_t)s->world);
    HIP_TRY(hipMemcpyAsync(all.data(), dev + bytes, bytes * (size_t)s->world, hipMemcpyDeviceToHost, s->side_stream));
    HIP_TRY(hipStreamSynchronize(s->side_stream));
    return DMX_OK;
}

// element-wise OR of a few host flags over the ranks
int any_rank(dmxShard *s, bool *flags, int n)
{
    if (s->world == 1) return DMX_OK;
    int32_t v[8];
    for (int i = 0; i < n; i++) v[i] = flags[i] ? 1 : 0;
    if (s->coll.all_reduce_max(s->coll.ctx, v, n) != 0) return DMX_EHIP;
    for (int i = 0; i < n; i++) flags[i] = v[i] != 0;
    return DMX_OK;
}

// ---- the exchange: ring of two send buffers, side stream, events -------------------------------------------------------
int ex_drain(dmxShard *s)
{
    if (s->have_done[0] || s->have_done[1]) HIP_TRY(hipStreamWaitEvent(main_stream(s), s->done[s->last], 0));   // the side stream is in order
    return DMX_OK;
}
int ex_disarm(dmxShard *s)
{
    if (s->armed) { SH_TRY(dmxBatchSetBoundaryPack(s->b, nullptr, 0, 0)); s->armed = false; }
    return DMX_OK;
}
int ex_before_step(dmxShard *s, bool fused)
{
    const int slot = (int)(s->k % 2);
    if (s->have_done[slot]) HIP_TRY(hipStreamWaitEvent(main_stream(s), s->done[slot], 0));   // the exchange that last read this send buffer
    if (fused) { SH_TRY(dmxBatchSetBoundaryPack(s->b, s->send[slot], s->side, s->n - s->side)); s->armed = true; }
    else SH_TRY(ex_disarm(s));
    return DMX_OK;
}
int ex_pack(dmxShard *s, bool fused)
{
    // (the step kernel packs the two boundary rows itself; the spare slots -- bodies adopted from the upper neighbour, which that
    //  neighbour must keep seeing -- follow by a small gather)
    if (!fused) SH_TRY(dmxBatchGatherBodies(s->b, s->send_idx, s->n_send, s->send[s->k % 2]));
    else if (s->spare > 0)
        SH_TRY(dmxBatchGatherBodies(s->b, s->send_idx + 2 * s->side, s->spare, (char *)s->send[s->k % 2] + (size_t)(2 * s->side) * STATE_REALS * s->rsize));
    HIP_TRY(hipEventRecord(s->packed, main_stream(s)));
    HIP_TRY(hipStreamWaitEvent(s->side_stream, s->packed, 0));
    return DMX_OK;
}
int ex_exchange(dmxShard *s, bool check_ghosts)
{
    const int slot = (int)(s->k % 2);
    const size_t row_bytes = (size_t)s->side * STATE_REALS * s->rsize, bytes = (size_t)s->n_send * STATE_REALS * s->rsize;
    if (s->coll.all_gather(s->coll.ctx, s->send[slot], s->recv, bytes, s->side_stream) != 0) return DMX_EHIP;
    const char *r = (const char *)s->recv;
    const void *lo = s->rank > 0 ? r + (size_t)(s->rank - 1) * bytes + row_bytes : nullptr;            // lower neighbour's upper row
    const void *hi = s->rank < s->world - 1 ? r + (size_t)(s->rank + 1) * bytes : nullptr;            // upper neighbour's lower row
    if (lo || hi) SH_TRY(dmxBatchRefreshGhostsOnStream(s->b, s->side_stream, s->n_active, s->side, lo, s->side, hi, check_ghosts ? 1 : 0));
    // the lower neighbour's spare slots: the bodies it adopted from THIS rank come back as ghosts, so that this rank's bodies
    // behind the boundary row still see them (zones, pair search); slots of bodies not adopted are class NONE and inert
    if (s->rank > 0 && s->spare > 0)
        SH_TRY(dmxBatchRefreshGhostsOnStream(s->b, s->side_stream, s->n_active + 2 * s->side, s->spare,
                                             r + (size_t)(s->rank - 1) * bytes + 2 * row_bytes, 0, nullptr, check_ghosts ? 1 : 0));
    HIP_TRY(hipEventRecord(s->done[slot], s->side_stream));
    s->have_done[slot] = true;
    s->last = slot;
    s->k++;
    s->stat[0]++;
    return DMX_OK;
}

int shard_tick(dmxShard *s, double h, bool check)
{
    SH_TRY(ex_before_step(s, true));
    SH_TRY(dmxBatchChunkTick(s->b, h, check ? 1 : 0));
    SH_TRY(ex_pack(s, true));
    return ex_exchange(s, check);
}

// ---- islands that span two ranks: the LOWER rank adopts the upper neighbour's boundary body (shard.py, _migrate) ----------
int migrate(dmxShard *s)
{
    dmxBatch *b = s->b;
    const int64_t side = s->side, hi0 = s->n_active + side;
    for (int round = 0; round < 8; round++) {
        const int32_t *pairs, *inv; int64_t np, ninv;
        SH_TRY(dmxBatchFindPairs(b, &pairs, &np, &inv, &ninv));
        const int32_t *cross; int64_t ncross;
        SH_TRY(dmxBatchCrossPairs(b, &cross, &ncross));
        bool any[1] = { ncross > 0 };
        SH_TRY(any_rank(s, any, 1));
        if (!any[0]) return DMX_OK;
        std::vector<int64_t> adopt;
        int64_t stuck = 0;
        for (int64_t c = 0; c < ncross; c++) {
            const int64_t i = cross[2 * c], g = cross[2 * c + 1];
            if (g >= hi0 && g < hi0 + side) adopt.push_back(g);            // the upper neighbour's first row: this rank adopts
            else if (i >= side) stuck++;           // a body behind my boundary row reaches the lower neighbour's row, or a body the
                                                   // lower neighbour adopted from me: only first-row bodies can follow it down
        }
        std::sort(adopt.begin(), adopt.end());
        adopt.erase(std::unique(adopt.begin(), adopt.end()), adopt.end());
        if ((int64_t)adopt.size() > kNoticeCap || s->spare_used + (int64_t)adopt.size() > s->spare) { stuck++; adopt.clear(); }
        // notice: [count, stuck, (index in the upper neighbour's first row, my spare slot) ...]
        std::vector<double> mine((size_t)2 * kNoticeCap + 2, 0.0), all;
        mine[0] = (double)adopt.size(); mine[1] = (double)stuck;
        for (size_t a = 0; a < adopt.size(); a++) { mine[2 + 2 * a] = (double)(adopt[a] - hi0); mine[3 + 2 * a] = (double)(s->spare_used + (int64_t)a); }
        SH_TRY(gather_host(s, mine, all));
        const size_t nw = (size_t)2 * kNoticeCap + 2;
        double stuck_all = 0;
        for (int r = 0; r < s->world; r++) stuck_all += all[(size_t)r * nw + 1];
        if (stuck_all > 0) {
            fprintf(stderr, "libode_mi355: rank %d: an island spans two ranks and cannot be migrated (a body beyond the boundary row "
                            "reaches across the face, or the spare slots are used up)\n", s->rank);
            return DMX_ECROSS;
        }
        std::vector<char> tmp((size_t)STATE_REALS * 8);
        auto copy_body = [&](int64_t from, int64_t to) -> int {
            for (int field : { DMX_STATE, DMX_MASS, DMX_INERTIA, DMX_SIDES }) {
                SH_TRY(dmxBatchDownload(b, field, tmp.data(), from, 1));
                SH_TRY(dmxBatchUpload(b, field, tmp.data(), to, 1));
            }
            return DMX_OK;
        };
        const uint8_t none = DMX_GEOM_NONE;
        for (int64_t g : adopt) {                              // the ghost becomes a body of this rank's own, in a spare slot
            const int64_t slot = s->n + s->spare_used++;
            SH_TRY(copy_body(g, slot));
            const uint8_t cls = s->ghost_gtype[(size_t)(g - s->n_active)];
            SH_TRY(dmxBatchUploadGeomType(b, &cls, slot, 1));
            SH_TRY(dmxBatchUploadGeomType(b, &none, g, 1));    // the ghost is switched off for good
            s->stat[4]++;
        }
        if (s->rank > 0) {                                     // the lower neighbour adopted these bodies of my first row
            const double *nb = all.data() + (size_t)(s->rank - 1) * nw;
            for (int a = 0; a < (int)nb[0]; a++) {
                const int64_t j = (int64_t)nb[2 + 2 * a], kslot = (int64_t)nb[3 + 2 * a];
                // it stays visible here as a ghost: the slot that mirrors the neighbour's spare slot takes its geometry and, until
                // the next exchange brings the neighbour's copy, its state
                const int64_t ret = s->n_active + 2 * side + kslot;
                SH_TRY(copy_body(j, ret));
                const uint8_t cls = b->h_gtype[(size_t)j];
                SH_TRY(dmxBatchUploadGeomType(b, &cls, ret, 1));
                SH_TRY(dmxBatchUploadGeomType(b, &none, j, 1));
                double park[3] = { 0.0, -1.0e6 - (double)j, 0.0 }, zero[3] = { 0, 0, 0 };
                float parkf[3] = { 0.f, (float)park[1], 0.f }, zerof[3] = { 0, 0, 0 };
                const bool f32 = s->rsize == 4;
                SH_TRY(dmxBatchUpload(b, DMX_POS, f32 ? (const void *)parkf : (const void *)park, j, 1));
                SH_TRY(dmxBatchUpload(b, DMX_LVEL, f32 ? (const void *)zerof : (const void *)zero, j, 1));
                SH_TRY(dmxBatchUpload(b, DMX_AVEL, f32 ? (const void *)zerof : (const void *)zero, j, 1));
                s->stat[5]++;
            }
        }
    }
    fprintf(stderr, "libode_mi355: rank %d: islands spanning two ranks keep growing after 8 migration rounds\n", s->rank);
    return DMX_ECROSS;
}

int exact_tick(dmxShard *s, double h)
{
    SH_TRY(ex_drain(s));
    SH_TRY(migrate(s));
    SH_TRY(ex_before_step(s, false));
    SH_TRY(dmxBatchExactTick(s->b, h));
    SH_TRY(ex_pack(s, false));
    s->stat[3]++;
    return ex_exchange(s, false);
}

// begin a chunk on every rank; (exact_only, ballistic) OR-ed / AND-ed over the ranks
int begin_chunk(dmxShard *s, bool *exact_only, bool *ballistic)
{
    SH_TRY(ex_drain(s));                 // zones, snapshot and flag reset see the last exchange's ghost rows
    int eo, ba;
    SH_TRY(dmxBatchChunkBegin(s->b, &eo, &ba));
    bool f[2] = { eo != 0, ba == 0 };
    SH_TRY(any_rank(s, f, 2));
    *exact_only = f[0]; *ballistic = !f[1];
    return DMX_OK;
}

// k ticks of one chunk (shard.py, _fast_ticks): a ballistic chunk is proven by the test at its first and last tick and only
// its last tick exchanges; otherwise every tick is tested and exchanged
int fast_ticks(dmxShard *s, double h, int k, bool ballistic)
{
    if (ballistic) {
        SH_TRY(ex_disarm(s));
        if (k > 1) SH_TRY(dmxBatchChunkTicks(s->b, h, k - 1, 1, 0));
        return shard_tick(s, h, true);
    }
    for (int t = 0; t < k; t++) SH_TRY(shard_tick(s, h, true));
    return DMX_OK;
}

// up to k ticks as one closed chunk; *advanced = the ticks taken
int run_chunk(dmxShard *s, double h, int k, const bool *begun, int *advanced)
{
    for (int attempt = 0; attempt < 3; attempt++) {
        bool exact_only, ballistic;
        if (begun && attempt == 0) { exact_only = begun[0]; ballistic = begun[1]; }
        else SH_TRY(begin_chunk(s, &exact_only, &ballistic));
        if (exact_only || attempt == 2) break;      // crowded bodies or pending forces somewhere: everyone steps exactly
        SH_TRY(fast_ticks(s, h, k, ballistic));
        SH_TRY(ex_drain(s));
        int violated, warn_here;
        SH_TRY(dmxBatchChunkEnd(s->b, &violated, &warn_here));
        bool f[2] = { violated != 0, warn_here != 0 };
        SH_TRY(any_rank(s, f, 2));
        if (!f[0]) {
            SH_TRY(dmxBatchChunkCommit(s->b, k, warn_here));
            if (!f[1] && k >= s->chunk) s->chunk = std::min(2 * s->chunk, kChunkMax);
            s->stat[1]++;
            *advanced = k;
            return DMX_OK;
        }
        SH_TRY(dmxBatchChunkRollback(s->b));        // every rank returns to the chunk's start (ghost slots included)
        s->stat[2]++;
        s->chunk = kChunkMin;
        k = std::min(k, kChunkMin);
    }
    for (int t = 0; t < k; t++) SH_TRY(exact_tick(s, h));
    *advanced = k;
    return DMX_OK;
}

int run_chunks(dmxShard *s, double h, int nsteps, const bool *begun)
{
    int remaining = nsteps;
    while (remaining > 0) {
        int adv = 0;
        SH_TRY(run_chunk(s, h, std::min(remaining, s->chunk), begun, &adv));
        begun = nullptr;
        remaining -= adv;
    }
    return DMX_OK;
}

int settle(dmxShard *s)
{
    dmxShard::Open oc;
    std::swap(oc, s->oc);
    if (!oc.open) return DMX_OK;
    if (!oc.checked) {
        // closed before its length: the poses after its last tick inside their zones prove the ticks before (straight
        // horizontal lines, convex zones); the boundary rows go out by an explicit gather
        SH_TRY(dmxBatchCheckZonesOnStream(s->b, main_stream(s), 0, s->n_active));
        SH_TRY(ex_before_step(s, false));
        SH_TRY(ex_pack(s, false));
        SH_TRY(ex_exchange(s, true));
    }
    SH_TRY(ex_drain(s));
    int violated, warn_here;
    SH_TRY(dmxBatchChunkEnd(s->b, &violated, &warn_here));
    bool f[2] = { violated != 0, warn_here != 0 };
    SH_TRY(any_rank(s, f, 2));
    if (!f[0]) {
        SH_TRY(dmxBatchChunkCommit(s->b, oc.ticks, warn_here));
        if (!f[1] && oc.ticks >= s->chunk) s->chunk = std::min(2 * s->chunk, kChunkMax);
        s->stat[1]++;
        return DMX_OK;
    }
    SH_TRY(dmxBatchChunkRollback(s->b));
    s->stat[2]++;
    s->chunk = kChunkMin;
    for (auto &sg : oc.segs) SH_TRY(run_chunks(s, sg.first, sg.second, nullptr));
    return DMX_OK;
}

// lazily closed ballistic chunks (shard.py, _run_lazy): a chunk stays open across Run calls, so a caller issuing a few ticks
// per call pays a chunk's exchange, flag read and flag all-reduce once per 32-256 ticks
int run_lazy(dmxShard *s, double h, int nsteps)
{
    int remaining = nsteps;
    while (remaining > 0) {
        dmxShard::Open &oc = s->oc;
        if (!oc.open) {
            bool begun[2];
            SH_TRY(begin_chunk(s, &begun[0], &begun[1]));
            if (begun[0] || !begun[1]) {
                // crowded bodies, pending forces or bent paths somewhere: this stretch goes chunk by chunk
                const int k = std::min(remaining, s->chunk);
                SH_TRY(run_chunks(s, h, k, begun));
                remaining -= k;
                continue;
            }
            oc.open = true; oc.checked = false; oc.ticks = 0; oc.budget = s->chunk; oc.segs.clear();
        }
        const int k = std::min(remaining, oc.budget - oc.ticks);
        const bool closes = oc.ticks + k >= oc.budget, first = oc.ticks == 0;
        SH_TRY(ex_disarm(s));
        if (closes) {
            // the chunk's last tick carries the zone test and the chunk's one exchange
            if (k > 1) SH_TRY(dmxBatchChunkTicks(s->b, h, k - 1, first ? 1 : 0, 0));
            SH_TRY(shard_tick(s, h, true));
            oc.checked = true;
        } else {
            SH_TRY(dmxBatchChunkTicks(s->b, h, k, first ? 1 : 0, 0));
        }
        if (!oc.segs.empty() && oc.segs.back().first == h) oc.segs.back().second += k;
        else oc.segs.push_back({ h, k });
        oc.ticks += k;
        remaining -= k;
        if (closes) SH_TRY(settle(s));
    }
    return DMX_OK;
}

// Once, at set-up: the neighbours' boundary bodies' extents, classes and mass properties into the ghost slots, so the
// broadphase sees the ghosts at their true size and a ghost can be adopted as it stands
int share_geometry(dmxShard *s)
{
    dmxBatch *b = s->b;
    const int64_t side = s->side;
    std::vector<double> mine((size_t)s->n_send * GEO_REALS), all;
    std::vector<char> tmp((size_t)3 * 8);
    auto rd = [&](int field, int64_t slot, int k, double *out) -> int {
        SH_TRY(dmxBatchDownload(b, field, tmp.data(), slot, 1));
        for (int c = 0; c < k; c++) out[c] = s->rsize == 4 ? (double)((float *)tmp.data())[c] : ((double *)tmp.data())[c];
        return DMX_OK;
    };
    for (int64_t t = 0; t < s->n_send; t++) {
        const int64_t slot = t < side ? t : s->n - side + (t - side);
        double *o = mine.data() + (size_t)t * GEO_REALS;
        SH_TRY(rd(DMX_SIDES, slot, 3, o));
        o[3] = (double)b->h_gtype[(size_t)slot];
        SH_TRY(rd(DMX_MASS, slot, 1, o + 4));
        SH_TRY(rd(DMX_INERTIA, slot, 3, o + 5));
    }
    SH_TRY(gather_host(s, mine, all));
    s->ghost_gtype.assign((size_t)(2 * side), 0);
    auto put = [&](int64_t first, const double *rows, size_t goff) -> int {
        for (int64_t t = 0; t < side; t++) {
            const double *r = rows + (size_t)t * GEO_REALS;
            float f3[3]; double d3[3];
            auto up = [&](int field, const double *v, int k) -> int {
                for (int c = 0; c < k; c++) { f3[c] = (float)v[c]; d3[c] = v[c]; }
                return dmxBatchUpload(b, field, s->rsize == 4 ? (const void *)f3 : (const void *)d3, first + t, 1);
            };
            SH_TRY(up(DMX_SIDES, r, 3));
            const uint8_t cls = (uint8_t)r[3];
            SH_TRY(dmxBatchUploadGeomType(b, &cls, first + t, 1));
            SH_TRY(up(DMX_MASS, r + 4, 1));
            SH_TRY(up(DMX_INERTIA, r + 5, 3));
            s->ghost_gtype[goff + (size_t)t] = cls;
        }
        return DMX_OK;
    };
    if (s->rank > 0) SH_TRY(put(s->n_active, all.data() + ((size_t)(s->rank - 1) * s->n_send + side) * GEO_REALS, 0));
    if (s->rank < s->world - 1) SH_TRY(put(s->n_active + side, all.data() + (size_t)(s->rank + 1) * s->n_send * GEO_REALS, (size_t)side));
    return DMX_OK;
}

int create_common(dmxShard **out, dmxBatch *b, int64_t side, int64_t rows, int64_t spare, int rank, int world, const dmxCollectives &coll,
                  RcclCtx *own)
{
    if (!out || !b || side < 4 || side % 4 || rows < 2 || spare < 0 || spare % 4 || world < 1 || rank < 0 || rank >= world) return DMX_EINVAL;
    const int64_t n = side * rows, n_active = n + spare, n_total = n_active + 2 * side + spare;
    if (b->n != n_total) {
        fprintf(stderr, "libode_mi355: dmxShardCreate: the batch has %lld slots, the layout needs side * rows + 2 * spare + 2 * side = %lld\n",
                (long long)b->n, (long long)n_total);
        return DMX_EINVAL;
    }
    SH_TRY(dmx_settle(b));
    HIP_TRY(hipSetDevice(b->device));
    dmxShard *s = new (std::nothrow) dmxShard();
    if (!s) return DMX_ENOMEM;
    s->b = b; s->side = side; s->rows = rows; s->spare = spare; s->n = n; s->n_active = n_active; s->n_total = n_total;
    s->n_send = 2 * side + spare; s->rank = rank; s->world = world; s->rsize = b->rsize; s->coll = coll; s->own = own;
    int rc = DMX_OK;
    do {
        if ((rc = dmxBatchSetActiveCount(b, n_active)) != DMX_OK) break;
        if (hipStreamCreateWithFlags(&s->side_stream, hipStreamNonBlocking) != hipSuccess) { rc = DMX_EHIP; break; }
        if (hipEventCreateWithFlags(&s->packed, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&s->done[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&s->done[1], hipEventDisableTiming) != hipSuccess) { rc = DMX_EHIP; break; }
        const size_t buf = (size_t)s->n_send * STATE_REALS * s->rsize;
        if (hipMalloc(&s->send[0], buf) != hipSuccess || hipMalloc(&s->send[1], buf) != hipSuccess ||
            hipMalloc(&s->recv, buf * (size_t)world) != hipSuccess || hipMalloc((void **)&s->send_idx, (size_t)s->n_send * sizeof(int32_t)) != hipSuccess) { rc = DMX_ENOMEM; break; }
        std::vector<int32_t> idx((size_t)s->n_send);
        for (int64_t t = 0; t < side; t++) { idx[(size_t)t] = (int32_t)t; idx[(size_t)(side + t)] = (int32_t)(n - side + t); }
        for (int64_t t = 0; t < spare; t++) idx[(size_t)(2 * side + t)] = (int32_t)(n + t);       // the spare slots: bodies adopted from above
        if (hipMemcpy(s->send_idx, idx.data(), idx.size() * sizeof(int32_t), hipMemcpyHostToDevice) != hipSuccess) { rc = DMX_EHIP; break; }
        if ((rc = share_geometry(s)) != DMX_OK) break;
        // prime: one exchange of the boundary rows as they stand, so the ghost slots hold the neighbours' bodies where they are
        // (not at the origin) before the first broadphase build
        if ((rc = ex_before_step(s, false)) != DMX_OK || (rc = ex_pack(s, false)) != DMX_OK || (rc = ex_exchange(s, false)) != DMX_OK ||
            (rc = ex_drain(s)) != DMX_OK) break;
    } while (0);
    if (rc != DMX_OK) { s->own = nullptr; dmxShardDestroy(s); return rc; }
    *out = s;
    return DMX_OK;
}

}  // namespace

extern "C" int dmxShardRcclUniqueId(void *id_out)
{
    if (!id_out) return DMX_EINVAL;
    RcclApi *r = rccl();
    if (!r) return DMX_ENODEVICE;
    return rccl_check(r->GetUniqueId(id_out), "ncclGetUniqueId");
}

extern "C" int dmxShardCreate(dmxShardID *out, dmxBatchID batch, int64_t side, int64_t rows, int64_t spare, int rank, int world,
                              const dmxCollectives *collectives)
{
    if (!collectives || !collectives->all_gather || !collectives->all_reduce_max) return DMX_EINVAL;
    return create_common(out, batch, side, rows, spare, rank, world, *collectives, nullptr);
}

extern "C" int dmxShardCreateRccl(dmxShardID *out, dmxBatchID batch, int64_t side, int64_t rows, int64_t spare, int rank, int world,
                                  const void *rccl_unique_id)
{
    if (!batch || !rccl_unique_id) return DMX_EINVAL;
    RcclApi *r = rccl();
    if (!r) return DMX_ENODEVICE;
    HIP_TRY(hipSetDevice(batch->device));
    RcclCtx *c = new (std::nothrow) RcclCtx();
    if (!c) return DMX_ENOMEM;
    NcclId id;
    memcpy(id.bytes, rccl_unique_id, sizeof(id.bytes));
    int rc = rccl_check(((comm_init_fn)r->CommInitRank)(&c->comm, world, id, rank), "ncclCommInitRank");
    if (rc == DMX_OK && (hipMalloc((void **)&c->flags_dev, 16 * sizeof(int32_t)) != hipSuccess ||
                         hipHostMalloc((void **)&c->flags_host, 16 * sizeof(int32_t), hipHostMallocDefault) != hipSuccess)) rc = DMX_EHIP;
    if (rc == DMX_OK) {
        dmxCollectives coll = { c, rccl_all_gather, rccl_all_reduce_max };
        rc = create_common(out, batch, side, rows, spare, rank, world, coll, c);
    }
    if (rc == DMX_OK) c->stream = (*out)->side_stream;
    else {
        if (c->comm) (void)r->CommDestroy(c->comm);
        if (c->flags_dev) (void)hipFree(c->flags_dev);
        if (c->flags_host) (void)hipHostFree(c->flags_host);
        delete c;
    }
    return rc;
}

// which librccl this process's shard loop is bound to (the path dladdr reports for ncclCommInitRank), and whether `s` holds a
// communicator that ncclCommInitRank brought up: what a log of a multi-GPU run should say about its collectives
extern "C" int dmxShardRcclInfo(dmxShardID s, char *path_out, int cap, int *comm_up)
{
    if (comm_up) *comm_up = (s && s->own && ((RcclCtx *)s->own)->comm) ? 1 : 0;
    if (path_out && cap > 0) {
        path_out[0] = 0;
        RcclApi *r = rccl();
        Dl_info info;
        if (r && dladdr((void *)r->CommInitRank, &info) && info.dli_fname) { strncpy(path_out, info.dli_fname, (size_t)cap - 1); path_out[cap - 1] = 0; }
    }
    return rccl() ? DMX_OK : DMX_ENODEVICE;
}

extern "C" int dmxShardRun(dmxShardID s, double h, int nticks)
{
    if (!s || !(h > 0) || nticks < 0) return DMX_EINVAL;
    HIP_TRY(hipSetDevice(s->b->device));
    return run_lazy(s, h, nticks);
}

extern "C" int dmxShardSettle(dmxShardID s)
{
    if (!s) return DMX_EINVAL;
    HIP_TRY(hipSetDevice(s->b->device));
    SH_TRY(settle(s));
    SH_TRY(ex_drain(s));
    return ex_disarm(s);
}

extern "C" int dmxShardStats(dmxShardID s, int64_t out[6])
{
    if (!s || !out) return DMX_EINVAL;
    for (int i = 0; i < 6; i++) out[i] = s->stat[i];
    return DMX_OK;
}

extern "C" int dmxShardDestroy(dmxShardID s)
{
    if (!s) return DMX_EINVAL;
    (void)hipSetDevice(s->b->device);
    if (s->side_stream) (void)hipStreamSynchronize(s->side_stream);
    if (s->armed) (void)dmxBatchSetBoundaryPack(s->b, nullptr, 0, 0);      // the batch must not keep aiming at buffers that die here
    for (void *p : { s->send[0], s->send[1], s->recv, (void *)s->send_idx, s->scratch }) if (p) (void)hipFree(p);
    for (hipEvent_t e : { s->packed, s->done[0], s->done[1] }) if (e) (void)hipEventDestroy(e);
    hipStream_t side_stream = s->side_stream;
    if (s->own) {
        RcclApi *r = rccl();
        if (r && s->own->comm) (void)r->CommDestroy(s->own->comm);
        if (s->own->flags_dev) (void)hipFree(s->own->flags_dev);
        if (s->own->flags_host) (void)hipHostFree(s->own->flags_host);
        delete s->own;
    }
    if (side_stream) (void)hipStreamDestroy(side_stream);
    delete s;
    return DMX_OK;
}
