// tests/harness/shard_host_stub.cpp -- a HOST stand-in for what csrc/dmx_shard.cpp (the rank loop of the island-sharded world,
// include/dmx_shard.h) calls: the HIP runtime entry points it uses and the dmxBatch* entry points it drives, over arrays in host
// memory.  tests/test_shard_c_loop.py links this file with the product's own dmx_shard.o -- the object code that ships in
// libode_mi355.so, unchanged -- and runs two and three ranks of it over gloo on a machine without a GPU: chunks, rollbacks,
// replays, exact ticks, ghost refreshes and migrations take the paths they take on the device, only the bodies move by a rule
// simple enough to predict (x += h v; one scripted body also gains speed), so that a tick applied twice, dropped, or exchanged late shows in the poses.
//
// Test infrastructure: nothing of the product links or loads this.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <vector>

#include "../../rl-ode-physics_amd/csrc/dmx_batch_priv.hpp"

namespace {

constexpr int SR = 13;      // pos3 quat4 lvel3 avel3

struct HostWorld {
    std::vector<double> st, snap, sides, mass, inertia, zone;       // zone: x, z per slot at the last ChunkBegin
    double safe = 1.0;
    bool violated = false, warn = false;
    int exact_only_chunks = 0;          // script: the first so many ChunkBegin calls answer "exact only"
    int ballistic = 1;                  // script: what ChunkBegin says about the paths
    int64_t accel_slot = -1; double accel = 0;      // script: this own body's v.x grows by h * accel every tick, before it moves
    std::vector<int32_t> cross;         // script: the (own body, ghost slot) pairs the NEXT CrossPairs call reports, once
    std::vector<int32_t> no_pairs;
    // log
    int64_t fast_ticks = 0, exact_ticks = 0, begins = 0, commits = 0, committed_ticks = 0, rollbacks = 0, ghost_refreshes = 0,
            ghost_checks = 0, zone_checks = 0, gathers = 0, packs = 0;
};
std::map<dmxBatch *, HostWorld *> worlds;
HostWorld *W(dmxBatch *b) { return worlds.at(b); }

void zone_test(dmxBatch *b, HostWorld *w, int64_t first, int64_t count)
{
    for (int64_t i = first; i < first + count; i++) {
        if (b->h_gtype[(size_t)i] == DMX_GEOM_NONE) continue;
        const double dx = w->st[(size_t)i * SR] - w->zone[2 * (size_t)i], dz = w->st[(size_t)i * SR + 2] - w->zone[2 * (size_t)i + 1];
        const double d2 = dx * dx + dz * dz, s2 = w->safe * w->safe;
        if (!(d2 < s2)) w->violated = true;
        else if (!(d2 < s2 * 0.0625)) w->warn = true;
    }
}
void pack(dmxBatch *b, HostWorld *w)
{
    if (!b->pack_out) return;
    double *o = (double *)b->pack_out;
    for (int64_t i = 0; i < b->pack_lo; i++) memcpy(o + i * SR, &w->st[(size_t)i * SR], SR * sizeof(double));
    for (int64_t i = 0; i < b->pack_lo; i++) memcpy(o + (b->pack_lo + i) * SR, &w->st[(size_t)(b->pack_hi + i) * SR], SR * sizeof(double));
    w->packs++;
}
void advance(dmxBatch *b, HostWorld *w, double h)
{
    for (int64_t i = 0; i < b->n_active; i++) {
        if (b->h_gtype[(size_t)i] == DMX_GEOM_NONE) continue;
        double *s = &w->st[(size_t)i * SR];
        if (i == w->accel_slot) s[7] = s[7] + h * w->accel;
        for (int c = 0; c < 3; c++) s[c] = s[c] + h * s[7 + c];
    }
}
int field_span(int field, int *off, std::vector<double> HostWorld::**arr)
{
    *arr = &HostWorld::st;
    switch (field) {
    case DMX_POS: *off = 0; return 3;
    case DMX_QUAT: case DMX_QUAT_RAW: *off = 3; return 4;
    case DMX_LVEL: *off = 7; return 3;
    case DMX_AVEL: *off = 10; return 3;
    case DMX_STATE: *off = 0; return 13;
    case DMX_MASS: *arr = &HostWorld::mass; *off = 0; return 1;
    case DMX_INERTIA: *arr = &HostWorld::inertia; *off = 0; return 3;
    case DMX_SIDES: *arr = &HostWorld::sides; *off = 0; return 3;
    }
    return 0;
}

}  // namespace

// ---- the test's own handle on the stand-in -------------------------------------------------------------------------------------
extern "C" int stubBatchCreate(dmxBatchID *out, int64_t n_total)
{
    dmxBatch *b = new dmxBatch();
    b->n = b->stride = n_total; b->n_active = n_total; b->precision = DMX_F64; b->rsize = 8; b->device = 0;
    b->h_gtype.assign((size_t)n_total, DMX_GEOM_NONE);
    HostWorld *w = new HostWorld();
    w->st.assign((size_t)n_total * SR, 0.0); w->sides.assign((size_t)n_total * 3, 0.0); w->mass.assign((size_t)n_total, 0.0);
    w->inertia.assign((size_t)n_total * 3, 0.0); w->zone.assign((size_t)n_total * 2, 0.0);
    worlds[b] = w;
    *out = b;
    return DMX_OK;
}
extern "C" int stubBatchDestroy(dmxBatchID b) { delete worlds.at(b); worlds.erase(b); delete b; return DMX_OK; }
extern "C" int stubBatchScript(dmxBatchID b, double safe, int exact_only_chunks, int ballistic, const int32_t *cross, int n_cross,
                               int64_t accel_slot, double accel)
{
    HostWorld *w = W(b);
    w->accel_slot = accel_slot; w->accel = accel;
    w->safe = safe; w->exact_only_chunks = exact_only_chunks; w->ballistic = ballistic;
    w->cross.assign(cross, cross + 2 * (size_t)n_cross);
    return DMX_OK;
}
extern "C" int stubBatchLog(dmxBatchID b, int64_t out[11])
{
    HostWorld *w = W(b);
    const int64_t v[11] = { w->fast_ticks, w->exact_ticks, w->begins, w->commits, w->committed_ticks, w->rollbacks, w->ghost_refreshes,
                            w->ghost_checks, w->zone_checks, w->gathers, w->packs };
    memcpy(out, v, sizeof(v));
    return DMX_OK;
}
extern "C" int stubBatchGeomType(dmxBatchID b, uint8_t *out) { memcpy(out, b->h_gtype.data(), b->h_gtype.size()); return DMX_OK; }

// ---- the batch entry points the rank loop drives (include/dmx_batch.h) -----------------------------------------------------------
int dmx_settle(dmxBatch *) { return DMX_OK; }

extern "C" int dmxBatchUpload(dmxBatchID b, int field, const void *src, int64_t first, int64_t count)
{
    int off; std::vector<double> HostWorld::*arr;
    const int k = field_span(field, &off, &arr);
    if (k == 0 || first < 0 || first + count > b->n) return DMX_EINVAL;
    std::vector<double> &a = W(b)->*arr;
    const int stride = arr == &HostWorld::st ? SR : k;
    for (int64_t i = 0; i < count; i++) memcpy(&a[(size_t)(first + i) * stride + off], (const double *)src + i * k, k * sizeof(double));
    return DMX_OK;
}
extern "C" int dmxBatchDownload(dmxBatchID b, int field, void *dst, int64_t first, int64_t count)
{
    int off; std::vector<double> HostWorld::*arr;
    const int k = field_span(field, &off, &arr);
    if (k == 0 || first < 0 || first + count > b->n) return DMX_EINVAL;
    std::vector<double> &a = W(b)->*arr;
    const int stride = arr == &HostWorld::st ? SR : k;
    for (int64_t i = 0; i < count; i++) memcpy((double *)dst + i * k, &a[(size_t)(first + i) * stride + off], k * sizeof(double));
    return DMX_OK;
}
extern "C" int dmxBatchUploadGeomType(dmxBatchID b, const uint8_t *src, int64_t first, int64_t count)
{
    if (first < 0 || first + count > b->n) return DMX_EINVAL;
    memcpy(&b->h_gtype[(size_t)first], src, (size_t)count);
    return DMX_OK;
}
extern "C" int dmxBatchSetActiveCount(dmxBatchID b, int64_t n_active)
{
    if (n_active < 0 || n_active > b->n) return DMX_EINVAL;
    b->n_active = n_active;
    return DMX_OK;
}
extern "C" int dmxBatchSetBoundaryPack(dmxBatchID b, void *out, int64_t lo, int64_t hi) { b->pack_out = out; b->pack_lo = lo; b->pack_hi = hi; return DMX_OK; }
extern "C" int dmxBatchGatherBodies(dmxBatchID b, const int32_t *idx, int64_t count, void *out)
{
    HostWorld *w = W(b);
    for (int64_t k = 0; k < count; k++) {
        if (idx[k] < 0 || idx[k] >= b->n) return DMX_EINVAL;
        memcpy((double *)out + k * SR, &w->st[(size_t)idx[k] * SR], SR * sizeof(double));
    }
    w->gathers++;
    return DMX_OK;
}
extern "C" int dmxBatchRefreshGhostsOnStream(dmxBatchID b, void *, int64_t first, int64_t count_lo, const void *src_lo, int64_t count_hi,
                                             const void *src_hi, int check)
{
    HostWorld *w = W(b);
    if (first < b->n_active || first + count_lo + count_hi > b->n) return DMX_EINVAL;
    if (src_lo) memcpy(&w->st[(size_t)first * SR], src_lo, (size_t)count_lo * SR * sizeof(double));
    if (src_hi) memcpy(&w->st[(size_t)(first + count_lo) * SR], src_hi, (size_t)count_hi * SR * sizeof(double));
    w->ghost_refreshes++;
    if (check) {
        if (src_lo) zone_test(b, w, first, count_lo);
        if (src_hi) zone_test(b, w, first + count_lo, count_hi);
        w->ghost_checks++;
    }
    return DMX_OK;
}
extern "C" int dmxBatchChunkBegin(dmxBatchID b, int *exact_only, int *ballistic)
{
    HostWorld *w = W(b);
    w->snap = w->st;
    for (int64_t i = 0; i < b->n; i++) { w->zone[2 * (size_t)i] = w->st[(size_t)i * SR]; w->zone[2 * (size_t)i + 1] = w->st[(size_t)i * SR + 2]; }
    w->violated = w->warn = false;
    *exact_only = w->exact_only_chunks > 0 ? 1 : 0;
    if (w->exact_only_chunks > 0) w->exact_only_chunks--;
    *ballistic = w->ballistic;
    w->begins++;
    return DMX_OK;
}
extern "C" int dmxBatchChunkTick(dmxBatchID b, double h, int check)
{
    HostWorld *w = W(b);
    advance(b, w, h);
    if (check) { zone_test(b, w, 0, b->n_active); w->zone_checks++; }
    pack(b, w);
    w->fast_ticks++;
    return DMX_OK;
}
extern "C" int dmxBatchChunkTicks(dmxBatchID b, double h, int nticks, int check_first, int check_last)
{
    for (int t = 0; t < nticks; t++) {
        const int rc = dmxBatchChunkTick(b, h, (t == 0 && check_first) || (t == nticks - 1 && check_last));
        if (rc != DMX_OK) return rc;
    }
    return DMX_OK;
}
extern "C" int dmxBatchCheckZonesOnStream(dmxBatchID b, void *, int64_t first, int64_t count)
{
    HostWorld *w = W(b);
    if (first < 0 || first + count > b->n) return DMX_EINVAL;
    zone_test(b, w, first, count);
    w->zone_checks++;
    return DMX_OK;
}
extern "C" int dmxBatchChunkEnd(dmxBatchID b, int *violated, int *warn) { *violated = W(b)->violated; *warn = W(b)->warn; return DMX_OK; }
extern "C" int dmxBatchChunkCommit(dmxBatchID b, int ticks, int) { W(b)->commits++; W(b)->committed_ticks += ticks; return DMX_OK; }
extern "C" int dmxBatchChunkRollback(dmxBatchID b)
{
    HostWorld *w = W(b);
    if (w->snap.size() != w->st.size()) return DMX_EINVAL;
    w->st = w->snap;
    w->rollbacks++;
    return DMX_OK;
}
extern "C" int dmxBatchExactTick(dmxBatchID b, double h)
{
    HostWorld *w = W(b);
    advance(b, w, h);
    w->exact_ticks++;
    return DMX_OK;
}
extern "C" int dmxBatchFindPairs(dmxBatchID b, const int32_t **pairs, int64_t *n_pairs, const int32_t **inv, int64_t *n_inv)
{
    HostWorld *w = W(b);
    b->fp_cross.swap(w->cross);          // the scripted cross pairs are met once
    w->cross.clear();
    *pairs = w->no_pairs.data(); *n_pairs = 0; *inv = w->no_pairs.data(); *n_inv = 0;
    return DMX_OK;
}
extern "C" int dmxBatchCrossPairs(dmxBatchID b, const int32_t **pairs, int64_t *n_pairs)
{
    *pairs = b->fp_cross.data(); *n_pairs = (int64_t)b->fp_cross.size() / 2;
    return DMX_OK;
}

// ---- the HIP runtime entry points the rank loop uses: host memory, everything in order and done on return -----------------------
extern "C" {
hipError_t hipSetDevice(int) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "host stub"; }
hipError_t hipMalloc(void **p, size_t bytes) { *p = calloc(bytes ? bytes : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned int) { return hipMalloc(p, bytes); }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int) { *s = (hipStream_t)calloc(1, 8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free((void *)s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)calloc(1, 8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free((void *)e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
}
