// dmx_batch.cpp -- the batch C ABI declared in include/dmx_batch.h.
//
// Host side only: owns the HBM slab, the stream and the world parameters, and
// enqueues the kernels of dmx_kernels.hip.  There is no CPU execution path:
// without a HIP device every entry point fails with DMX_ENODEVICE.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <algorithm>
#include <cmath>
#include <vector>

#include "dmx_batch_priv.hpp"
#include "dmx_lcp.hpp"

// a chunk dmxBatchStep left open is closed (flag read, rollback + replay if need be) before anything observes or changes the batch
#define SETTLE(b)                                  \
    do {                                           \
        const int rc_settle_ = dmx_settle(b);      \
        if (rc_settle_ != DMX_OK) return rc_settle_; \
    } while (0)

static const int k_field_comp0[DMX_NFIELDS] = { C_POS, C_QUAT, C_LVEL, C_AVEL, C_MASS, C_INERTIA, C_SIDES, C_FORCE, C_TORQUE, C_QUAT, C_POS };
static const int k_field_k[DMX_NFIELDS] = { 3, 4, 3, 3, 1, 3, 3, 3, 3, 4, C_MASS };
static_assert(C_POS == 0 && C_QUAT == 3 && C_LVEL == 7 && C_AVEL == 10 && C_MASS == 13, "DMX_STATE is components 0..12");

extern "C" const char *dmxVersion(void) { return "libode_mi355 0.1 (gfx950)"; }

extern "C" int dmxDeviceCount(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return DMX_ENODEVICE;
    return n;
}

static int ensure_stage(dmxBatch *b, size_t bytes)
{
    if (bytes <= b->stage_bytes) return DMX_OK;
    if (b->stage) HIP_TRY(hipFree(b->stage));
    b->stage = nullptr; b->stage_bytes = 0;
    HIP_TRY(hipMalloc(&b->stage, bytes));
    b->stage_bytes = bytes;
    return DMX_OK;
}

static_assert(DMX_SLAB_TILE == SLAB_TILE && DMX_SLAB_COMPONENTS == C_COUNT, "include/dmx_batch.h documents the slab tiling");

template <class T> static int fill_defaults(dmxBatch *b)
{
    // every slot, pad included: mass 1, inertia 1 (dBodyCreate default), q = identity
    const int ones[] = { C_QUAT, C_MASS, C_INERTIA, C_INERTIA + 1, C_INERTIA + 2 };
    for (void *slab : { b->slab, b->slab_alt }) {
        HIP_TRY(hipMemsetAsync(slab, 0, (size_t)C_COUNT * b->stride * sizeof(T), b->stream));
        for (int c : ones)
            HIP_TRY(launch_fill_component<T>((T *)slab, c, T(1), b->stride, b->stream));
    }
    HIP_TRY(hipMemsetAsync(b->gtype, 0, (size_t)b->stride, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return DMX_OK;
}

extern "C" int dmxBatchCreate(dmxBatchID *out, int64_t n, int precision, int device)
{
    if (!out || n <= 0 || (precision != DMX_F32 && precision != DMX_F64)) return DMX_EINVAL;
    *out = nullptr;
    int ndev = dmxDeviceCount();
    if (ndev < 0) {
        fprintf(stderr, "libode_mi355: no HIP device available; this library has no CPU path\n");
        return DMX_ENODEVICE;
    }
    if (device < 0 || device >= ndev) return DMX_EINVAL;
    HIP_TRY(hipSetDevice(device));
    dmxBatch *b = new (std::nothrow) dmxBatch();
    if (!b) return DMX_ENOMEM;
    b->n = n;
    b->n_active = n;
    b->h_sides.assign((size_t)3 * n, 0.0);
    b->h_gtype.assign((size_t)n, 0);
    b->stride = (n + 255) / 256 * 256;
    b->precision = precision;
    b->device = device;
    b->rsize = precision == DMX_F32 ? 4 : 8;
    b->cfm = precision == DMX_F32 ? 1e-5 : 1e-10;      // dWorldCreate default per precision [ODE]
    if (const char *v = getenv("DMX_VEC")) b->vec = atoi(v);
    if (const char *v = getenv("DMX_WIDE")) {            // 1: integrate_free_wide; 2: integrate_free_dma (DMX_WIDE_BLOCKS per CU, default 4)
        if (atoi(v) == 1) b->vec = -16;
        if (atoi(v) == 2) { const char *k = getenv("DMX_WIDE_BLOCKS"); const int pc = k ? atoi(k) : 4; b->vec = -(32 + (pc > 0 && pc < 32 ? pc : 4)); }
    }
    if (const char *v = getenv("DMX_MIN_WAVES")) b->min_waves = atoi(v);
    if (const char *v = getenv("DMX_NT")) b->nt = atoi(v);
    if (const char *v = getenv("DMX_OOP")) b->oop = atoi(v);
    b->prof_on = getenv("DMX_HOST_PROFILE") != nullptr;
    {
        // every translation unit's code object now, not at the first tick that launches one of its kernels (DMX_PRELOAD=0: lazily)
        static const bool preload = [] { const char *e = getenv("DMX_PRELOAD"); return !(e && atoi(e) == 0); }();
        if (preload) {
            const int rb = b->precision == DMX_F32 ? 4 : 8;
            (void)dmx::dmx_touch_kernels(rb); (void)dmx::dmx_touch_islands(rb); (void)dmx::dmx_touch_broadphase(rb);
            (void)dmx::dmx_touch_narrow(rb); (void)dmx::dmx_touch_exact(rb);
        }
    }
    if (const char *v = getenv("DMX_LAZY_CHUNKS")) b->lazy_chunks = atoi(v) != 0;
    if (const char *v = getenv("DMX_STATIC_FAST")) b->static_fast = atoi(v) != 0;
    int rc = DMX_OK;
    do {
        if (hipStreamCreateWithFlags(&b->own_stream, hipStreamNonBlocking) != hipSuccess) { rc = DMX_EHIP; break; }
        b->stream = b->own_stream;
        if (hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess) { rc = DMX_EHIP; break; }
        if (hipMalloc(&b->slab, (size_t)C_COUNT * b->stride * b->rsize) != hipSuccess) { rc = DMX_ENOMEM; break; }
        if (hipMalloc(&b->slab_alt, (size_t)C_COUNT * b->stride * b->rsize) != hipSuccess) { rc = DMX_ENOMEM; break; }
        if (hipMalloc((void **)&b->gtype, (size_t)b->stride) != hipSuccess) { rc = DMX_ENOMEM; break; }
        if (hipMalloc((void **)&b->bflags, (size_t)b->stride) != hipSuccess) { rc = DMX_ENOMEM; break; }
        b->h_bflags.assign((size_t)b->stride, 0);
        for (int64_t i = 0; i < n; i++) b->h_bflags[(size_t)i] = BF_ALIVE;
        if (hipMemcpy(b->bflags, b->h_bflags.data(), (size_t)b->stride, hipMemcpyHostToDevice) != hipSuccess) { rc = DMX_EHIP; break; }
        // one diagnostics slot per wave of the fused step (plain stores, summed on the host when asked),
        // plus one atomically accumulated slot for the island path
        b->n_diag = (size_t)(b->stride / 64) + 1;
        if (hipMalloc((void **)&b->diag, b->n_diag * sizeof(StepDiag)) != hipSuccess) { rc = DMX_ENOMEM; break; }
        if (hipMalloc((void **)&b->diag_isl, sizeof(StepDiag)) != hipSuccess) { rc = DMX_ENOMEM; break; }
        if (hipHostMalloc((void **)&b->diag_host, b->n_diag * sizeof(StepDiag)) != hipSuccess) { rc = DMX_ENOMEM; break; }
        if (hipMemset(b->diag, 0, b->n_diag * sizeof(StepDiag)) != hipSuccess) { rc = DMX_EHIP; break; }
        if (hipMemset(b->diag_isl, 0, sizeof(StepDiag)) != hipSuccess) { rc = DMX_EHIP; break; }
        rc = precision == DMX_F32 ? fill_defaults<float>(b) : fill_defaults<double>(b);
    } while (0);
    if (rc != DMX_OK) {
        fprintf(stderr, "libode_mi355: dmxBatchCreate failed (%d): %s\n", rc, hipGetErrorString(hipGetLastError()));
        dmxBatchDestroy(b);
        return rc;
    }
    *out = b;
    return DMX_OK;
}

extern "C" int dmxBatchDestroy(dmxBatchID b)
{
    if (!b) return DMX_EINVAL;
    (void)hipSetDevice(b->device);
    (void)dmx_settle(b);
    if (b->own_stream) (void)hipStreamSynchronize(b->own_stream);
    if (b->prof_on) {
        static const char *names[12] = { "exact tick up to the record", "  of it: waiting for the record", "safe-zone rebuilds", "fast chunks (sync loop)",
                                         "joints: canonical + union-find", "joints: level schedules", "joints: staging fill",
                                         "joints: upload + launch", "fused kernel for the rest", "", "", "" };
        fprintf(stderr, "libode_mi355 host profile (exact ticks: %lld):\n", (long long)b->stat_careful_ticks);
        for (int k = 0; k < 9; k++) fprintf(stderr, "  %-32s %9.3f ms total\n", names[k], b->prof[k] * 1e3);
    }
    if (b->exs_ticks > 0) {          // DMX_EXS_TIMING=1
        static const char *fn[9] = { "", "zero", "grid insert", "pair count", "scan", "pair write", "union", "flatten", "scan" };
        static const char *bn[9] = { "", "island keys", "block sort", "sorted lists", "scan", "island bounds", "fill + big flags", "scan", "level schedules" };
        fprintf(stderr, "libode_mi355 small-scene exact tick, stage averages over %ld ticks:\n", b->exs_ticks);
        for (int k = 1; k < 9; k++) fprintf(stderr, "  front %-18s %7.2f us\n", fn[k], b->exs_acc[k] / (double)b->exs_ticks / 100.0);
        for (int k = 1; k < 9; k++) fprintf(stderr, "  back  %-18s %7.2f us\n", bn[k], b->exs_acc[32 + k] / (double)b->exs_ticks / 100.0);
    }
    dmx::lcp_grid_free(b);
    if (b->slab) (void)hipFree(b->slab);
    if (b->slab_alt) (void)hipFree(b->slab_alt);
    if (b->gtype) (void)hipFree(b->gtype);
    if (b->bflags) (void)hipFree(b->bflags);
    for (dmxBatch::DevBuf *d : { &b->jd_int, &b->jd_real, &b->jd_rows, &b->jd_rowjb, &b->jd_bscr, &b->jd_local, &b->jd_lcp, &b->jd_lcp_off,
                                &b->jd_lcp_int, &b->jd_order })
        if (d->p) (void)hipFree(d->p);
    for (dmxBatch::DevBuf *d : { &b->bp_count, &b->bp_items, &b->bp_flags, &b->bp_inpair, &b->bp_snapshot, &b->hull, &b->cbuf, &b->ccount,
                                &b->ex_arena, &b->ex_body, &b->ex_last, &b->ex_aabb, &b->sbox, &b->hull_planes })
        if (d->p) (void)hipFree(d->p);
    if (b->bp_flags_host) (void)hipHostFree(b->bp_flags_host);
    if (b->ex_counts_host) (void)hipHostFree(b->ex_counts_host);
    if (b->jh_int) (void)hipHostFree(b->jh_int);
    if (b->jh_real) (void)hipHostFree(b->jh_real);
    if (b->diag) (void)hipFree(b->diag);
    if (b->diag_isl) (void)hipFree(b->diag_isl);
    if (b->diag_host) (void)hipHostFree(b->diag_host);
    if (b->stage) (void)hipFree(b->stage);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    if (b->fork_ev) (void)hipEventDestroy(b->fork_ev);
    if (b->join_ev) (void)hipEventDestroy(b->join_ev);
    if (b->fork_stream) (void)hipStreamDestroy(b->fork_stream);
    if (b->own_stream) (void)hipStreamDestroy(b->own_stream);
    delete b;
    return DMX_OK;
}

extern "C" int64_t dmxBatchBodyCount(dmxBatchID b) { return b ? b->n : DMX_EINVAL; }
extern "C" int dmxBatchPrecision(dmxBatchID b) { return b ? b->precision : DMX_EINVAL; }
extern "C" int64_t dmxBatchStride(dmxBatchID b) { return b ? b->stride : DMX_EINVAL; }

extern "C" int dmxBatchSetGravity(dmxBatchID b, double x, double y, double z)
{ if (!b) return DMX_EINVAL; SETTLE(b); b->g[0] = x; b->g[1] = y; b->g[2] = z; return DMX_OK; }
extern "C" int dmxBatchSetERP(dmxBatchID b, double erp) { if (!b) return DMX_EINVAL; SETTLE(b); b->erp = erp; return DMX_OK; }
extern "C" int dmxBatchSetCFM(dmxBatchID b, double cfm) { if (!b) return DMX_EINVAL; SETTLE(b); b->cfm = cfm; return DMX_OK; }
extern "C" int dmxBatchSetQuickStep(dmxBatchID b, int iters, double w)
{ if (!b || iters < 0) return DMX_EINVAL; SETTLE(b); b->iters = iters; b->sor_w = w; return DMX_OK; }
extern "C" int dmxBatchSetGyroMode(dmxBatchID b, int mode)
{ if (!b || mode < 0 || mode > 2) return DMX_EINVAL; SETTLE(b); b->gyro = mode; return DMX_OK; }
extern "C" int dmxBatchSetSurface(dmxBatchID b, int mode, double mu, double bounce, double bounce_vel)
{
    if (!b) return DMX_EINVAL;
    SETTLE(b);
    b->surf_mode = mode; b->mu = mu < 0 ? 0 : mu; b->bounce = bounce; b->bounce_vel = bounce_vel;
    return DMX_OK;
}
extern "C" int dmxBatchSetMaxContacts(dmxBatchID b, int m)
{ if (!b || m < 1) return DMX_EINVAL; SETTLE(b); b->max_contacts = m; return DMX_OK; }

extern "C" int dmxBatchSetPlane(dmxBatchID b, double a, double bb, double c, double d, int enable)
{
    if (!b) return DMX_EINVAL;
    SETTLE(b);
    b->plane[0] = a; b->plane[1] = bb; b->plane[2] = c; b->plane[3] = d;
    b->plane_on = enable ? 1 : 0;
    return DMX_OK;
}

// ---- upload / download -------------------------------------------------------------------------
template <class T>
static int upload_t(dmxBatch *b, int field, const void *host, int64_t first, int64_t count)
{
    const int k = k_field_k[field];
    const size_t bytes = (size_t)count * k * sizeof(T);
    int rc = ensure_stage(b, bytes);
    if (rc != DMX_OK) return rc;
    const T *src = (const T *)host;
    std::vector<T> tmp;
    if (field == DMX_QUAT) {
        // dBodySetQuaternion: store the normalised quaternion
        tmp.assign(src, src + (size_t)count * 4);
        for (int64_t i = 0; i < count; i++) {
            Q4<T> q = { tmp[4 * i], tmp[4 * i + 1], tmp[4 * i + 2], tmp[4 * i + 3] };
            normalize(q);
            tmp[4 * i] = q.w; tmp[4 * i + 1] = q.x; tmp[4 * i + 2] = q.y; tmp[4 * i + 3] = q.z;
        }
        src = tmp.data();
    }
    HIP_TRY(hipMemcpyAsync(b->stage, src, bytes, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(launch_aos_to_soa<T>((T *)b->slab, b->stride, k_field_comp0[field], k, first, count,
                                 (const T *)b->stage, b->stream));
    if (field == DMX_MASS || field == DMX_INERTIA || field == DMX_SIDES)       // constants live in both slabs
        HIP_TRY(launch_aos_to_soa<T>((T *)b->slab_alt, b->stride, k_field_comp0[field], k, first, count,
                                     (const T *)b->stage, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));   // host buffer (and tmp) may be released on return
    if (field == DMX_FORCE || field == DMX_TORQUE) b->ext_pending = true;
    if (field == DMX_SIDES) {
        const T *p = (const T *)host;
        for (int64_t i = 0; i < count * 3; i++) b->h_sides[(size_t)(3 * first + i)] = (double)p[i];
        b->bp_rmax = 0;
    }
    if (field == DMX_POS || field == DMX_SIDES || field == DMX_STATE) b->bp_valid = false;     // poses / extents changed under the safe zones
    return DMX_OK;
}

template <class T>
static int download_t(dmxBatch *b, int field, void *host, int64_t first, int64_t count)
{
    const int k = k_field_k[field];
    const size_t bytes = (size_t)count * k * sizeof(T);
    int rc = ensure_stage(b, bytes);
    if (rc != DMX_OK) return rc;
    HIP_TRY(launch_soa_to_aos<T>((const T *)b->slab, b->stride, k_field_comp0[field], k, first, count,
                                 (T *)b->stage, b->stream));
    HIP_TRY(hipMemcpyAsync(host, b->stage, bytes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return DMX_OK;
}

static bool range_ok(dmxBatch *b, int field, const void *p, int64_t first, int64_t count)
{
    return b && p && field >= 0 && field < DMX_NFIELDS && first >= 0 && count >= 0 && first + count <= b->n;
}

extern "C" int dmxBatchUpload(dmxBatchID b, int field, const void *host, int64_t first, int64_t count)
{
    if (!range_ok(b, field, host, first, count)) return DMX_EINVAL;
    SETTLE(b);
    if (count == 0) return DMX_OK;
    HIP_TRY(hipSetDevice(b->device));
    return b->precision == DMX_F32 ? upload_t<float>(b, field, host, first, count)
                                   : upload_t<double>(b, field, host, first, count);
}

extern "C" int dmxBatchDownload(dmxBatchID b, int field, void *host, int64_t first, int64_t count)
{
    if (!range_ok(b, field, host, first, count)) return DMX_EINVAL;
    SETTLE(b);
    if (count == 0) return DMX_OK;
    HIP_TRY(hipSetDevice(b->device));
    return b->precision == DMX_F32 ? download_t<float>(b, field, host, first, count)
                                   : download_t<double>(b, field, host, first, count);
}

extern "C" int dmxBatchUploadGeomType(dmxBatchID b, const uint8_t *types, int64_t first, int64_t count)
{
    if (!b || !types || first < 0 || count < 0 || first + count > b->n) return DMX_EINVAL;
    SETTLE(b);
    if (count == 0) return DMX_OK;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipMemcpyAsync(b->gtype + first, types, (size_t)count, hipMemcpyHostToDevice, b->stream));
    // the count of box / sphere slots, kept up to date by what this range loses and gains (not by a pass over all n slots: the
    // sharded loop's migrate() calls this two or three times per adopted body inside the tick loop, on batches of up to 16 Mi slots)
    for (int64_t i = 0; i < count; i++) {
        const uint8_t was = b->h_gtype[(size_t)(first + i)], is = types[i];
        b->n_simple -= (was == GEOM_BOX || was == GEOM_SPHERE) ? 1 : 0;
        b->n_simple += (is == GEOM_BOX || is == GEOM_SPHERE) ? 1 : 0;
    }
    memcpy(b->h_gtype.data() + first, types, (size_t)count);
    if (b->scount.p)        // a slot that changes class must not keep its old class's contact count (np_static / np_convex_static
                            // write the counts of the slots they serve; nobody serves a slot of no class in a hulls-only batch)
        HIP_TRY(hipMemsetAsync((int *)b->scount.p + first, 0, (size_t)count * sizeof(int), b->stream));
    b->has_simple = b->n_simple > 0;
    b->bp_rmax = 0; b->bp_valid = false;
    HIP_TRY(hipStreamSynchronize(b->stream));
    return DMX_OK;
}

extern "C" int dmxBatchUploadBodyFlags(dmxBatchID b, const uint8_t *flags, int64_t first, int64_t count)
{
    if (!b || !flags || first < 0 || count < 0 || first + count > b->n) return DMX_EINVAL;
    SETTLE(b);
    if (count == 0) return DMX_OK;
    HIP_TRY(hipSetDevice(b->device));
    for (int64_t i = 0; i < count; i++) b->h_bflags[(size_t)(first + i)] = flags[i];
    HIP_TRY(hipMemcpyAsync(b->bflags + first, b->h_bflags.data() + first, (size_t)count, hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return DMX_OK;
}

int dmx_ensure_dev(dmxBatch::DevBuf &d, size_t bytes)
{
    if (bytes <= d.bytes) return DMX_OK;
    if (d.p) HIP_TRY(hipFree(d.p));
    d.p = nullptr; d.bytes = 0;
    size_t want = bytes + bytes / 2 + 256;
    HIP_TRY(hipMalloc(&d.p, want));
    d.bytes = want;
    return DMX_OK;
}

extern "C" void *dmxBatchDevicePtr(dmxBatchID b, int field, int component)
{
    if (!b || field < 0 || field >= DMX_NFIELDS || component < 0 || component >= k_field_k[field]) return nullptr;
    if (dmx_settle(b) != DMX_OK) return nullptr;
    return (char *)b->slab + (size_t)slab_ix(k_field_comp0[field] + component, 0) * b->rsize;
}

// ---- stepping ----------------------------------------------------------------------------------
template <class T> static int step_t(dmxBatch *b, double h, int nsteps, int64_t first, int64_t count, bool reset_diag)
{
    StepParams<T> P = dmx_make_params<T>(b, h);
    if (first != 0) P.pack_out = nullptr;      // the boundary pack is defined on whole-slab launches only
    if (P.cbuf != nullptr) { P.cbuf += (size_t)first * CONVEX_MAXC * 4; P.ccount += first; }   // per-body contact slots follow the range
    if (P.sbuf != nullptr) { P.sbuf += sc_ix(0, 0, first); P.scount += first; }                 // (first is a multiple of the tile)
    // contact-free ticks go ticks_per_launch at a time (state in registers between ticks); the first tick alone when
    // external force accumulators are pending (they act once)
    const int per = (b->plane_on || b->n_static > 0 || first != 0) ? 1 : std::max(1, b->ticks_per_launch);
    for (int s = 0; s < nsteps;) {
        (void)reset_diag;   // every wave overwrites its own slot each tick: nothing to clear
        const bool ext = b->ext_pending && s == 0;
        P.ticks = ext ? 1 : std::min(per, nsteps - s);
        T *S = (T *)b->slab + slab_ix(0, first);
        if (b->oop && first == 0 && count == b->n && !ext) {
            // experiment (DMX_OOP): read one slab, write the other, swap -- every tick
            HIP_TRY(launch_step<T>(S, (T *)b->slab_alt, b->gtype, b->stride, count, P, false, b->diag, b->stream));
            std::swap(b->slab, b->slab_alt);
        } else {
            HIP_TRY(launch_step<T>(S, S, b->gtype + first, b->stride, count, P, ext, b->diag + first / 64, b->stream));
        }
        s += P.ticks;
    }
    b->stepped_with_plane = dmx_fused_contacts(b);
    b->last_islands = false;
    return DMX_OK;
}

extern "C" int dmxBatchStep(dmxBatchID b, double h, int nsteps)
{
    if (!b || !(h > 0) || nsteps < 0) return DMX_EINVAL;
    HIP_TRY(hipSetDevice(b->device));
    if (b->bp_enabled) return dmx_step_collide(b, h, nsteps);
    SETTLE(b);
    int rc = b->precision == DMX_F32 ? step_t<float>(b, h, nsteps, 0, b->n_active, true)
                                     : step_t<double>(b, h, nsteps, 0, b->n_active, true);
    b->ext_pending = false;   // the step cleared the accumulators
    b->last_mixed = false;
    return rc;
}

extern "C" int dmxBatchSetStepper(dmxBatchID b, int stepper)
{
    if (!b || (stepper != DMX_STEPPER_QUICK && stepper != DMX_STEPPER_EXACT)) return DMX_EINVAL;
    SETTLE(b);
    b->stepper_exact = stepper == DMX_STEPPER_EXACT;
    return DMX_OK;
}

extern "C" int dmxBatchLcpStats(dmxBatchID b, int64_t out[8])
{
    if (!b || !out) return DMX_EINVAL;
    dmx::lcp_grid_stats(b, out);
    return DMX_OK;
}

extern "C" int dmxBatchSetRowOrder(dmxBatchID b, int order, uint32_t seed)
{
    if (!b || (order != DMX_ORDER_CREATION && order != DMX_ORDER_ODE)) return DMX_EINVAL;
    SETTLE(b);
    b->row_order_ode = order == DMX_ORDER_ODE;
    b->ode_rand = seed;
    return DMX_OK;
}

extern "C" int dmxBatchSetSnapshotMode(dmxBatchID b, int mode)
{
    if (!b || (mode != DMX_SNAPSHOT_PINGPONG && mode != DMX_SNAPSHOT_COPY)) return DMX_EINVAL;
    SETTLE(b);
    if (b->flipped) return DMX_EINVAL;          // not inside a chunk that has already advanced
    b->flip_armed = false;
    b->snapshot_mode = mode;
    return DMX_OK;
}

extern "C" int dmxBatchSetExactPipeline(dmxBatchID b, int mode)
{
    if (!b || mode < DMX_EXACT_AUTO || mode > DMX_EXACT_ONE_WORKGROUP) return DMX_EINVAL;
    SETTLE(b);
    b->exact_pipeline = mode;
    return DMX_OK;
}

extern "C" int dmxBatchSetClassPairs(dmxBatchID b, int class_a, int class_b, int enable)
{
    if (!b || class_a < DMX_GEOM_SPHERE || class_a > DMX_GEOM_CONVEX || class_b < DMX_GEOM_SPHERE || class_b > DMX_GEOM_CONVEX) return DMX_EINVAL;
    SETTLE(b);
    const uint32_t bits = (1u << (4 * class_a + class_b)) | (1u << (4 * class_b + class_a));
    b->class_pairs = enable ? (b->class_pairs | bits) : (b->class_pairs & ~bits);
    b->bp_valid = false;            // zones only keep apart what can collide
    return DMX_OK;
}

extern "C" int dmxBatchSetStaticPath(dmxBatchID b, int mode)
{
    if (!b || (mode != DMX_STATIC_EXACT && mode != DMX_STATIC_FUSED)) return DMX_EINVAL;
    SETTLE(b);
    b->static_fast = mode == DMX_STATIC_FUSED;
    b->bp_valid = false;            // who counts as crowded depends on it
    return DMX_OK;
}

extern "C" int dmxBatchSetBodyCollisions(dmxBatchID b, int enable)
{
    if (!b) return DMX_EINVAL;
    SETTLE(b);
    b->bp_enabled = enable ? 1 : 0;
    b->bp_valid = false;
    return DMX_OK;
}

// ---- static box geoms (AddBodyMap, main.c:735-761) -------------------------------------------------
template <class T> static int set_static_boxes_t(dmxBatch *b, int32_t n, const double *sides, const double *pos, const double *rot)
{
    std::vector<T> h((size_t)n * SBOX_REALS, T(0));
    for (int32_t s = 0; s < n; s++) {
        T *o = h.data() + (size_t)s * SBOX_REALS;
        T R[3][3], side[3];
        for (int a = 0; a < 3; a++) {
            o[SBOX_POS + a] = (T)pos[3 * s + a];
            side[a] = o[SBOX_SIDE + a] = (T)sides[3 * s + a];
            for (int c = 0; c < 3; c++) R[a][c] = o[SBOX_R + 3 * a + c] = (T)rot[12 * s + 4 * a + c];
        }
        for (int a = 0; a < 3; a++) {           // the geom's AABB: centre +- sum_j |R_aj| side_j / 2 [ODE dxBox::computeAABB]
            const T r = T(0.5) * (tabs(R[a][0] * side[0]) + tabs(R[a][1] * side[1]) + tabs(R[a][2] * side[2]));
            o[SBOX_LO + a] = o[SBOX_POS + a] - r;
            o[SBOX_HI + a] = o[SBOX_POS + a] + r;
        }
    }
    int rc;
    if (n > 0) {
        if ((rc = dmx_ensure_dev(b->sbox, h.size() * sizeof(T))) != DMX_OK) return rc;
        HIP_TRY(hipMemcpyAsync(b->sbox.p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        // the fused path's contact buffer: SC_MAXC contacts of SC_REALS reals per slot, tiled like the slab
        if ((rc = dmx_ensure_dev(b->sbuf, (size_t)b->stride * SC_MAXC * SC_REALS * sizeof(T))) != DMX_OK) return rc;
        if ((rc = dmx_ensure_dev(b->scount, (size_t)b->stride * sizeof(int))) != DMX_OK) return rc;
        HIP_TRY(hipMemset(b->scount.p, 0, (size_t)b->stride * sizeof(int)));
    }
    b->n_static = n;
    b->bp_valid = false;
    return DMX_OK;
}

extern "C" int dmxBatchSetStaticBoxes(dmxBatchID b, int32_t n, const double *sides, const double *pos, const double *rot3x4)
{
    if (!b || n < 0 || n > DMX_MAX_STATIC_BOXES || (n > 0 && (!sides || !pos || !rot3x4))) return DMX_EINVAL;
    SETTLE(b);
    HIP_TRY(hipSetDevice(b->device));
    return b->precision == DMX_F32 ? set_static_boxes_t<float>(b, n, sides, pos, rot3x4)
                                   : set_static_boxes_t<double>(b, n, sides, pos, rot3x4);
}

// ---- convex bodies ------------------------------------------------------------------------------
extern "C" int dmxBatchSetConvexHull(dmxBatchID b, int32_t n_points, const double *points_xyz, double *radius_out)
{
    if (!b || n_points < 4 || !points_xyz) return DMX_EINVAL;
    SETTLE(b);
    HIP_TRY(hipSetDevice(b->device));
    int rc;
    if ((rc = dmx_ensure_dev(b->hull, (size_t)n_points * 3 * b->rsize)) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->cbuf, (size_t)b->stride * CONVEX_MAXC * 4 * b->rsize)) != DMX_OK) return rc;
    if ((rc = dmx_ensure_dev(b->ccount, (size_t)b->stride * sizeof(int))) != DMX_OK) return rc;
    double r2 = 0;
    for (int32_t i = 0; i < n_points; i++) {
        const double *p = points_xyz + 3 * (size_t)i;
        r2 = std::max(r2, p[0] * p[0] + p[1] * p[1] + p[2] * p[2]);
    }
    if (radius_out) *radius_out = std::sqrt(r2);
    HIP_TRY(hipStreamSynchronize(b->stream));
    if (b->precision == DMX_F32) {
        std::vector<float> f((size_t)n_points * 3);
        for (size_t i = 0; i < f.size(); i++) f[i] = (float)points_xyz[i];
        HIP_TRY(hipMemcpy(b->hull.p, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
    } else {
        HIP_TRY(hipMemcpy(b->hull.p, points_xyz, (size_t)n_points * 3 * sizeof(double), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemset(b->ccount.p, 0, (size_t)b->stride * sizeof(int)));
    b->hull_n = n_points;
    b->bp_valid = false;
    return DMX_OK;
}

extern "C" int dmxBatchSetConvexHullFaces(dmxBatchID b, int32_t n_faces, const double *planes)
{
    if (!b || n_faces < 0 || (n_faces > 0 && !planes)) return DMX_EINVAL;
    SETTLE(b);
    HIP_TRY(hipSetDevice(b->device));
    int rc;
    if (n_faces > 0) {
        if ((rc = dmx_ensure_dev(b->hull_planes, (size_t)n_faces * 4 * b->rsize)) != DMX_OK) return rc;
        HIP_TRY(hipStreamSynchronize(b->stream));
        if (b->precision == DMX_F32) {
            std::vector<float> f((size_t)n_faces * 4);
            for (size_t i = 0; i < f.size(); i++) f[i] = (float)planes[i];
            HIP_TRY(hipMemcpy(b->hull_planes.p, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
        } else {
            HIP_TRY(hipMemcpy(b->hull_planes.p, planes, (size_t)n_faces * 4 * sizeof(double), hipMemcpyHostToDevice));
        }
    }
    b->hull_nf = n_faces;
    return DMX_OK;
}

// ---- the collision-checked loop in pieces (include/dmx_batch.h) --------------------------------
extern "C" int dmxBatchChunkBegin(dmxBatchID b, int *exact_only, int *ballistic)
{
    if (!b || !exact_only || !ballistic) return DMX_EINVAL;
    SETTLE(b);
    HIP_TRY(hipSetDevice(b->device));
    return dmx_chunk_begin(b, exact_only, ballistic);
}
extern "C" int dmxBatchChunkTick(dmxBatchID b, double h, int check)
{
    if (!b || !(h > 0)) return DMX_EINVAL;
    HIP_TRY(hipSetDevice(b->device));
    return dmx_chunk_tick(b, h, check);
}
extern "C" int dmxBatchChunkTicks(dmxBatchID b, double h, int nticks, int check_first, int check_last)
{
    if (!b || !(h > 0) || nticks < 0) return DMX_EINVAL;
    HIP_TRY(hipSetDevice(b->device));
    return dmx_chunk_ticks(b, h, nticks, check_first, check_last);
}
extern "C" int dmxBatchSetTicksPerLaunch(dmxBatchID b, int ticks)
{
    if (!b || ticks < 1 || ticks > 64) return DMX_EINVAL;
    SETTLE(b);
    b->ticks_per_launch = ticks;
    return DMX_OK;
}
extern "C" int dmxBatchCheckZonesOnStream(dmxBatchID b, void *hip_stream, int64_t first, int64_t count)
{
    if (!b || first < 0 || count < 0 || first + count > b->n) return DMX_EINVAL;
    SETTLE(b);
    HIP_TRY(hipSetDevice(b->device));
    return dmx_check_zones(b, (hipStream_t)hip_stream, first, count);
}
extern "C" int dmxBatchRefreshGhostsOnStream(dmxBatchID b, void *hip_stream, int64_t first, int64_t count_lo,
                                            const void *src_lo, int64_t count_hi, const void *src_hi, int check)
{
    if (!b || first < b->n_active || count_lo < 0 || count_hi < 0 || first + count_lo + count_hi > b->n) return DMX_EINVAL;
    SETTLE(b);
    if (check && !b->bp_flags.p) return DMX_EINVAL;          // no chunk begun: there are no zones to test against
    HIP_TRY(hipSetDevice(b->device));
    uint32_t *flags = (uint32_t *)b->bp_flags.p;
    if (b->precision == DMX_F32)
        HIP_TRY(launch_refresh_ghosts<float>((float *)b->slab, first, count_lo, (const float *)src_lo, count_hi, (const float *)src_hi,
                                             check, flags, (hipStream_t)hip_stream));
    else
        HIP_TRY(launch_refresh_ghosts<double>((double *)b->slab, first, count_lo, (const double *)src_lo, count_hi, (const double *)src_hi,
                                              check, flags, (hipStream_t)hip_stream));
    return DMX_OK;
}
extern "C" int dmxBatchChunkEnd(dmxBatchID b, int *violated, int *warn)
{
    if (!b || !violated || !warn) return DMX_EINVAL;
    HIP_TRY(hipSetDevice(b->device));
    return dmx_chunk_end(b, violated, warn);
}
extern "C" int dmxBatchChunkCommit(dmxBatchID b, int ticks, int refresh_zones)
{
    if (!b || ticks < 0) return DMX_EINVAL;
    return dmx_chunk_commit(b, ticks, refresh_zones);
}
extern "C" int dmxBatchChunkRollback(dmxBatchID b)
{
    if (!b) return DMX_EINVAL;
    HIP_TRY(hipSetDevice(b->device));
    return dmx_chunk_rollback(b);
}
extern "C" int dmxBatchExactTick(dmxBatchID b, double h)
{
    if (!b || !(h > 0)) return DMX_EINVAL;
    SETTLE(b);
    HIP_TRY(hipSetDevice(b->device));
    return dmx_exact_tick(b, h);
}

extern "C" int dmxBatchFindPairs(dmxBatchID b, const int32_t **pairs, int64_t *n_pairs, const int32_t **involved, int64_t *n_involved)
{
    if (!b || !pairs || !n_pairs || !involved || !n_involved) return DMX_EINVAL;
    SETTLE(b);
    HIP_TRY(hipSetDevice(b->device));
    const int rc = dmx_find_pairs(b);
    if (rc != DMX_OK) return rc;
    *pairs = b->fp_pairs.data(); *n_pairs = (int64_t)(b->fp_pairs.size() / 2);
    *involved = b->fp_inv.data(); *n_involved = (int64_t)b->fp_inv.size();
    return DMX_OK;
}

extern "C" int dmxBatchCrossPairs(dmxBatchID b, const int32_t **pairs, int64_t *n_pairs)
{
    if (!b || !pairs || !n_pairs) return DMX_EINVAL;
    *pairs = b->fp_cross.data(); *n_pairs = (int64_t)(b->fp_cross.size() / 2);
    return DMX_OK;
}

extern "C" int dmxBatchCollisionStats(dmxBatchID b, int64_t out[6])
{
    if (!b || !out) return DMX_EINVAL;
    SETTLE(b);
    out[0] = b->stat_fast_ticks; out[1] = b->stat_careful_ticks; out[2] = b->stat_rebuilds;
    out[3] = b->stat_pair_ticks; out[4] = (int64_t)b->last_pairs; out[5] = (int64_t)b->bp_crowded;
    return DMX_OK;
}

extern "C" int dmxBatchCollisionStatsEx(dmxBatchID b, int64_t out[8])
{
    if (!b || !out) return DMX_EINVAL;
    int rc = dmxBatchCollisionStats(b, out);
    if (rc != DMX_OK) return rc;
    out[6] = b->stat_unsupported; out[7] = b->stat_spec_ticks;
    return DMX_OK;
}

extern "C" int dmxBatchSetBoundaryPack(dmxBatchID b, void *out_dev, int64_t lo_count, int64_t hi_first)
{
    if (!b || lo_count < 0 || hi_first < lo_count || hi_first > b->n_active) return DMX_EINVAL;
    SETTLE(b);
    b->pack_out = out_dev; b->pack_lo = lo_count; b->pack_hi = hi_first;
    return DMX_OK;
}

extern "C" int dmxBatchSetActiveCount(dmxBatchID b, int64_t n_active)
{
    if (!b || n_active < 0 || n_active > b->n) return DMX_EINVAL;
    SETTLE(b);
    if (n_active != b->n && n_active % 4 != 0) return DMX_EINVAL;   // the last 16 B pack must not reach into ghost slots
    b->n_active = n_active;
    return DMX_OK;
}

extern "C" int dmxBatchStepRange(dmxBatchID b, double h, int64_t first, int64_t count, int reset_diag)
{
    if (!b || !(h > 0) || first < 0 || count < 0 || first + count > b->n_active) return DMX_EINVAL;
    SETTLE(b);
    // lanes own 16 B packs of consecutive bodies: ranges must start on a pack and end on one (or at the end)
    const int64_t pack = 16 / (int64_t)b->rsize;
    if (first % pack != 0 || (count % pack != 0 && first + count != b->n_active)) return DMX_EINVAL;
    if (first % 64 != 0) return DMX_EINVAL;      // ranges start on a wave so per-wave diagnostics slots stay disjoint
    if (count == 0) return DMX_OK;
    HIP_TRY(hipSetDevice(b->device));
    return b->precision == DMX_F32 ? step_t<float>(b, h, 1, first, count, reset_diag != 0)
                                   : step_t<double>(b, h, 1, first, count, reset_diag != 0);
}

extern "C" int dmxBatchSynchronize(dmxBatchID b)
{
    if (!b) return DMX_EINVAL;
    SETTLE(b);
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return DMX_OK;
}

extern "C" int dmxBatchSetStream(dmxBatchID b, void *hip_stream)
{
    if (!b) return DMX_EINVAL;
    SETTLE(b);
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipStreamSynchronize(b->stream));
    b->stream = hip_stream ? (hipStream_t)hip_stream : b->own_stream;
    return DMX_OK;
}

extern "C" int dmxBatchGetStream(dmxBatchID b, void **hip_stream)
{
    if (!b || !hip_stream) return DMX_EINVAL;
    *hip_stream = (void *)b->stream;
    return DMX_OK;
}

extern "C" int dmxBatchStepTimed(dmxBatchID b, double h, int nsteps, float *ms)
{
    if (!b || !ms) return DMX_EINVAL;
    HIP_TRY(hipSetDevice(b->device));
    HIP_TRY(hipEventRecord(b->ev0, b->stream));
    int rc = dmxBatchStep(b, h, nsteps);
    if (rc != DMX_OK) return rc;
    HIP_TRY(hipEventRecord(b->ev1, b->stream));
    HIP_TRY(hipEventSynchronize(b->ev1));
    HIP_TRY(hipEventElapsedTime(ms, b->ev0, b->ev1));
    return DMX_OK;
}

static int fetch_diag(dmxBatch *b, unsigned long long *contacts, double *residual)
{
    HIP_TRY(hipSetDevice(b->device));
    if (b->last_islands) {
        HIP_TRY(hipMemcpyAsync(b->diag_host, b->diag_isl, sizeof(StepDiag), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        *contacts = b->diag_host[0].contacts; *residual = b->diag_host[0].residual;
        return DMX_OK;
    }
    const size_t nw = (size_t)((b->n_active + 63) / 64);
    HIP_TRY(hipMemcpyAsync(b->diag_host, b->diag, nw * sizeof(StepDiag), hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    unsigned long long c = 0; double r = 0;
    for (size_t i = 0; i < nw; i++) { c += b->diag_host[i].contacts; r += b->diag_host[i].residual; }
    if (b->last_mixed) {      // bodies in pairs went through the island kernel: add its tally
        HIP_TRY(hipMemcpyAsync(b->diag_host, b->diag_isl, sizeof(StepDiag), hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        c += b->diag_host[0].contacts; r += b->diag_host[0].residual;
    }
    *contacts = c; *residual = r;
    return DMX_OK;
}

extern "C" int dmxBatchLastContactCount(dmxBatchID b, int64_t *n)
{
    if (!b || !n) return DMX_EINVAL;
    SETTLE(b);
    if (!b->stepped_with_plane) { *n = 0; return DMX_OK; }
    unsigned long long c; double r;
    int rc = fetch_diag(b, &c, &r);
    if (rc != DMX_OK) return rc;
    *n = (int64_t)c;
    return DMX_OK;
}

extern "C" int dmxBatchLastResidual(dmxBatchID b, double *res)
{
    if (!b || !res) return DMX_EINVAL;
    SETTLE(b);
    if (!b->stepped_with_plane) { *res = 0; return DMX_OK; }
    unsigned long long c; double r;
    int rc = fetch_diag(b, &c, &r);
    if (rc != DMX_OK) return rc;
    *res = r;
    return DMX_OK;
}

// ---- pose snapshot -----------------------------------------------------------------------------
extern "C" int dmxBatchPackTransforms(dmxBatchID b, void *out_dev, int64_t first, int64_t count)
{
    if (!b || !out_dev || first < 0 || count < 0 || first + count > b->n) return DMX_EINVAL;
    SETTLE(b);
    HIP_TRY(hipSetDevice(b->device));
    if (b->precision == DMX_F32)
        HIP_TRY(launch_pack_transforms<float>((const float *)b->slab, b->stride, first, count, (float *)out_dev, b->stream));
    else
        HIP_TRY(launch_pack_transforms<double>((const double *)b->slab, b->stride, first, count, (double *)out_dev, b->stream));
    return DMX_OK;
}

extern "C" int dmxBatchDownloadTransforms(dmxBatchID b, void *out_host, int64_t first, int64_t count)
{
    if (!b || !out_host || first < 0 || count < 0 || first + count > b->n) return DMX_EINVAL;
    if (count == 0) return DMX_OK;
    HIP_TRY(hipSetDevice(b->device));
    const size_t bytes = (size_t)count * 16 * b->rsize;
    int rc = ensure_stage(b, bytes);
    if (rc != DMX_OK) return rc;
    rc = dmxBatchPackTransforms(b, b->stage, first, count);
    if (rc != DMX_OK) return rc;
    HIP_TRY(hipMemcpyAsync(out_host, b->stage, bytes, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(hipStreamSynchronize(b->stream));
    return DMX_OK;
}

// ---- boundary exchange -------------------------------------------------------------------------
extern "C" int dmxBatchGatherBodies(dmxBatchID b, const int32_t *idx_dev, int64_t count, void *out_dev)
{
    if (!b || count < 0 || (count > 0 && (!idx_dev || !out_dev))) return DMX_EINVAL;
    SETTLE(b);
    HIP_TRY(hipSetDevice(b->device));
    if (b->precision == DMX_F32)
        HIP_TRY(launch_gather<float>((const float *)b->slab, b->stride, idx_dev, count, (float *)out_dev, b->stream));
    else
        HIP_TRY(launch_gather<double>((const double *)b->slab, b->stride, idx_dev, count, (double *)out_dev, b->stream));
    return DMX_OK;
}

static int scatter_on(dmxBatch *b, const int32_t *idx_dev, int64_t count, const void *in_dev, hipStream_t st)
{
    if (!b || count < 0 || (count > 0 && (!idx_dev || !in_dev))) return DMX_EINVAL;
    SETTLE(b);
    b->bp_valid = false;
    HIP_TRY(hipSetDevice(b->device));
    if (b->precision == DMX_F32)
        HIP_TRY(launch_scatter<float>((float *)b->slab, b->stride, idx_dev, count, (const float *)in_dev, st));
    else
        HIP_TRY(launch_scatter<double>((double *)b->slab, b->stride, idx_dev, count, (const double *)in_dev, st));
    return DMX_OK;
}

extern "C" int dmxBatchScatterBodies(dmxBatchID b, const int32_t *idx_dev, int64_t count, const void *in_dev)
{
    return scatter_on(b, idx_dev, count, in_dev, b ? b->stream : nullptr);
}

// same, on a caller-chosen HIP stream: lets the boundary exchange write ghost slots from a side stream while the
// batch's own stream is already integrating the next tick (ghost slots are never touched by the step kernels)
extern "C" int dmxBatchScatterBodiesOnStream(dmxBatchID b, const int32_t *idx_dev, int64_t count, const void *in_dev,
                                             void *hip_stream)
{
    return scatter_on(b, idx_dev, count, in_dev, (hipStream_t)hip_stream);
}
