// dmx_exact.hpp -- interface of the device-resident bookkeeping of the exact tick (dmx_exact.hip), used by dmx_general.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dmx_internal.hpp"

namespace dmx {

// capacities the host sized the arrays for (estimates from earlier ticks): body pairs, involved bodies, rows of islands
// that get a workgroup.
struct ExactCaps {
    uint32_t pairs, inv, rows;
    uint32_t nstatic;       // static box geoms of the batch (not an estimate)
    // entries, in joint creation order: [0, inv) involved body k against the ground plane; then for every static box s the
    // block [inv (1 + s), inv (2 + s)): body k against static box s; then the body pairs
    __host__ __device__ uint32_t pair_entry0() const { return inv * (1u + nstatic); }
    __host__ __device__ uint32_t entries() const { return inv * (1u + nstatic) + pairs; }
    // contact slots: 8 per entry (a box yields at most 4 against the plane, a convex hull up to 8)
    __host__ __device__ size_t static_slot0() const { return (size_t)8 * inv; }
    __host__ __device__ size_t pair_slot0() const { return (size_t)8 * inv + (size_t)8 * nstatic * inv; }
    __host__ __device__ size_t slots() const { return pair_slot0() + (size_t)8 * pairs; }
};

// what the pipeline found, read back by the host once per tick (device struct, 80 B)
struct ExactCounts {
    uint32_t npairs, ninv, ni, njoints, nbig, big_rows, big_max_bodies, big_max_width;
    uint32_t overflow;      // bit 0: pairs / involved bodies above capacity; bit 1: level-schedule rows above capacity
    uint32_t cross;         // a pair reaches into a ghost slot (an island spanning two ranks); the pair:
    uint32_t cross_a, cross_b;
    uint32_t spec_ok;       // small-scene kernels: 1 = the island solve and the fused step enqueued BEHIND the bookkeeping may go ahead
                            // without the host (no overflow, no island across ranks, every island a one-wavefront job of
                            // solve_island_wg<64>); 0 = they return at once and the host launches what the counts call for
    uint32_t ncross;        // pairs (own body, ghost slot) met; the first EX_CROSS_CAP of them are in ExactBuffers::cross_list
    uint32_t bp_overflow;   // the grid's BPF_OVERFLOW flag as of the pair search (a bucket overflowed: the host widens them and searches
                            // again), carried here so that one read-back brings everything
    uint32_t big_max_rows;  // rows of the largest island that gets a workgroup (the launch shape of its solve: rows in registers or streamed)
    uint32_t pad_[3];
    uint32_t seq;           // small-scene kernels: the caller's sequence number, written to the host copy LAST: a host that watches
                            // for it has the whole record (and the flags) without waiting for the stream
};

constexpr uint32_t EX_CROSS_CAP = 256;
constexpr int EX_STAGE_PARTNERS = 32;      // (a body in a crowded pen has twenty AABB partners above it; past this a body walks again)
// the speculative island launch reserves LDS for islands of up to this many bodies (a workgroup's accumulators: 6 reals each);
// a tick with a larger island clears spec_ok and is launched by the host with the island's true size
constexpr uint32_t EX_SPEC_ISLAND_BODIES = 512;

template <class T> struct ExactBuffers {
    ExactCounts *counts;
    int32_t *cross_list;                        // [2 EX_CROSS_CAP] (own body, ghost slot): islands spanning two ranks
    void *temp; size_t temp_bytes;              // rocPRIM scratch
    uint64_t *pc, *inc;                         // [n_active] per body: (owned pairs << 32 | involved), and its inclusive scan
    uint8_t *inpair;                            // [stride] the fused kernel's skip mask
    int32_t *pairs;                             // [2 pairs]
    int32_t *inv, *parent, *root;               // [inv]
    uint32_t *rf, *rinc;                        // [inv] root flags, their inclusive scan
    T *gpos, *gnormal, *gdepth;                 // [3 / 3 / 1 per contact slot]
    uint32_t *cc;                               // [entries] contacts per entry
    uint32_t *keys, *vals, *keys_s, *vals_s;    // [entries]
    uint64_t *sc, *sinc;                        // [entries]
    int *body_off, *con_off, *row_off;          // [inv + 1]
    int *bodies;                                // [inv]
    int *cb1, *cb2, *csrc, *crow;               // [contact slots]
    uint64_t *bg, *binc;                        // [inv]
    int *big, *big_list, *lev_count;            // [inv]
    int *lev_off;                               // [rows + inv]
    int *lev_rows, *row_level;                  // [rows]
    int32_t *last;                              // [stride] per slot, -1 when idle
    uint64_t *stamps;                           // [64] stage time stamps of the small-scene kernels (100 MHz ticks)
    int32_t *stage;                             // optional [n_active EX_STAGE_PARTNERS]: the partners above a body, left by the count pass for the write pass
};

size_t exact_temp_bytes(const ExactCaps &cap, int64_t n_active);
hipError_t exact_init_last(int32_t *last, int64_t n, hipStream_t st);
// The pipeline on `st` in two parts; the grid G (with its records) must have been filled (bp_insert) and B.counts zeroed on
// the same stream before.  pairs: canonical pair list, involved bodies, union-find initialised, counts.npairs / ninv.  group: everything else.
template <class T>
hipError_t launch_exact_pairs(const T *S, const uint8_t *gtype, int64_t n_active, const GridParams<T> &G,
                              const ExactBuffers<T> &B, const ExactCaps &cap, hipStream_t st);
template <class T>
hipError_t launch_exact_group(const T *S, const uint8_t *gtype, int64_t n_active, const GridParams<T> &G, const StepParams<T> &P,
                              const ExactBuffers<T> &B, const ExactCaps &cap, int rpc, int big_rows, hipStream_t st);

// the record B.counts and the grid's flags into pinned host memory behind everything enqueued so far; the record's last word is `seq`
hipError_t launch_exact_publish(const ExactCounts *counts, const uint32_t *flags, ExactCounts *host_counts, uint32_t *host_flags, uint32_t seq,
                                hipStream_t st);

// Small scenes (exact_small_fits): the same pipeline as two one-workgroup kernels around the narrowphase.  front: also
// fills the grid (what fill_grid does); group: narrowphase + the rest, zeroes *diag.  A non-null host_counts / host_flags
// (device-visible pinned memory) receives ExactCounts and the BPF_* flags at the end of that kernel.
bool exact_small_fits(int64_t n, uint32_t grid_mask, const ExactCaps &cap);
// Many bodies, few of them involved: the body-sized stages as launches over the chip (fill_grid, launch_exact_pairs,
// launch_exact_roots), the entry-sized ones -- sort, joints by island, level schedules: a dozen launches of a few microseconds of work
// each -- as launch_exact_small_group's one-workgroup kernel, when the entry arrays fit it (exact_back_fits)
bool exact_back_fits(const ExactCaps &cap);
template <class T>
hipError_t launch_exact_roots(const ExactBuffers<T> &B, const ExactCaps &cap, hipStream_t st);
template <class T>
hipError_t launch_exact_small_front(T *S, const uint8_t *gtype, int64_t n, int64_t n_active, const GridParams<T> &G, const ExactBuffers<T> &B,
                                    const ExactCaps &cap, ExactCounts *host_counts, uint32_t *host_flags, uint32_t seq, hipStream_t st);
template <class T>
hipError_t launch_exact_small_group(const T *S, const uint8_t *gtype, const GridParams<T> &G, const StepParams<T> &P, const ExactBuffers<T> &B,
                                    const ExactCaps &cap, int rpc, int big_rows, StepDiag *diag, ExactCounts *host_counts,
                                    uint32_t *host_flags, uint32_t seq, int64_t n_slots, hipStream_t st);

// solve_island_wg<64> over `max_big` islands, enqueued behind launch_exact_small_group before the host has its counts: the
// workgroups ask the record on the device (counts_dev) whether they exist and whether the launch may act (ExactCounts::spec_ok)
template <class T>
hipError_t launch_islands_speculative(T *S, const uint8_t *bflags, int64_t stride, const IslandSet<T> &I, const StepParams<T> &P,
                                      StepDiag *diag, const ExactCounts *counts_dev, unsigned max_big, hipStream_t st);

// the same launch carrying the fused ground-plane step for everyone else too (Pf: skip mask, gate): see solve_islands_and_step
template <class T>
hipError_t launch_islands_and_step_speculative(T *S, const uint8_t *bflags, const uint8_t *gtype, int64_t stride, int64_t n, const IslandSet<T> &I,
                                               const StepParams<T> &P, const StepParams<T> &Pf, StepDiag *diag_isl, StepDiag *diag_fused,
                                               const ExactCounts *counts_dev, unsigned max_big, hipStream_t st);

}  // namespace dmx
