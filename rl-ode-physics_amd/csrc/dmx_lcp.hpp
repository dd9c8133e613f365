// dmx_lcp.hpp -- dWorldStep's exact island solve for LARGE islands, spread over the whole chip (dmx_lcp.hip).
// The reference calls dWorldStep at 120 Hz with up to 512 bodies that pile into ONE island of 2 000 - 2 600 constraint rows
// (/root/reference/src/main.c:208,213, /root/reference/inc/body.h:6): one workgroup per island (lcp_island_wg) is the wrong
// shape for that.  See the header of dmx_lcp.hip for the method.
#pragma once

#include <stdint.h>
#include <vector>

#include "dmx_batch_priv.hpp"

namespace dmx {

// rows at or above which an island's exact solve leaves lcp_island_wg (one workgroup) for the grid solve (DMX_LCP_GRID_ROWS)
int lcp_grid_threshold();
// islands above this many rows are not solved exactly at all: the tick falls back to QuickStep's SOR with a warning (DMX_MAX_EXACT_ROWS)
int lcp_max_exact_rows();

// one island handed to the grid solve: its index, its row count, and per row (island-relative, creation order) whether it is
// unbounded (lo = -inf, hi = +inf: never leaves the free set), whether its clamped value could be non-zero, and the key the
// active set is remembered under from tick to tick (0: none)
struct LcpIslandRows {
    int isl = 0, m = 0;
    int row_base = 0;                 // the island's first row in the set's flat row arrays (3 x its first contact)
    std::vector<uint8_t> unbounded;
    std::vector<uint64_t> key;
    // the rows of every body of the island, ascending: body k's (k = its index within the island) are bodyrows[boff[k] .. boff[k+1]),
    // each entry 2 * row + side (0: the body is the row's first, 1: its second)
    std::vector<int> boff, bodyrows;
};

// Solve island `R.isl` of the set exactly and step its bodies (the launch sequence is enqueued on b->stream; the host waits for
// each pivoting round's verdict).  Rows / body scratch of I are used as by the other island kernels.
template <class T>
int lcp_grid_solve(dmxBatch *b, const IslandSet<T> &I, const StepParams<T> &P, const LcpIslandRows &R);

// called once per exact tick, before the first / after the last lcp_grid_solve: rotates the remembered active sets
void lcp_grid_begin_tick(dmxBatch *b);
void lcp_grid_end_tick(dmxBatch *b);
void lcp_grid_free(dmxBatch *b);
// solves, pivoting rounds (total, most in one solve), last island's rows / unbounded / bounded, single-flip rounds, ticks that
// fell back to the SOR because an island exceeded lcp_max_exact_rows()
void lcp_grid_stats(dmxBatch *b, int64_t out[8]);
void lcp_grid_count_fallback(dmxBatch *b);
hipError_t dmx_touch_lcp(int real_bytes);
// dWorldStep's small and medium islands: one workgroup per island of I.big_list, the whole solve in LDS (lcp_island_lds).
// lcp_lds_fits: does an island of m rows, nbd of them bounded, fit?  (Islands that do not go to the grid solve.)
bool lcp_lds_fits(int real_bytes, int m, int nbd);
size_t lcp_lds_need(int real_bytes, int m, int nbd);
template <class T>
hipError_t launch_lcp_lds(T *S, const uint8_t *bflags, int64_t stride, const IslandSet<T> &I, const StepParams<T> &P, StepDiag *diag,
                          size_t lds_bytes, hipStream_t st);

}  // namespace dmx
