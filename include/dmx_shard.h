/*
 * include/dmx_shard.h -- one rank's tick loop of the island-sharded world, behind the C ABI (north_star:
 * "bodies shard across the 8 GPUs of one node as independent dynamics islands with a RCCL all-gather over xGMI of boundary
 * body state").  The reference has no multi-device stepping at all (its one physics loop is main.c:206-216); a C host that
 * wants it calls, per rank (one process per GPU):
 *
 *     dmxBatchCreate / Upload ...                     the rank's slab of the scene: rows x side bodies, row-major
 *     dmxShardRcclUniqueId(id)  on rank 0, handed to the other ranks by whatever the host program has (MPI, a socket, a file)
 *     dmxShardCreateRccl(&s, batch, side, rows, spare, rank, world, id)
 *     loop:  dmxShardRun(s, h, nticks);               replaces  dmxBatchStep  (main.c:211-215 in bulk)
 *            dmxShardSettle(s); dmxBatchDownload...    whenever poses are wanted
 *     dmxShardDestroy(s)
 *
 * Slab layout (the batch must have side * rows + spare + 2 * side + spare slots): [0, n) the rank's own bodies, n = side * rows,
 * row r = slots [r * side, (r + 1) * side), rows ordered along the sharding axis; [n, n + spare) empty slots that take bodies
 * adopted from the upper neighbour when an island spans the shared face; then `side` ghost slots for the lower neighbour's
 * last row, `side` for the upper neighbour's first row, and `spare` for the lower neighbour's spare slots -- a body that
 * neighbour adopted from this rank stays visible here as a ghost, so this rank's bodies behind the boundary row still see it:
 * they either follow it down (first-row bodies) or the contact is reported on every rank (DMX_ECROSS), never missed.
 * dmxShardCreate* calls dmxBatchSetActiveCount itself, shares the boundary rows' geometry (extents, classes, mass properties)
 * with the neighbours and primes the ghost slots.
 *
 * What a tick does is what rl-ode-physics_amd/shard.py documents (the same loop, which that module now binds): ticks run in
 * collision-proof chunks; the step kernel packs the boundary rows' new state itself; a side stream all-gathers them and one
 * kernel refreshes (and zone-tests) the ghost slots while the batch's stream integrates on; in a ballistic chunk only the
 * chunk's last tick exchanges; the ranks OR their violation flags (one small all-reduce) and all commit or all roll back;
 * exact ticks probe for (own body, ghost) pairs first and migrate such an island to the lower rank.
 *
 * Collectives are two calls.  dmxShardCreateRccl binds them to RCCL (ncclAllGather on the side stream, ncclAllReduce for the
 * flags; librccl is loaded at that call, the library does not link it).  dmxShardCreate takes them from the caller: tests
 * inject host-staged ones so that several ranks can share one GPU, which RCCL does not allow.
 */
#ifndef DMX_SHARD_H
#define DMX_SHARD_H

#include <stddef.h>
#include <stdint.h>
#include "dmx_batch.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dmxShard *dmxShardID;

typedef struct dmxCollectives {
    void *ctx;
    /* every rank contributes `bytes` bytes at send_dev; recv_dev receives world * bytes, rank r's at offset r * bytes.  Device
     * memory; the work is to be ENQUEUED on hip_stream (a hipStream_t) -- or done before returning, after synchronising that
     * stream, by implementations that stage through the host.  Returns 0 on success. */
    int (*all_gather)(void *ctx, const void *send_dev, void *recv_dev, size_t bytes, void *hip_stream);
    /* element-wise maximum of n int32 values in HOST memory over the ranks, in place, blocking.  Returns 0 on success. */
    int (*all_reduce_max)(void *ctx, int32_t *vals, int n);
} dmxCollectives;

#define DMX_RCCL_ID_BYTES 128
/* rank 0: a fresh RCCL unique id (ncclGetUniqueId) for the other ranks' dmxShardCreateRccl */
int dmxShardRcclUniqueId(void *id_out);
/* diagnostics: the path of the librccl this process loaded (up to cap - 1 characters; empty if none) and whether `s` (may be
 * NULL) holds a communicator brought up by ncclCommInitRank.  Returns DMX_ENODEVICE when no librccl could be loaded. */
int dmxShardRcclInfo(dmxShardID s, char *path_out, int cap, int *comm_up);
int dmxShardCreateRccl(dmxShardID *out, dmxBatchID batch, int64_t side, int64_t rows, int64_t spare, int rank, int world,
                       const void *rccl_unique_id);
int dmxShardCreate(dmxShardID *out, dmxBatchID batch, int64_t side, int64_t rows, int64_t spare, int rank, int world,
                   const dmxCollectives *collectives);
/* nticks ticks of step h.  Returns without waiting for the device; a ballistic chunk may stay open across calls. */
int dmxShardRun(dmxShardID s, double h, int nticks);
/* close the open chunk (zone test, exchange, flag all-reduce; rollback + replay on a violation anywhere) and wait for the
 * exchange in flight: after this the batch may be read.  Collective: every rank calls it at the same point. */
int dmxShardSettle(dmxShardID s);
/* counters: [0] exchanges issued, [1] chunks committed, [2] chunks rolled back, [3] exact ticks, [4] bodies adopted from the
 * upper neighbour, [5] own bodies retired to the lower neighbour */
int dmxShardStats(dmxShardID s, int64_t out[6]);
int dmxShardDestroy(dmxShardID s);

#ifdef __cplusplus
}
#endif
#endif
