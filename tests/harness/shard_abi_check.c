/* tests/harness/shard_abi_check.c -- a plain C99 client of include/dmx_shard.h: one rank's slab of a sharded world stepped
 * by dmxShardRun through a ONE-RANK RCCL communicator (the library's own RCCL binding: ncclGetUniqueId, ncclCommInitRank,
 * ncclAllGather on the side stream), as a C host like the reference's main.c would drive it.  Reads the slab's bodies from
 * a binary file written by tests/test_gpu_shard_abi.py (doubles), steps in calls of 1, 7 and `rest` ticks, prints the 13-real
 * state of every body as hex doubles for the test to hold against the oracle.
 * Usage: shard_abi_check scene.bin side rows ticks f32|f64 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "dmx_batch.h"
#include "dmx_shard.h"

#define CHECK(call) do { int rc_ = (call); if (rc_ != DMX_OK) { fprintf(stderr, "%s -> %d\n", #call, rc_); return 10; } } while (0)

static int upload(dmxBatchID b, int field, const double *src, int k, int64_t n, int f32)
{
    int64_t i;
    int rc;
    if (!f32) return dmxBatchUpload(b, field, src, 0, n);
    {
        float *t = (float *)malloc((size_t)(n * k) * sizeof(float));
        for (i = 0; i < n * k; i++) t[i] = (float)src[i];
        rc = dmxBatchUpload(b, field, t, 0, n);
        free(t);
    }
    return rc;
}

int main(int argc, char **argv)
{
    int64_t side, rows, n, total, i;
    int ticks, f32, c;
    double *d;
    uint8_t *types;
    dmxBatchID b = NULL;
    dmxShardID s = NULL;
    char id[DMX_RCCL_ID_BYTES];
    int64_t st[6];
    FILE *f;
    if (argc < 6) return 2;
    side = atoll(argv[2]); rows = atoll(argv[3]); ticks = atoi(argv[4]); f32 = strcmp(argv[5], "f32") == 0;
    n = side * rows; total = n + 2 * side;          /* no spare slots: nothing crosses a one-rank world's faces */
    d = (double *)malloc((size_t)n * 21 * sizeof(double));
    f = fopen(argv[1], "rb");
    if (!f || fread(d, sizeof(double), (size_t)n * 21, f) != (size_t)n * 21) { fprintf(stderr, "scene file\n"); return 2; }
    fclose(f);
    CHECK(dmxBatchCreate(&b, total, f32 ? DMX_F32 : DMX_F64, 0));
    CHECK(dmxBatchSetGravity(b, 0.0, -9.8, 0.0));
    /* file layout, body-major blocks: pos 3n, quat 4n, lvel 3n, avel 3n, mass n, inertia 3n, sides 3n, then n class bytes as doubles */
    CHECK(upload(b, DMX_POS, d, 3, n, f32));
    CHECK(upload(b, DMX_QUAT, d + 3 * n, 4, n, f32));
    CHECK(upload(b, DMX_LVEL, d + 7 * n, 3, n, f32));
    CHECK(upload(b, DMX_AVEL, d + 10 * n, 3, n, f32));
    CHECK(upload(b, DMX_MASS, d + 13 * n, 1, n, f32));
    CHECK(upload(b, DMX_INERTIA, d + 14 * n, 3, n, f32));
    CHECK(upload(b, DMX_SIDES, d + 17 * n, 3, n, f32));
    types = (uint8_t *)malloc((size_t)n);
    for (i = 0; i < n; i++) types[i] = (uint8_t)d[20 * n + i];
    CHECK(dmxBatchUploadGeomType(b, types, 0, n));
    CHECK(dmxShardRcclUniqueId(id));
    CHECK(dmxShardCreateRccl(&s, b, side, rows, 0, 0, 1, id));
    CHECK(dmxShardRun(s, 1.0 / 60.0, 1));
    CHECK(dmxShardRun(s, 1.0 / 60.0, 7));
    CHECK(dmxShardRun(s, 1.0 / 60.0, ticks - 8));
    CHECK(dmxShardSettle(s));
    CHECK(dmxShardStats(s, st));
    printf("stats exchanges %lld committed %lld rolled_back %lld exact %lld\n", (long long)st[0], (long long)st[1], (long long)st[2], (long long)st[3]);
    if (f32) {
        float *o = (float *)malloc((size_t)n * 13 * sizeof(float));
        CHECK(dmxBatchDownload(b, DMX_STATE, o, 0, n));
        for (i = 0; i < n; i++) { printf("body"); for (c = 0; c < 13; c++) printf(" %a", (double)o[13 * i + c]); printf("\n"); }
        free(o);
    } else {
        double *o = (double *)malloc((size_t)n * 13 * sizeof(double));
        CHECK(dmxBatchDownload(b, DMX_STATE, o, 0, n));
        for (i = 0; i < n; i++) { printf("body"); for (c = 0; c < 13; c++) printf(" %a", o[13 * i + c]); printf("\n"); }
        free(o);
    }
    CHECK(dmxShardDestroy(s));
    CHECK(dmxBatchDestroy(b));
    free(d); free(types);
    return 0;
}
