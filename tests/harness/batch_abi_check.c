/* tests/harness/batch_abi_check.c -- a plain C99 client of include/dmx_batch.h and include/dmx_hull.h: proves the
 * headers are C (not C++) clean and that a C host drives the batch path end to end.  Drops n boxes on the ground plane
 * for `steps` ticks and prints the lowest and highest resting height; exit code 0 when every box rests on the plane.
 * Usage: batch_abi_check [n] [steps]     (without a HIP device dmxBatchCreate fails and the program exits 3) */
#include <stdio.h>
#include <stdlib.h>
#include "dmx_batch.h"
#include "dmx_hull.h"

int main(int argc, char **argv)
{
    int64_t n = argc > 1 ? atoll(argv[1]) : 4096, i;
    int steps = argc > 2 ? atoi(argv[2]) : 240, rc;
    dmxBatchID b = NULL;
    float *pos, *sides, *out;
    uint8_t *types;
    float lo = 1e30f, hi = -1e30f;
    double cube[8 * 3], hull_pts[8 * 3];
    dmxHullInfo info;
    int k = 0, x, y, z;

    /* the hull builder is host code: works without a device */
    for (x = 0; x < 2; x++) for (y = 0; y < 2; y++) for (z = 0; z < 2; z++) { cube[k++] = x; cube[k++] = y; cube[k++] = z; }
    if (dmxHullBuild(cube, 8, 1.0, hull_pts, NULL, 8, &info) != 8 || info.n_faces != 12) { fprintf(stderr, "hull builder\n"); return 2; }

    printf("%s, %d device(s)\n", dmxVersion(), dmxDeviceCount());
    rc = dmxBatchCreate(&b, n, DMX_F32, 0);
    if (rc != DMX_OK) { fprintf(stderr, "dmxBatchCreate: %d\n", rc); return 3; }
    pos = (float *)malloc((size_t)n * 3 * sizeof(float));
    sides = (float *)malloc((size_t)n * 3 * sizeof(float));
    out = (float *)malloc((size_t)n * 16 * sizeof(float));
    types = (uint8_t *)malloc((size_t)n);
    for (i = 0; i < n; i++) {
        pos[3 * i] = 2.5f * (float)(i % 64); pos[3 * i + 1] = 1.0f + 0.01f * (float)(i % 7); pos[3 * i + 2] = 2.5f * (float)(i / 64);
        sides[3 * i] = sides[3 * i + 1] = sides[3 * i + 2] = 0.5f;
        types[i] = DMX_GEOM_BOX;
    }
    dmxBatchSetGravity(b, 0.0, -9.8, 0.0);
    dmxBatchSetPlane(b, 0.0, 1.0, 0.0, 0.0, 1);
    if (dmxBatchUpload(b, DMX_POS, pos, 0, n) || dmxBatchUpload(b, DMX_SIDES, sides, 0, n) ||
        dmxBatchUploadGeomType(b, types, 0, n)) { fprintf(stderr, "upload\n"); return 4; }
    if (dmxBatchStep(b, 1.0 / 60.0, steps) || dmxBatchDownloadTransforms(b, out, 0, n)) { fprintf(stderr, "step\n"); return 5; }
    for (i = 0; i < n; i++) { float yy = out[16 * i + 13]; if (yy < lo) lo = yy; if (yy > hi) hi = yy; }
    printf("%lld boxes, %d ticks: resting heights %.4f .. %.4f\n", (long long)n, steps, (double)lo, (double)hi);
    dmxBatchDestroy(b);
    free(pos); free(sides); free(out); free(types);
    return (lo > 0.24f && hi < 0.26f) ? 0 : 1;
}
