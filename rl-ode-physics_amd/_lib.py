"""ctypes binding of libode_mi355.so (the C ABI of include/dmx_batch.h).

Fails loudly when the library has not been built: there is no Python or CPU
substitute for the HIP path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libode_mi355.so")

# symbols of include/dmx_batch.h (checked by tests/test_abi.py against the header)
BATCH_SYMBOLS = [
    "dmxVersion", "dmxDeviceCount", "dmxBatchCreate", "dmxBatchDestroy", "dmxBatchBodyCount",
    "dmxBatchPrecision", "dmxBatchSetGravity", "dmxBatchSetERP", "dmxBatchSetCFM",
    "dmxBatchSetQuickStep", "dmxBatchSetGyroMode", "dmxBatchSetSurface", "dmxBatchSetMaxContacts",
    "dmxBatchSetPlane", "dmxBatchUpload", "dmxBatchDownload", "dmxBatchUploadGeomType",
    "dmxBatchDevicePtr", "dmxBatchStride", "dmxBatchStep", "dmxBatchSynchronize", "dmxBatchSetStream",
    "dmxBatchStepTimed", "dmxBatchLastContactCount", "dmxBatchLastResidual", "dmxBatchPackTransforms",
    "dmxBatchDownloadTransforms", "dmxBatchGatherBodies", "dmxBatchScatterBodies",
    "dmxBatchStepJoints", "dmxBatchUploadBodyFlags", "dmxBatchSetActiveCount", "dmxBatchStepRange",
    "dmxBatchGetStream", "dmxBatchSetBodyCollisions", "dmxBatchCollisionStats",
    "dmxBatchScatterBodiesOnStream", "dmxBatchSetBoundaryPack",
    "dmxBatchChunkBegin", "dmxBatchChunkTick", "dmxBatchCheckZonesOnStream", "dmxBatchChunkEnd",
    "dmxBatchChunkCommit", "dmxBatchChunkRollback", "dmxBatchExactTick", "dmxBatchRefreshGhostsOnStream", "dmxBatchSetConvexHull", "dmxBatchChunkTicks", "dmxBatchSetTicksPerLaunch",
    "dmxBatchSetSnapshotMode", "dmxBatchSetStaticBoxes", "dmxBatchSetStepper", "dmxBatchSetConvexHullFaces",
    "dmxBatchCollisionStatsEx", "dmxBatchFindPairs", "dmxBatchCrossPairs", "dmxBatchSetRowOrder", "dmxBatchLcpStats", "dmxBatchSetExactPipeline", "dmxBatchSetStaticPath", "dmxBatchSetClassPairs",
]
SHARD_SYMBOLS = ["dmxShardRcclUniqueId", "dmxShardRcclInfo", "dmxShardCreateRccl", "dmxShardCreate", "dmxShardRun", "dmxShardSettle", "dmxShardStats", "dmxShardDestroy"]

_lib = None


def load():
    """Load libode_mi355.so once; raise if it is missing (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the HIP path.")
    # PyTorch-ROCm bundles its own libamdhip64.so.7; load it first so this library binds to the same
    # HIP runtime instance (one runtime per process: streams, events and device pointers are shared).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    P, I, L, D = C.c_void_p, C.c_int, C.c_int64, C.c_double

    def sig(name, res, *args):
        f = getattr(lib, name)
        f.restype = res
        f.argtypes = list(args)

    sig("dmxVersion", C.c_char_p)
    sig("dmxDeviceCount", I)
    sig("dmxBatchCreate", I, C.POINTER(P), L, I, I)
    sig("dmxBatchDestroy", I, P)
    sig("dmxBatchBodyCount", L, P)
    sig("dmxBatchPrecision", I, P)
    sig("dmxBatchSetGravity", I, P, D, D, D)
    sig("dmxBatchSetERP", I, P, D)
    sig("dmxBatchSetCFM", I, P, D)
    sig("dmxBatchSetQuickStep", I, P, I, D)
    sig("dmxBatchSetGyroMode", I, P, I)
    sig("dmxBatchSetSurface", I, P, I, D, D, D)
    sig("dmxBatchSetMaxContacts", I, P, I)
    sig("dmxBatchSetPlane", I, P, D, D, D, D, I)
    sig("dmxBatchUpload", I, P, I, P, L, L)
    sig("dmxBatchDownload", I, P, I, P, L, L)
    sig("dmxBatchUploadGeomType", I, P, P, L, L)
    sig("dmxBatchDevicePtr", P, P, I, I)
    sig("dmxBatchStride", L, P)
    sig("dmxBatchStep", I, P, D, I)
    sig("dmxBatchSynchronize", I, P)
    sig("dmxBatchSetStream", I, P, P)
    sig("dmxBatchStepTimed", I, P, D, I, C.POINTER(C.c_float))
    sig("dmxBatchLastContactCount", I, P, C.POINTER(L))
    sig("dmxBatchLastResidual", I, P, C.POINTER(D))
    sig("dmxBatchPackTransforms", I, P, P, L, L)
    sig("dmxBatchDownloadTransforms", I, P, P, L, L)
    sig("dmxBatchGatherBodies", I, P, P, L, P)
    sig("dmxBatchScatterBodies", I, P, P, L, P)
    sig("dmxBatchStepJoints", I, P, D, L, P)
    sig("dmxBatchUploadBodyFlags", I, P, P, L, L)
    sig("dmxBatchSetActiveCount", I, P, L)
    sig("dmxBatchStepRange", I, P, D, L, L, I)
    sig("dmxBatchGetStream", I, P, C.POINTER(P))
    sig("dmxBatchSetBodyCollisions", I, P, I)
    sig("dmxBatchCollisionStats", I, P, C.POINTER(L))
    sig("dmxBatchScatterBodiesOnStream", I, P, P, L, P, P)
    sig("dmxBatchSetBoundaryPack", I, P, P, L, L)
    sig("dmxBatchChunkBegin", I, P, C.POINTER(I), C.POINTER(I))
    sig("dmxBatchChunkTick", I, P, D, I)
    sig("dmxBatchCheckZonesOnStream", I, P, P, L, L)
    sig("dmxBatchChunkEnd", I, P, C.POINTER(I), C.POINTER(I))
    sig("dmxBatchChunkCommit", I, P, I, I)
    sig("dmxBatchChunkRollback", I, P)
    sig("dmxBatchExactTick", I, P, D)
    sig("dmxBatchRefreshGhostsOnStream", I, P, P, L, L, P, L, P, I)
    sig("dmxBatchSetConvexHull", I, P, C.c_int32, P, C.POINTER(D))
    sig("dmxBatchChunkTicks", I, P, D, I, I, I)
    sig("dmxBatchSetTicksPerLaunch", I, P, I)
    sig("dmxBatchSetSnapshotMode", I, P, I)
    sig("dmxBatchSetExactPipeline", I, P, I)
    sig("dmxBatchSetStaticPath", I, P, I)
    sig("dmxBatchSetClassPairs", I, P, I, I, I)
    # include/dmx_shard.h
    sig("dmxShardRcclUniqueId", I, P)
    sig("dmxShardCreateRccl", I, C.POINTER(P), P, L, L, L, I, I, P)
    sig("dmxShardCreate", I, C.POINTER(P), P, L, L, L, I, I, P)
    sig("dmxShardRcclInfo", I, P, P, I, C.POINTER(I))
    sig("dmxShardRun", I, P, D, I)
    sig("dmxShardSettle", I, P)
    sig("dmxShardStats", I, P, C.POINTER(L))
    sig("dmxShardDestroy", I, P)
    sig("dmxBatchSetStaticBoxes", I, P, C.c_int32, P, P, P)
    sig("dmxBatchSetStepper", I, P, I)
    sig("dmxBatchLcpStats", I, P, P)
    sig("dmxBatchSetConvexHullFaces", I, P, C.c_int32, P)
    sig("dmxBatchCollisionStatsEx", I, P, C.POINTER(L))
    sig("dmxBatchFindPairs", I, P, C.POINTER(P), C.POINTER(L), C.POINTER(P), C.POINTER(L))
    sig("dmxBatchCrossPairs", I, P, C.POINTER(P), C.POINTER(L))
    sig("dmxBatchSetRowOrder", I, P, I, C.c_uint32)
    _lib = lib
    return lib
