"""Island sharding across the GPUs of one node.

Dynamics islands are independent -- static geometry does not link islands -- so a scene shards across
GPUs as whole islands, one process per GPU.  For the grid scenes rank r owns one slab of `rows` grid rows
stacked along z.  The only data a neighbour ever needs is the state of the bodies next to the shared slab
face (they are the ones a body-body broadphase on the neighbour can reach): the slab's first and last
row.  Each tick every rank

  1. steps its two boundary rows                                   (dmxBatchStepRange)
  2. packs their 13-real state into one buffer                     (dmxBatchGatherBodies)
  3. all-gathers the buffers over RCCL / xGMI, asynchronously      (torch.distributed, backend nccl)
  4. steps the slab interior while the collective is in flight     (dmxBatchStepRange)
  5. writes the neighbours' rows into its ghost slots              (dmxBatchScatterBodies)

Ghost slots live behind the rank's own bodies ([n_active, n) of the batch) and are never stepped.
For slabs that are farther apart than a broadphase cell (BASELINE configs[3], >= 10 m) the boundary set
is empty and `exchange="none"` skips steps 2-3-5.

The exchange is written against a tiny `ops` interface (gather / scatter / buffers) so the index logic
runs unchanged on CPU tensors with the gloo backend in tests (tests/test_shard_gloo.py).
"""
import numpy as np
import torch
import torch.distributed as dist

STATE_REALS = 13      # pos3 quat4 lvel3 avel3


class SlabLayout:
    """Row-major slab: body i sits in grid row i // side (z) and column i % side (x)."""

    def __init__(self, side, rows):
        self.side, self.rows = int(side), int(rows)
        self.n = self.side * self.rows
        assert self.rows >= 2 and self.side % 4 == 0
        self.lower = np.arange(0, self.side, dtype=np.int32)                 # first row: faces rank-1
        self.upper = np.arange(self.n - self.side, self.n, dtype=np.int32)   # last row:  faces rank+1
        self.send_idx = np.concatenate([self.lower, self.upper])
        self.n_send = 2 * self.side
        # ghost slots behind the active bodies: rank-1's upper row, then rank+1's lower row
        self.ghost_lo = np.arange(self.n, self.n + self.side, dtype=np.int32)
        self.ghost_hi = np.arange(self.n + self.side, self.n + 2 * self.side, dtype=np.int32)
        self.n_total = self.n + 2 * self.side

    @property
    def interior(self):
        return self.side, self.n - 2 * self.side          # first, count


class DeviceOps:
    """gather/scatter through the C ABI on the batch's HIP stream; buffers are torch CUDA tensors."""

    def __init__(self, world_batch, device):
        self.w = world_batch
        self.device = device
        self.torch_dtype = torch.float32 if world_batch.dtype.itemsize == 4 else torch.float64

    def empty(self, *shape):
        return torch.empty(shape, dtype=self.torch_dtype, device=self.device)

    def index(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(self.device)

    def gather(self, idx, out):
        self.w.gather_bodies(idx.data_ptr(), idx.numel(), out.data_ptr())

    def scatter(self, idx, src):
        assert src.is_contiguous()
        self.w.scatter_bodies(idx.data_ptr(), idx.numel(), src.data_ptr())


class BoundaryExchange:
    def __init__(self, ops, layout, rank, world_size, group=None):
        self.ops, self.L, self.rank, self.world = ops, layout, rank, world_size
        self.group = group
        self.send_idx = ops.index(layout.send_idx)
        self.ghost_lo = ops.index(layout.ghost_lo)
        self.ghost_hi = ops.index(layout.ghost_hi)
        self.send = ops.empty(layout.n_send, STATE_REALS)
        # all_gather_into_tensor wants the ranks' buffers concatenated along dim 0
        self.recv_flat = ops.empty(world_size * layout.n_send, STATE_REALS)
        self.recv = self.recv_flat.view(world_size, layout.n_send, STATE_REALS)
        self.work = None

    def pack(self):
        self.ops.gather(self.send_idx, self.send)

    def start(self):
        """All-gather every rank's boundary rows; returns immediately (the collective runs on RCCL's stream)."""
        self.work = dist.all_gather_into_tensor(self.recv_flat, self.send, group=self.group, async_op=True)

    def finish(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        side = self.L.side
        if self.rank > 0:                                   # lower neighbour's upper row
            self.ops.scatter(self.ghost_lo, self.recv[self.rank - 1, side:2 * side])
        if self.rank < self.world - 1:                      # upper neighbour's lower row
            self.ops.scatter(self.ghost_hi, self.recv[self.rank + 1, 0:side])

    def exchange(self):
        self.pack()
        self.start()
        self.finish()


class ShardedStepper:
    """One rank's tick loop: boundary rows first, exchange overlapped with the interior."""

    def __init__(self, world_batch, layout, rank, world_size, exchange="boundary", device=None):
        self.w, self.L = world_batch, layout
        self.exchange = None
        if world_size > 1 and exchange == "boundary":
            self.exchange = BoundaryExchange(DeviceOps(world_batch, device), layout, rank, world_size)
        self.graph = None
        self.graph_steps = 0

    def tick(self, h):
        w, L = self.w, self.L
        if self.exchange is None:
            w.step(h, 1)
            return
        w.step_range(h, 0, L.side, reset_diag=True)
        w.step_range(h, L.n - L.side, L.side)
        self.exchange.pack()
        self.exchange.start()
        first, count = L.interior
        w.step_range(h, first, count)
        self.exchange.finish()

    def run(self, h, nsteps):
        if self.exchange is None:
            self.w.step(h, nsteps)          # the C loop: no per-tick host work
            return
        if self.graph is not None:
            reps, rest = divmod(nsteps, self.graph_steps)
            for _ in range(reps):
                self.graph.replay()
            nsteps = rest
        for _ in range(nsteps):
            self.tick(h)

    def capture(self, h, steps_per_graph, stream):
        """Capture `steps_per_graph` ticks (kernels + the RCCL all-gather) into one HIP graph so a replay
        costs one host call; returns False (and stays eager) if capture is not possible."""
        if self.exchange is None:
            return False
        try:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for _ in range(steps_per_graph):
                    self.tick(h)
            self.graph, self.graph_steps = g, steps_per_graph
            return True
        except Exception as e:      # noqa: BLE001 -- capture support varies; eager is always correct
            self.graph = None
            torch.cuda.synchronize()
            print(f"[shard] graph capture unavailable ({type(e).__name__}: {e}); running eagerly", flush=True)
            return False
