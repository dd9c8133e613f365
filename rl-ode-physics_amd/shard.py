"""Island sharding across the GPUs of one node.

Dynamics islands are independent -- static geometry does not link islands -- so a scene shards across
GPUs as whole islands, one process per GPU.  For the grid scenes rank r owns one slab of `rows` grid rows
stacked along z.  The only data a neighbour ever needs is the state of the bodies next to the shared slab
face (the ones a body-body broadphase on the neighbour can reach): the slab's first and last row.

Every tick each rank
  1. steps its slab; the step kernel also writes the new 13-real state of the two boundary rows, packed,
     into the send buffer                                     (dmxBatchSetBoundaryPack, batch stream)
  2. (hosts without the fused pack, e.g. the CPU test double, gather the rows explicitly)
  3. all-gathers the packed rows over RCCL / xGMI              (torch.distributed, side stream)
  4. writes its neighbours' rows into its ghost slots          (dmxBatchScatterBodiesOnStream, side stream)
Steps 3-4 of tick k run on the side stream while the batch stream already integrates tick k+1: ghost slots
([n_active, n) of the batch) are never read or written by the step kernels, and the pack of tick k+1 waits for
exchange k-1 to have drained the (double-buffered) send buffer.  Consumers of ghost state (broadphase rebuild / pair
search) wait for the in-flight exchange first (`drain()`).

For slabs farther apart than a broadphase cell (BASELINE configs[3], >= 10 m) the boundary set is empty and
`exchange="none"` skips steps 2-4.

The exchange is written against a tiny `ops` interface (gather / scatter / buffers / streams) so the index
logic and the collective run unchanged on CPU tensors with the gloo backend (tests/test_shard_gloo.py).
"""
import contextlib

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
    """gather on the batch's HIP stream, scatter on the side stream; buffers are torch CUDA tensors."""

    def __init__(self, world_batch, device, main_stream):
        self.w = world_batch
        self.device = device
        self.torch_dtype = torch.float32 if world_batch.dtype.itemsize == 4 else torch.float64
        self.main = main_stream                      # the stream the batch launches on
        self.side = torch.cuda.Stream(device=device)
        self.packed = torch.cuda.Event()
        self.done = [torch.cuda.Event(), torch.cuda.Event()]     # per send buffer: "the exchange that read it finished"
        self.have_done = [False, False]

    def empty(self, *shape):
        return torch.empty(shape, dtype=self.torch_dtype, device=self.device)

    def index(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(self.device)

    fused_pack = True      # the step kernel fills the send buffer itself

    def arm_pack(self, out, layout):
        """Point the next whole-slab tick's boundary pack at `out`."""
        self.w.set_boundary_pack(out.data_ptr(), layout.side, layout.n - layout.side)

    def gather(self, idx, out):
        self.w.gather_bodies(idx.data_ptr(), idx.numel(), out.data_ptr())

    def scatter(self, idx, src):
        assert src.is_contiguous()
        self.w.scatter_bodies_on(self.side.cuda_stream, idx.data_ptr(), idx.numel(), src.data_ptr())

    # -- stream choreography -------------------------------------------------------------------------------
    def before_pack(self, k=0):
        if self.have_done[k & 1]:
            self.main.wait_event(self.done[k & 1])   # exchange k-2 (same send buffer) has finished; k-1 may still run

    def after_pack(self):
        self.packed.record(self.main)
        self.side.wait_event(self.packed)

    def side_stream(self):
        return torch.cuda.stream(self.side)

    def after_exchange(self, k=0):
        self.done[k & 1].record(self.side)
        self.have_done[k & 1] = True

    def drain(self):
        for b in (0, 1):
            if self.have_done[b]:
                self.main.wait_event(self.done[b])

    def forget(self):
        self.have_done = [False, False]


class BoundaryExchange:
    def __init__(self, ops, layout, rank, world_size, group=None):
        self.ops, self.L, self.rank, self.world = ops, layout, rank, world_size
        self.group = group
        self.send_idx = ops.index(layout.send_idx)
        self.ghost_lo = ops.index(layout.ghost_lo)
        self.ghost_hi = ops.index(layout.ghost_hi)
        # two send buffers: tick k's step kernel fills one while exchange k-1 still reads the other
        self.sends = [ops.empty(layout.n_send, STATE_REALS), ops.empty(layout.n_send, STATE_REALS)]
        self.k = 0
        self.fused = bool(getattr(ops, "fused_pack", False))
        # all_gather_into_tensor wants the ranks' buffers concatenated along dim 0
        self.recv_flat = ops.empty(world_size * layout.n_send, STATE_REALS)
        self.recv = self.recv_flat.view(world_size, layout.n_send, STATE_REALS)

    @property
    def send(self):
        return self.sends[self.k & 1]

    def before_step(self):
        """Called before the tick's step kernel is enqueued: make sure this tick's send buffer is free and, with a
        fused pack, aim the step kernel at it."""
        self.ops.before_pack(self.k)
        if self.fused:
            self.ops.arm_pack(self.send, self.L)

    def pack(self):
        if not self.fused:
            self.ops.gather(self.send_idx, self.send)
        self.ops.after_pack()

    def exchange(self):
        """All-gather every rank's boundary rows and refresh the ghost slots (on the ops' side stream)."""
        with self.ops.side_stream():
            dist.all_gather_into_tensor(self.recv_flat, self.send, group=self.group)
            side = self.L.side
            if self.rank > 0:                                   # lower neighbour's upper row
                self.ops.scatter(self.ghost_lo, self.recv[self.rank - 1, side:2 * side])
            if self.rank < self.world - 1:                      # upper neighbour's lower row
                self.ops.scatter(self.ghost_hi, self.recv[self.rank + 1, 0:side])
        self.ops.after_exchange(self.k)
        self.k += 1

    def tick(self):
        """pack + exchange for hosts that do not interleave a step kernel (tests)."""
        self.before_step()
        self.pack()
        self.exchange()

    def drain(self):
        self.ops.drain()


class ShardedStepper:
    """One rank's tick loop: step the slab, then hand the boundary rows to the side-stream exchange."""

    def __init__(self, world_batch, layout, rank, world_size, exchange="boundary", device=None, stream=None):
        self.w, self.L = world_batch, layout
        self.exchange = None
        if world_size > 1 and exchange == "boundary":
            self.exchange = BoundaryExchange(DeviceOps(world_batch, device, stream), layout, rank, world_size)
        self.graph = None
        self.graph_steps = 0

    def tick(self, h):
        ex = self.exchange
        if ex is None:
            self.w.step(h, 1)
            return
        ex.before_step()
        self.w.step(h, 1)
        ex.pack()
        ex.exchange()

    def run(self, h, nsteps):
        if self.exchange is None:
            self.w.step(h, nsteps)          # the C loop: no per-tick host work
            return
        if self.graph is not None:
            reps, nsteps = divmod(nsteps, self.graph_steps)
            for _ in range(reps):
                self.graph.replay()
        for _ in range(nsteps):
            self.tick(h)

    def drain(self):
        if self.exchange is not None:
            self.exchange.drain()

    def capture(self, h, steps_per_graph, stream):
        """Capture `steps_per_graph` ticks (kernels, the RCCL all-gather and the stream choreography) into one
        HIP graph so a replay costs one host call; returns False (and stays eager) if capture is not possible."""
        if self.exchange is None:
            return False
        ops = self.exchange.ops
        try:
            ops.drain()
            torch.cuda.synchronize()
            ops.forget()                               # no event edges from outside the capture
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for _ in range(steps_per_graph):
                    self.tick(h)
                ops.drain()                            # the side stream rejoins before the capture ends
            ops.forget()
            self.graph, self.graph_steps = g, steps_per_graph
            return True
        except Exception as e:      # noqa: BLE001 -- capture support varies; eager is always correct
            self.graph = None
            ops.forget()
            with contextlib.suppress(Exception):
                torch.cuda.synchronize()
            print(f"[shard] graph capture unavailable ({type(e).__name__}: {e}); running eagerly", flush=True)
            return False
