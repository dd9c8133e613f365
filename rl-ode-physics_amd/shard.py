"""Island sharding across the GPUs of one node.

Dynamics islands are independent -- static geometry does not link islands -- so a scene shards across
GPUs as whole islands, one process per GPU.  For the grid scenes rank r owns one slab of `rows` grid rows
stacked along z.  The only data a neighbour ever needs is the state of the bodies next to the shared slab
face (the ones a body-body broadphase on the neighbour can reach): the slab's first and last row.

An exchanging tick
  1. steps the slab; the step kernel also writes the new 13-real state of the two boundary rows, packed,
     into the send buffer                                     (dmxBatchSetBoundaryPack, batch stream)
  2. (hosts without the fused pack, e.g. the CPU test double, gather the rows explicitly)
  3. all-gathers the packed rows over RCCL / xGMI              (torch.distributed, side stream)
  4. writes its neighbours' rows into its ghost slots and tests them against their safe zones, in one launch
                                                               (dmxBatchRefreshGhostsOnStream, side stream)
Steps 3-4 of tick k run on the side stream while the batch stream already integrates tick k+1: ghost slots
([n_active, n) of the batch) are never read or written by the step kernels, and a tick's pack waits for the exchange
that last read its send buffer (a ring of buffers).  Consumers of ghost state (broadphase rebuild / pair search) wait
for the in-flight exchange first (`drain()`).  Which ticks exchange: every tick, except inside a ballistic chunk of
the collision-checked loop, where only the chunk's last tick does (see ShardedStepper._fast_ticks).

For slabs farther apart than a broadphase cell (BASELINE configs[3], >= 10 m) the boundary set is empty and
`exchange="none"` skips steps 2-4.

Body-body collisions (dSpaceCollide for body pairs) ride along as in the single-GPU loop: every body, ghosts included,
carries a broadphase safe zone; the step kernel checks the rank's own bodies, a small kernel on the side stream checks
the ghost slots right after their refresh.  Ticks run in chunks; at a chunk's end the ranks OR their violation flags
(one tiny all-reduce) and either all commit or all roll back to the chunk's snapshot and replay it -- first with fresh
zones, then tick by tick on the exact path (pair search, narrowphase, island solve).  A pair of bodies owned by two
different ranks makes an island that spans them: the product loop -- the same loop behind the C ABI, include/dmx_shard.h,
bound by CShardedStepper at the end of this module -- probes for such pairs before an exact tick and migrates the island to
the lower rank (the boundary body is adopted into a spare slot, the upper rank retires its copy and keeps seeing the body as a
ghost); the classes below, kept as the index-logic reference the CPU tests drive over gloo, report it (DMX_ECROSS).

The exchange is written against a tiny `ops` interface (gather / scatter / buffers / streams) so the index
logic and the collective run unchanged on CPU tensors with the gloo backend (tests/test_shard_gloo.py).
"""
import contextlib

import numpy as np
import torch
import torch.distributed as dist

STATE_REALS = 13      # pos3 quat4 lvel3 avel3


class SlabLayout:
    """Row-major slab: body i sits in grid row i // side (z) and column i % side (x).

    Slots: [0, n) the rank's own bodies; [n, n_active) `spare` empty slots (stepped, geometry class NONE) that take bodies
    adopted from the upper neighbour when an island spans the shared face; then ghost copies of the neighbours' boundary rows
    (2 x side) and of the lower neighbour's spare slots (`spare`: a body that neighbour adopted from this rank stays visible
    here) -- include/dmx_shard.h."""

    def __init__(self, side, rows, spare=0):
        self.side, self.rows, self.spare = int(side), int(rows), int(spare)
        self.n = self.side * self.rows
        assert self.rows >= 2 and self.side % 4 == 0 and self.spare % 4 == 0
        self.n_active = self.n + self.spare
        self.lower = np.arange(0, self.side, dtype=np.int32)                 # first row: faces rank-1
        self.upper = np.arange(self.n - self.side, self.n, dtype=np.int32)   # last row:  faces rank+1
        self.send_idx = np.concatenate([self.lower, self.upper])
        self.n_send = 2 * self.side
        # ghost slots behind the active bodies: rank-1's upper row, then rank+1's lower row
        self.ghost_lo = np.arange(self.n_active, self.n_active + self.side, dtype=np.int32)
        self.ghost_hi = np.arange(self.n_active + self.side, self.n_active + 2 * self.side, dtype=np.int32)
        self.n_total = self.n_active + 2 * self.side + self.spare

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
        self.set_ring(2)

    def set_ring(self, ring):
        """one event per send buffer: "the exchange that read it finished" """
        self.ring = int(ring)
        self.done = [torch.cuda.Event() for _ in range(self.ring)]
        self.have_done = [False] * self.ring

    def empty(self, *shape):
        return torch.empty(shape, dtype=self.torch_dtype, device=self.device)

    def index(self, arr):
        return torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int32)).to(self.device)

    fused_pack = True      # the step kernel fills the send buffer itself

    def arm_pack(self, out, layout):
        """Point the next whole-slab tick's boundary pack at `out`."""
        self.w.set_boundary_pack(out.data_ptr(), layout.side, layout.n - layout.side)
        self._armed = True

    def gather(self, idx, out):
        self.w.gather_bodies(idx.data_ptr(), idx.numel(), out.data_ptr())

    def scatter(self, idx, src):
        assert src.is_contiguous()
        self.w.scatter_bodies_on(self.side.cuda_stream, idx.data_ptr(), idx.numel(), src.data_ptr())

    # -- stream choreography -------------------------------------------------------------------------------
    def before_pack(self, k=0):
        if self.have_done[k % self.ring]:
            self.main.wait_event(self.done[k % self.ring])   # the exchange that last read this send buffer has finished; later ones may still run

    def after_pack(self):
        self.packed.record(self.main)
        self.side.wait_event(self.packed)

    def side_stream(self):
        return torch.cuda.stream(self.side)

    def after_exchange(self, k=0):
        self.done[k % self.ring].record(self.side)
        self.have_done[k % self.ring] = True
        self.last = k % self.ring

    def drain(self):
        if any(self.have_done):
            self.main.wait_event(self.done[self.last])       # the side stream is in order: its newest exchange covers the older ones

    def forget(self):
        self.have_done = [False] * self.ring

    def disarm_pack(self):
        if getattr(self, "_armed", True):            # (a call through the C ABI each time would be the only host work of a short run() call)
            self.w.set_boundary_pack(0, 0, 0)
            self._armed = False

    def refresh_ghosts(self, first, count_lo, src_lo, count_hi, src_hi, check):
        """both neighbours' rows into the ghost slots (and their zone test) in one launch on the side stream"""
        ptr = lambda t: None if t is None else t.data_ptr()
        self.w.refresh_ghosts_on(self.side.cuda_stream, first, count_lo, ptr(src_lo), count_hi, ptr(src_hi), check)

    def check_ghosts(self, first, count):
        """safe-zone test of the freshly written ghost slots, behind the scatter on the side stream"""
        self.w.check_zones_on(self.side.cuda_stream, first, count)

    def check_active(self, first, count):
        """safe-zone test of the rank's own bodies as they stand, on the batch's stream (a lazily kept chunk that is
        closed before it reaches its length: its last tick did not carry the test)"""
        self.w.check_zones_on(self.main.cuda_stream, first, count)

    def any_rank(self, flags, group=None):
        """element-wise logical OR of a few host flags over the ranks (one small all-reduce)"""
        flags = [bool(f) for f in flags]
        if dist.get_world_size(group) == 1:
            return flags
        t = torch.tensor([int(f) for f in flags], dtype=torch.int32, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return [bool(v) for v in t.tolist()]


class StagedDeviceOps(DeviceOps):
    """DeviceOps with the collectives staged through host memory (a gloo group): lets several ranks share one GPU, which
    RCCL does not allow -- for rehearsing the N>1 loop on a box with fewer GPUs than ranks (tests, `bench.py
    --rehearse-on-one-gpu`).  Everything but the collective itself is the product path."""

    def all_gather(self, out, mine, group=None):
        torch.cuda.current_stream().synchronize()
        h = mine.cpu()
        o = torch.empty((out.shape[0],) + tuple(h.shape[1:]), dtype=h.dtype)
        dist.all_gather_into_tensor(o, h, group=group)
        out.copy_(o)

    def any_rank(self, flags, group=None):
        t = torch.tensor([int(bool(f)) for f in flags], dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        return [bool(v) for v in t.tolist()]


class BoundaryExchange:
    def __init__(self, ops, layout, rank, world_size, group=None):
        self.ops, self.L, self.rank, self.world = ops, layout, rank, world_size
        self.group = group
        self.send_idx = ops.index(layout.send_idx)
        self.ghost_lo = ops.index(layout.ghost_lo)
        self.ghost_hi = ops.index(layout.ghost_hi)
        # a ring of send buffers: tick k's step kernel fills one while earlier exchanges still read the others
        self.k = 0
        self.count = 0            # exchanges issued (a captured graph counts its ticks once, at capture)
        self.set_ring(2)
        self.fused = bool(getattr(ops, "fused_pack", False))
        # all_gather_into_tensor wants the ranks' buffers concatenated along dim 0
        self.recv_flat = ops.empty(world_size * layout.n_send, STATE_REALS)
        self.recv = self.recv_flat.view(world_size, layout.n_send, STATE_REALS)

    def set_ring(self, ring):
        """(Re)size the ring of send buffers (tick k packs into buffer k mod ring and waits only for the exchange that
        last read that buffer)."""
        self.ops.drain()
        self.sends = [self.ops.empty(self.L.n_send, STATE_REALS) for _ in range(int(ring))]
        if hasattr(self.ops, "set_ring"):
            self.ops.set_ring(ring)

    @property
    def send(self):
        return self.sends[self.k % len(self.sends)]

    def _all_gather(self, out, mine):
        if hasattr(self.ops, "all_gather"):
            self.ops.all_gather(out, mine, self.group)     # a host that needs staging (tests on a backend without device collectives)
        else:
            dist.all_gather_into_tensor(out, mine, group=self.group)

    def before_step(self, fused=None):
        """Called before the tick's step kernel is enqueued: make sure this tick's send buffer is free and, with a
        fused pack, aim the step kernel at it (fused=False: the tick's kernels do not pack, `pack` will gather)."""
        fused = self.fused if fused is None else (fused and self.fused)
        self.ops.before_pack(self.k)
        if fused:
            self.ops.arm_pack(self.send, self.L)
        elif self.fused:
            self.ops.disarm_pack()

    def pack(self, fused=None):
        fused = self.fused if fused is None else (fused and self.fused)
        if not fused:
            self.ops.gather(self.send_idx, self.send)
        self.ops.after_pack()

    def exchange(self, check_ghosts=False):
        """All-gather every rank's boundary rows and refresh the ghost slots (on the ops' side stream); with
        check_ghosts the refreshed slots are tested against their broadphase safe zones there too."""
        with self.ops.side_stream():
            self._all_gather(self.recv_flat, self.send)
            side = self.L.side
            lo = self.recv[self.rank - 1, side:2 * side] if self.rank > 0 else None           # lower neighbour's upper row
            hi = self.recv[self.rank + 1, 0:side] if self.rank < self.world - 1 else None     # upper neighbour's lower row
            if hasattr(self.ops, "refresh_ghosts"):
                if lo is not None or hi is not None:
                    self.ops.refresh_ghosts(self.L.n_active, side, lo, side, hi, check_ghosts)
            else:
                if lo is not None:
                    self.ops.scatter(self.ghost_lo, lo)
                if hi is not None:
                    self.ops.scatter(self.ghost_hi, hi)
                if check_ghosts:
                    self.ops.check_ghosts(self.L.n_active, 2 * side)
        self.ops.after_exchange(self.k)
        self.k += 1
        self.count += 1

    def share_geometry(self, upload_ghost, sides, gtype, mass=None, inertia=None):
        """Once, at set-up: the neighbours' boundary bodies' extents, geometry types and mass properties into the ghost
        slots, so the broadphase sees the ghosts at their true size and a ghost can be adopted as it stands when an island
        spans the face.  sides [n,3] / gtype [n] / mass [n] / inertia [n,3] describe this rank's own bodies;
        upload_ghost(first_slot, sides_rows, gtype_rows, mass_rows, inertia_rows) writes ghost slots.  The geometry types of
        the ghost rows are kept (`ghost_gtype`): an adopted body takes its class from there."""
        L, side = self.L, self.L.side
        n = len(np.asarray(gtype))
        mass = np.ones(n) if mass is None else np.asarray(mass, dtype=np.float64).reshape(n)
        inertia = np.ones((n, 3)) if inertia is None else np.asarray(inertia, dtype=np.float64).reshape(n, 3)
        mine = self.ops.empty(L.n_send, 8)
        rows = np.concatenate([np.asarray(sides)[L.send_idx], np.asarray(gtype, dtype=np.float64)[L.send_idx, None],
                               mass[L.send_idx, None], inertia[L.send_idx]], axis=1)
        mine.copy_(torch.from_numpy(np.ascontiguousarray(rows)).to(mine.dtype))
        flat = self.ops.empty(self.world * L.n_send, 8)
        self._all_gather(flat, mine)
        got = flat.view(self.world, L.n_send, 8).cpu().numpy()
        self.ghost_gtype = np.zeros(2 * side, np.uint8)          # by ghost slot - n_active
        if self.rank > 0:
            r = got[self.rank - 1, side:2 * side]
            upload_ghost(int(L.ghost_lo[0]), r[:, :3], r[:, 3].astype(np.uint8), r[:, 4], r[:, 5:8])
            self.ghost_gtype[:side] = r[:, 3].astype(np.uint8)
        if self.rank < self.world - 1:
            r = got[self.rank + 1, 0:side]
            upload_ghost(int(L.ghost_hi[0]), r[:, :3], r[:, 3].astype(np.uint8), r[:, 4], r[:, 5:8])
            self.ghost_gtype[side:] = r[:, 3].astype(np.uint8)

    def prime(self):
        """Set-up: one exchange of the current boundary rows, so the ghost slots hold the neighbours' bodies where they
        are (not at the origin) before the first broadphase build."""
        self.before_step(fused=False)
        self.pack(fused=False)
        self.exchange()
        self.drain()

    def tick(self):
        """pack + exchange for hosts that do not interleave a step kernel (tests)."""
        self.before_step()
        self.pack()
        self.exchange()

    def drain(self):
        self.ops.drain()


class ShardedStepper:
    """One rank's tick loop: step the slab, then hand the boundary rows to the side-stream exchange.

    collide=True carries the body-body collision proof through the loop (module docstring); geometry=(sides, gtype)
    of this rank's bodies is then shared with the neighbours once so their ghosts have the right extents."""

    CHUNK_MIN, CHUNK_MAX = 32, 256

    def __init__(self, world_batch, layout, rank, world_size, exchange="boundary", device=None, stream=None,
                 collide=False, geometry=None, ops=None, group=None, exchange_every_tick=False, lazy=False):
        self.w, self.L = world_batch, layout
        self.rank, self.world = rank, world_size
        # lazy: ballistic chunks stay open across run() calls (a caller that issues a few ticks per call does not pay a
        # chunk's exchange, flag read and flag all-reduce per call); settle() closes the open chunk
        self.lazy = bool(lazy)
        self._open = None
        self.exchange = None
        if (world_size > 1 and exchange == "boundary") or ops is not None:
            ops = ops if ops is not None else DeviceOps(world_batch, device, stream)
            self.exchange = BoundaryExchange(ops, layout, rank, world_size, group=group)
        self.graph = None
        self.graph_steps = 0
        self.collide = bool(collide) and self.exchange is not None
        self.exchange_every_tick = bool(exchange_every_tick)
        self.chunk = self.CHUNK_MIN
        self.ballistic_graph = None
        if self.collide and geometry is not None:
            sides, gtype = geometry[0], geometry[1]
            mass = geometry[2] if len(geometry) > 2 else None
            inertia = geometry[3] if len(geometry) > 3 else None

            def upload_ghost(first, s_rows, g_rows, m_rows, i_rows):
                from .batch import SIDES, MASS, INERTIA
                world_batch.upload(SIDES, s_rows, first=first)
                world_batch.upload_geom_type(g_rows, first=first)
                world_batch.upload(MASS, m_rows, first=first)
                world_batch.upload(INERTIA, i_rows, first=first)
            self.exchange.share_geometry(upload_ghost, sides, gtype, mass, inertia)
        if self.collide:
            self.exchange.prime()

    # -- one tick ------------------------------------------------------------------------------------------
    def tick(self, h, check=None):
        """step + pack + exchange; check (collide mode): run the safe-zone test in this tick"""
        ex = self.exchange
        if ex is None:
            self.w.step(h, 1)
            return
        ex.before_step()
        if self.collide:
            self.w.chunk_tick(h, bool(check))
        else:
            self.w.step(h, 1)
        ex.pack()
        ex.exchange(check_ghosts=self.collide and bool(check))

    def exact_tick(self, h):
        """one tick on the exact path: the pair search reads the ghost slots, islands are solved by their own
        kernels, so the boundary rows are gathered after the tick instead of packed inside the step kernel"""
        ex = self.exchange
        ex.drain()
        # (an island that spans two ranks is reported by the exact tick, DMX_ECROSS; migrating it to one owner is the C loop's
        #  job -- csrc/dmx_shard.cpp, bound by CShardedStepper below)
        ex.before_step(fused=False)
        self.w.exact_tick(h)
        ex.pack(fused=False)
        ex.exchange()

    # -- the loop ------------------------------------------------------------------------------------------
    def run(self, h, nsteps):
        if self.exchange is None:
            self.w.step(h, nsteps)          # the C loop: no per-tick host work
            return
        if self.collide and self.lazy and self.graph is None and not self.exchange_every_tick:
            self._run_lazy(h, nsteps)
            return
        if self.collide:
            self._run_chunks(h, nsteps)
            return
        if self.graph is not None:
            reps, nsteps = divmod(nsteps, self.graph_steps)
            for _ in range(reps):
                self._replay()
        for _ in range(nsteps):
            self.tick(h)

    def _run_chunks(self, h, nsteps, begun=None):
        """the collision-checked loop, one closed chunk after another; `begun` = (exact_only, ballistic) of a chunk the
        caller has already begun (and OR-ed over the ranks)"""
        remaining = nsteps
        while remaining > 0:
            k = self.chunk
            if self.graph is not None:
                k = max(1, k // self.graph_steps) * self.graph_steps + 1      # whole replays + the tested last tick
            remaining -= self._chunk(h, min(remaining, k), begun)
            begun = None

    # -- lazily closed ballistic chunks ----------------------------------------------------------------------
    def _begin(self):
        """begin a chunk on every rank; -> (exact_only, ballistic), OR-ed / AND-ed over the ranks"""
        ex = self.exchange
        ex.drain()                           # zones, snapshot and flag reset see the last exchange's ghost rows
        exact_only, ballistic = self.w.chunk_begin()
        # every decision that shapes the loop (path, chunk length, exchange cadence) is taken on flags OR-ed over the
        # ranks, so all ranks issue the same sequence of collectives
        exact_only, not_ballistic = ex.ops.any_rank([exact_only, not ballistic], ex.group)
        return exact_only, not not_ballistic

    def _run_lazy(self, h, nsteps):
        ex, w = self.exchange, self.w
        remaining = nsteps
        while remaining > 0:
            if self._open is None:
                exact_only, ballistic = self._begin()
                if exact_only or not ballistic:
                    # crowded bodies, pending forces or bent paths somewhere: this stretch goes chunk by chunk
                    k = min(remaining, self.chunk)
                    self._run_chunks(h, k, begun=(exact_only, ballistic))
                    remaining -= k
                    continue
                self._open = {"ticks": 0, "budget": self.chunk, "segs": [], "checked": False}
            oc = self._open
            k = min(remaining, oc["budget"] - oc["ticks"])
            closes = oc["ticks"] + k >= oc["budget"]
            first = oc["ticks"] == 0
            if ex.fused:
                ex.ops.disarm_pack()
            if closes:
                # the chunk's last tick carries the zone test and the chunk's one exchange
                if k > 1:
                    w.chunk_ticks(h, k - 1, first, False)
                self.tick(h, check=True)
                oc["checked"] = True
            else:
                w.chunk_ticks(h, k, first, False)
            if oc["segs"] and oc["segs"][-1][0] == h:
                oc["segs"][-1][1] += k
            else:
                oc["segs"].append([h, k])
            oc["ticks"] += k
            remaining -= k
            if closes:
                self.settle()

    def settle(self):
        """Close the chunk run() may have left open: test, exchange, flag read and flag all-reduce; on a violation
        anywhere every rank rolls back to the chunk's start and replays its calls chunk by chunk."""
        oc, self._open = self._open, None
        if oc is None:
            return
        ex, w = self.exchange, self.w
        if not oc["checked"]:
            # closed before its length: the poses after its last tick inside their zones prove the ticks before
            # (straight horizontal lines, convex zones); the boundary rows go out by an explicit gather
            ex.ops.check_active(0, self.L.n_active)
            ex.before_step(fused=False)
            ex.pack(fused=False)
            ex.exchange(check_ghosts=True)
        ex.drain()
        violated, warn_here = w.chunk_end()
        violated, warn = ex.ops.any_rank([violated, warn_here], ex.group)
        if not violated:
            w.chunk_commit(oc["ticks"], refresh_zones=warn_here)
            if not warn and oc["ticks"] >= self.chunk:
                self.chunk = min(2 * self.chunk, self.CHUNK_MAX)
            return
        w.chunk_rollback()
        self.chunk = self.CHUNK_MIN
        for h, k in oc["segs"]:
            self._run_chunks(h, k)

    def _replay(self):
        self.exchange.ops.drain()       # eager exchanges still in flight use the ring too: let them finish first
        self.graph.replay()

    def _fast_ticks(self, h, k, ballistic):
        """k ticks of one chunk.

        Ballistic chunks (no ground plane, gravity along y: every body, ghosts included, moves on a straight horizontal
        line) are proven by the test at their first and last tick alone, and nothing else reads the ghost slots inside a
        chunk -- so only the chunk's last tick exchanges (`exchange_every_tick` forces the per-tick exchange anyway).
        Otherwise every tick is tested and exchanged."""
        if ballistic and not self.exchange_every_tick:
            if self.exchange.fused:
                self.exchange.ops.disarm_pack()
            if k > 1:
                self.w.chunk_ticks(h, k - 1, True, False)      # the library fuses them ticks_per_launch at a time
            self.tick(h, check=True)
            return
        g = self.graph_steps if (self.graph is not None and self.ballistic_graph == ballistic) else 0
        reps = (k - 1) // g if g else 0          # keep at least one eager tick: the chunk's last tick is always tested
        for _ in range(reps):
            self._replay()
        done = reps * g
        for s in range(done, k):
            self.tick(h, check=(not ballistic) or s == 0 or s == k - 1)

    def _chunk(self, h, k, begun=None):
        """Run up to k ticks as one chunk; returns the ticks actually advanced."""
        ex, w, ops = self.exchange, self.w, self.exchange.ops
        for attempt in range(3):
            exact_only, ballistic = begun if (begun is not None and attempt == 0) else self._begin()
            if exact_only or attempt == 2:
                break                            # crowded bodies or pending forces somewhere: everyone steps exactly
            self._fast_ticks(h, k, ballistic)
            ex.drain()
            violated, warn_here = w.chunk_end()
            violated, warn = ops.any_rank([violated, warn_here], ex.group)
            if not violated:
                w.chunk_commit(k, refresh_zones=warn_here)
                if not warn and k >= self.chunk:
                    self.chunk = min(2 * self.chunk, self.CHUNK_MAX)
                return k
            # some body on some rank left its zone: every rank returns to the chunk's start (ghost slots included)
            w.chunk_rollback()
            self.chunk = self.CHUNK_MIN
            k = min(k, self.CHUNK_MIN)
        for _ in range(k):
            self.exact_tick(h)
        return k

    def drain(self):
        if self.exchange is not None:
            self.exchange.drain()

    def close(self):
        """settle the open chunk, wait for the exchange, detach the batch from the send buffers"""
        if self.exchange is not None:
            self.settle()
            self.exchange.drain()
            if self.exchange.fused:
                self.exchange.ops.disarm_pack()

    def capture(self, h, steps_per_graph, stream, ring=2):
        """Capture `steps_per_graph` ticks (kernels, the RCCL all-gather and the stream choreography) into one
        HIP graph so a replay costs one host call; returns False (and stays eager) if capture is not possible.
        `ring` send buffers: 2 measured best (27.7 us/tick at 1 Mi bodies against 28-33 for deeper rings, which
        lengthen the graph's side branch -- profiles/r01_exchange_ring_ab.txt)."""
        if self.exchange is None:
            return False
        ops = self.exchange.ops
        try:
            self.settle()
            ops.drain()
            torch.cuda.synchronize()
            self.exchange.set_ring(ring)
            ballistic = None
            if self.collide:
                _, ballistic = self.w.chunk_begin()        # buffers exist before the capture; nothing is advanced
                if ballistic and not self.exchange_every_tick:
                    return False                           # one exchange per chunk: nothing worth capturing
                # a graph bakes the slab's address in: from here on the state must stay in one slab, so chunks keep
                # their rollback snapshot by copy instead of by ping-pong
                from .batch import SNAPSHOT_COPY
                self.w.set_snapshot_mode(SNAPSHOT_COPY)
            torch.cuda.synchronize()
            ops.forget()                               # no event edges from outside the capture
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for s in range(steps_per_graph):
                    self.tick(h, check=(not ballistic) or s == 0)
                ops.drain()                            # the side stream rejoins before the capture ends
            ops.forget()
            self.graph, self.graph_steps, self.ballistic_graph = g, steps_per_graph, ballistic
            return True
        except Exception as e:      # noqa: BLE001 -- capture support varies; eager is always correct
            self.graph = None
            ops.forget()
            with contextlib.suppress(Exception):
                torch.cuda.synchronize()
            print(f"[shard] graph capture unavailable ({type(e).__name__}: {e}); running eagerly", flush=True)
            return False


# ---------------------------------------------------------------------------------------------------------------------
# The same loop behind the C ABI (include/dmx_shard.h, csrc/dmx_shard.cpp): what a C host calls, and what bench.py's N > 1
# runs use.  This class only binds it.  The classes above stay as the index-logic reference that the CPU tests drive over
# gloo with host arrays standing in for the device batch (tests/test_shard_gloo.py); on a GPU the two are held against each
# other and against the oracle (tests/test_gpu_shard_abi.py).
# ---------------------------------------------------------------------------------------------------------------------
class CShardedStepper:
    """dmxShardCreate* / dmxShardRun / dmxShardSettle / dmxShardDestroy.

    collectives="rccl": the library's own RCCL binding (ncclAllGather on the side stream); the unique id comes from rank 0
    through the torch.distributed group that is already up (any backend).  collectives="staged": the two collectives are
    injected as callbacks that stage through host memory over that group -- several ranks can then share one GPU, which RCCL
    does not allow (rehearsals and tests; never a measurement)."""

    def __init__(self, world_batch, layout, rank, world_size, collectives="rccl", group=None):
        import ctypes as C
        self.w, self.L, self.rank, self.world, self.group = world_batch, layout, rank, world_size, group
        self.lib = world_batch.lib
        self.h = C.c_void_p()
        self._keep = None
        if collectives == "rccl":
            # Every rank first proves, on its own, what can fail locally -- librccl loads, the device answers, an id can be made
            # (dmxShardRcclUniqueId does all three; only rank 0's id is used) -- and the ranks AGREE on the outcome over the
            # torch.distributed group that is already up, BEFORE anybody enters an RCCL call: a rank that cannot load the library
            # must not leave its peers waiting in a broadcast or in ncclCommInitRank.  Either every rank goes on, or every rank
            # raises here together (bench.py then falls back on the Python loop on all of them).  What this does not cover: a
            # failure INSIDE ncclCommInitRank or inside the priming exchange on one rank only (RCCL's own bring-up; never seen,
            # never run on more than one GPU by the builder) -- the other ranks would wait there.
            ident = (C.c_char * 128)()
            rc_local = self.lib.dmxShardRcclUniqueId(ident)
            if world_size > 1:
                on_gpu = dist.get_backend(group) == "nccl"
                flag = torch.tensor([1 if rc_local != 0 else 0], dtype=torch.int32)
                if on_gpu:
                    flag = flag.cuda()
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
                if int(flag.item()) != 0:
                    from .batch import DmxError
                    raise DmxError("RCCL is not usable on " + ("this rank" if rc_local != 0 else "another rank") +
                                   f" (dmxShardRcclUniqueId code {rc_local}); every rank stops here together", rc_local or 1)
                t = torch.frombuffer(bytearray(ident.raw), dtype=torch.uint8).clone()
                if on_gpu:
                    t = t.cuda()
                dist.broadcast(t, src=0, group=group)
                ident = (C.c_char * 128).from_buffer_copy(bytes(t.cpu().numpy().tobytes()))
            else:
                _check_rc(rc_local, "dmxShardRcclUniqueId")
            rc_create = self.lib.dmxShardCreateRccl(C.byref(self.h), world_batch.h, layout.side, layout.rows, layout.spare, rank, world_size, ident)
            if world_size > 1:
                # ... and on whether the communicator and the priming exchange came up everywhere (a rank whose create returned an
                # error AFTER the collectives it shares with its peers -- allocation failures -- is met here, not in the tick loop)
                flag = torch.tensor([1 if rc_create != 0 else 0], dtype=torch.int32)
                if dist.get_backend(group) == "nccl":
                    flag = flag.cuda()
                dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
                if int(flag.item()) != 0 and rc_create == 0:
                    self.lib.dmxShardDestroy(self.h)
                    self.h = C.c_void_p()
                    from .batch import DmxError
                    raise DmxError("dmxShardCreateRccl failed on another rank; every rank stops here together", 1)
            _check_rc(rc_create, "dmxShardCreateRccl")
        else:
            self._keep = self._staged_collectives()
            _check_rc(self.lib.dmxShardCreate(C.byref(self.h), world_batch.h, layout.side, layout.rows, layout.spare, rank, world_size,
                                              C.byref(self._keep[0])), "dmxShardCreate")

    def _staged_collectives(self):
        """(dmxCollectives struct, callbacks kept alive): all-gather and max-all-reduce through host memory over the group"""
        import ctypes as C
        AG = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)
        AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32), C.c_int)
        hip = C.CDLL(None)              # the HIP runtime is already in the process (torch / the library)
        world, group = self.world, self.group

        def all_gather(_ctx, send, recv, nbytes, stream):
            try:
                if hip.hipStreamSynchronize(C.c_void_p(stream)) != 0:
                    return 1
                mine = (C.c_ubyte * nbytes)()
                if hip.hipMemcpy(mine, C.c_void_p(send), C.c_size_t(nbytes), 2) != 0:      # device to host
                    return 1
                h = torch.frombuffer(bytearray(mine), dtype=torch.uint8)
                o = torch.empty(world * nbytes, dtype=torch.uint8)
                dist.all_gather_into_tensor(o, h, group=group)
                buf = o.numpy().tobytes()
                return 0 if hip.hipMemcpy(C.c_void_p(recv), buf, C.c_size_t(len(buf)), 1) == 0 else 1      # host to device
            except Exception as e:      # noqa: BLE001 -- an exception must not unwind through the C caller
                print(f"[shard] staged all-gather failed: {type(e).__name__}: {e}", flush=True)
                return 1

        def all_reduce_max(_ctx, vals, n):
            try:
                t = torch.tensor([vals[i] for i in range(n)], dtype=torch.int32)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
                for i in range(n):
                    vals[i] = int(t[i])
                return 0
            except Exception as e:      # noqa: BLE001
                print(f"[shard] staged all-reduce failed: {type(e).__name__}: {e}", flush=True)
                return 1

        class Coll(C.Structure):
            _fields_ = [("ctx", C.c_void_p), ("all_gather", AG), ("all_reduce_max", AR)]
        ag, ar = AG(all_gather), AR(all_reduce_max)
        return Coll(None, ag, ar), ag, ar

    def rccl_info(self):
        """{"librccl": path the library's own binding loaded ("" if none), "comm_up": ncclCommInitRank brought a communicator up}"""
        import ctypes as C
        path = C.create_string_buffer(512)
        up = C.c_int(0)
        self.lib.dmxShardRcclInfo(self.h, path, 512, C.byref(up))
        return {"librccl": path.value.decode(errors="replace"), "comm_up": bool(up.value)}

    def run(self, h, nsteps):
        _check_rc(self.lib.dmxShardRun(self.h, float(h), int(nsteps)), "dmxShardRun")

    def settle(self):
        _check_rc(self.lib.dmxShardSettle(self.h), "dmxShardSettle")

    def drain(self):
        self.settle()

    def stats(self):
        import ctypes as C
        out = (C.c_int64 * 6)()
        _check_rc(self.lib.dmxShardStats(self.h, out), "dmxShardStats")
        return dict(zip(("exchanges", "committed", "rolled_back", "exact_ticks", "adopted", "retired"), [int(v) for v in out]))

    def close(self):
        if self.h:
            self.settle()
            _check_rc(self.lib.dmxShardDestroy(self.h), "dmxShardDestroy")
            self.h = None


def _check_rc(rc, what):
    if rc != 0:
        from .batch import DmxError
        raise DmxError(f"{what} failed with code {rc}", rc)
