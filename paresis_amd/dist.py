"""Multi-GPU sharding of the membrane-position loop (main.py:63 of the reference is a plain serial loop).

One process per GPU (torchrun).  Positions are independent given seed(pointNum), so rank r takes positions
r, r+G, r+2G, ... with no data-path collective; the only exchange is the final gather of the detector images onto
rank 0 with torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU for the tests).
"""
import datetime
import os
import time

import torch
import torch.distributed as td


class DistError(RuntimeError):
    """A condition under which the ranks can no longer be trusted to issue the same collectives: the caller must leave the
    process with a non-zero code (os._exit) instead of trying another collective on the same communicator."""


def timeout_s():
    """Upper bound for any single wait on the other ranks (PSX_DIST_TIMEOUT_S, default 300 s): the process-group timeout
    and the bounded wait of PositionGatherer.finish()."""
    return float(os.environ.get("PSX_DIST_TIMEOUT_S", "300"))


def local_device(backend, rank, world, n_devices=None):
    """The GPU of this rank under `nccl` (= RCCL): LOCAL_RANK, one rank per GPU.  More ranks than GPUs is refused here,
    before any collective -- RCCL would only report `ncclInvalidUsage: Duplicate GPU detected` from the first barrier (or
    hang).  Under `gloo` several ranks may share a GPU (rehearsals): LOCAL_RANK modulo the device count."""
    local = int(os.environ.get("LOCAL_RANK", rank))
    if n_devices is None:
        n_devices = torch.cuda.device_count()
    if backend == "nccl":
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        if local_world > n_devices or local >= n_devices:
            raise SystemExit("paresis_amd.dist: %d ranks on this node but %d GPU(s) visible: the nccl backend needs one GPU per "
                             "rank (use --backend gloo to rehearse several ranks on one GPU)" % (local_world, n_devices))
        return local
    return local % max(1, n_devices)


def init(backend=None):
    """Returns (rank, world).  A plain `python` launch (no RANK in the environment) is world size 1."""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return 0, 1
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if not td.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            dev = local_device(backend, rank, world)
            torch.cuda.set_device(dev)
            kw["device_id"] = torch.device("cuda", dev)       # RCCL otherwise guesses the device from the global rank
        elif torch.cuda.is_available():
            torch.cuda.set_device(local_device(backend, rank, world))
        td.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s()), **kw)
    return rank, world


def any_rank_failed(failed):
    """Collective decision: True on EVERY rank when `failed` is true on any.  To be called at a point every rank reaches
    whatever happened to it (e.g. after a try/except around a phase WITHOUT collectives), so that all ranks then take the
    same branch; a rank deciding on its own would issue different collectives from the others and hang them."""
    if not (td.is_available() and td.is_initialized()):
        return bool(failed)
    flag = torch.tensor([1 if failed else 0], dtype=torch.int32, device=_dev())
    td.all_reduce(flag, op=td.ReduceOp.MAX)
    return bool(int(flag.item()))


def agree_on_overlap(gatherer, inject_failure=False):
    """The ranks decide TOGETHER whether the round-by-round gather can be used: every rank allocates its buffers
    (PositionGatherer.prepare: no collective inside), the outcomes meet in one all_reduce(MAX), and either all ranks return
    True or all return False (one gather at the end instead).  inject_failure: test hook, this rank pretends its allocation
    failed."""
    failed = False
    try:
        if inject_failure:
            raise MemoryError("injected allocation failure (test)")
        gatherer.prepare()
    except (RuntimeError, MemoryError) as exc:
        print("rank %d: no buffers for the overlapped gather (%s: %s)" % (gatherer.rank, type(exc).__name__, exc))
        failed = True
    return not any_rank_failed(failed)


def barrier():
    if td.is_available() and td.is_initialized():
        td.barrier()


def broadcast_object(obj, rank, world, src=0):
    """The same small Python object on every rank (rank `src`'s)."""
    if world == 1:
        return obj
    box = [obj if rank == src else None]
    td.broadcast_object_list(box, src=src, device=_dev())
    return box[0]


def finish():
    if td.is_available() and td.is_initialized():
        td.barrier()


def my_positions(n_positions, rank, world):
    """Static strided partition: rank r owns r, r+world, ..."""
    return list(range(rank, n_positions, world))


def owner(position, world):
    return position % world


last_gather = {}          # what the last gather_positions() moved: {"packed": bool, "wire_bytes": bytes received by dst}


class _CountsWire:
    """One rank's contribution as bytes: [n 16-bit counts | pad to 8 | exception count, 0 (int32) | cap x (index, count)
    (int32)].  Counts >= 65535 are escaped into the exception table (rare: caustic peaks), so the packing is lossless for
    any integer image below 2^24 whose bright pixels fit the table: one per 16 pixels -- 2.5 bytes per pixel on the wire.  (One per
    256 until round 6: the 4096^2 XML experiment behind a sphere membrane has caustics of 8 x its mean of 30000 counts, 3.3 % of
    its pixels above 65534 -- every run of it fell back to float32, gpurun_out/r6s28.)"""

    def __init__(self, n, device, like=None):
        self.n = n
        self.cap = max(64, n // 16)
        self.off = (2 * n + 7) // 8 * 8
        self.bytes = torch.empty(self.off + 8 + 8 * self.cap, dtype=torch.uint8, device=device) if like is None else like
        self.counts = self.bytes[:2 * n].view(torch.int16)
        self.head = self.bytes[self.off:self.off + 8].view(torch.int32)
        self.exc = self.bytes[self.off + 8:].view(torch.int32)

    def pack(self, img, index0, flag):
        """Counts of img into slots [index0, index0 + img.numel()); flag raised when img cannot be packed."""
        out = self.counts[index0:index0 + img.numel()]
        if img.device != out.device:                      # gloo rehearsal of images computed on the GPU
            img = img.to(out.device)
        img = img.contiguous().view(-1)
        if img.is_cuda:                                   # HIP kernel; the host restatement below serves the gloo rehearsal
            from . import ops
            ops.pack_counts(img, out, index0, self.exc, self.head[:1], flag)
            return
        q = img.clamp(0, 16777216).to(torch.int32)
        if not bool((q.to(torch.float32) == img).all()):
            flag.fill_(1)
        big = torch.nonzero(q >= 65535).view(-1)
        e0 = int(self.head[0])
        self.head[0] = e0 + big.numel()
        if e0 + big.numel() > self.cap:
            flag.fill_(1)
        else:
            tab = self.exc.view(-1, 2)
            tab[e0:e0 + big.numel(), 0] = (big + index0).to(torch.int32)
            tab[e0:e0 + big.numel(), 1] = q[big]
        out.copy_(q.clamp(max=65535).to(torch.int16))     # two's complement wrap: the uint16 bit pattern

    def unpack(self, out=None):
        if self.bytes.is_cuda:
            from . import ops
            out = torch.empty(self.n, dtype=torch.float32, device=self.bytes.device) if out is None else out
            return ops.unpack_counts(self.counts, out, self.exc, self.head[:1])
        out = (self.counts.to(torch.int32) & 0xFFFF).to(torch.float32)
        m = min(int(self.head[0]), self.cap)
        tab = self.exc.view(-1, 2)[:m]
        out[tab[:, 0].to(torch.int64)] = tab[:, 1].to(torch.float32)
        return out


def gather_positions(results, n_positions, rank, world, dst=0, to_host=True, pack=True, force_collectives=False, timeout=None):
    """results: {position: tuple of tensors} computed on this rank.  Returns on `dst` a dict with every position
    (tensors on the host, or left in `dst`'s HBM with to_host=False), {} elsewhere.

    ONE fixed-shape gather for the whole run: every rank contributes the Sample/Reference stacks of its positions
    [rounds][2][nbins][n][n] (slot t = position t*world + rank, zeros when it has none) -- 7 point-to-point transfers into
    rank `dst` over xGMI, no ring.  `dst`'s inbound links bound it, so with pack=True detector images that are photon
    counts (integers: the shot-noise output) cross as 16-bit integers plus a short table of the pixels above 65534, half
    the bytes, and are widened again on `dst`; the packing checks every pixel and any rank finding something else makes ALL
    ranks send float32, so what arrives is bit for bit what was computed either way.  force_collectives: take the
    collective path in a one-rank group too (the tests drive RCCL itself that way on a one-GPU box)."""
    host = lambda tup: tuple(t.detach().cpu() if isinstance(t, torch.Tensor) and to_host else t for t in tup)
    last_gather.clear()
    if world == 1 and not force_collectives:
        return {p: host(v) for p, v in results.items()}
    out = {}
    dev = _dev()
    rounds = (n_positions + world - 1) // world
    proto = None
    for v in results.values():
        proto = v
        break
    shape_t = torch.zeros(4, dtype=torch.int64, device=dev)
    if proto is not None:
        shape_t[:3] = torch.tensor(proto[0].shape, dtype=torch.int64)
        shape_t[3] = 1
    td.all_reduce(shape_t, op=td.ReduceOp.MAX)          # ranks without work learn the stack shape
    shape = tuple(int(v) for v in shape_t[:3])
    per_img = shape[0] * shape[1] * shape[2]
    n_all = rounds * 2 * per_img
    wire = None
    if pack and n_all < 2 ** 31:
        wire = _CountsWire(n_all, dev)
        wire.head.zero_()
        flag = torch.zeros(1, dtype=torch.int32, device=dev)
        for t in range(rounds):
            p = t * world + rank
            if p in results:
                wire.pack(results[p][0], (2 * t) * per_img, flag)
                wire.pack(results[p][1], (2 * t + 1) * per_img, flag)
            else:
                wire.counts[2 * t * per_img:(2 * t + 2) * per_img].zero_()
        td.all_reduce(flag, op=td.ReduceOp.MAX)          # one image that is not photon counts anywhere: everybody sends float32
        if int(flag.item()):
            wire = None
    packed = wire is not None
    if packed:
        mine = wire.bytes
    else:
        mine = torch.empty((rounds, 2) + shape, dtype=torch.float32, device=dev)
        for t in range(rounds):
            p = t * world + rank
            if p in results:
                mine[t, 0].copy_(results[p][0])
                mine[t, 1].copy_(results[p][1])
            else:
                mine[t].zero_()
    bucket = [torch.empty_like(mine) for _ in range(world)] if rank == dst else None
    td.gather(mine, bucket, dst=dst)
    last_gather.update(packed=packed, wire_bytes=(world - 1) * mine.numel() * mine.element_size() if rank == dst else 0)
    if rank == dst:
        if to_host:
            bucket = [b.cpu() for b in bucket]
        if packed:
            bucket = [_CountsWire(n_all, b.device, like=b).unpack().view((rounds, 2) + shape) for b in bucket]
        for r in range(world):
            for t in range(rounds):
                q = t * world + r
                if q < n_positions:
                    out[q] = (bucket[r][t, 0], bucket[r][t, 1])
    extras = move_extras(results, rank, world, dst, to_host, timeout=timeout) if n_positions > 0 else None   # Propag / White / Dx,Dy: position 0 only
    if extras is not None:
        out[0] = extras
    return out if rank == dst else {}


class PackedPositions(dict):
    """{position: (Sample, Reference[, extras])} on a sink whose images stay in HBM (to_host=False): the stacks are kept as they
    crossed -- 16-bit photon counts + the table of brighter pixels, lossless -- and widened to float32 the first time a
    position is read; plain dict entries (position 0 with its extras) are served as they are.  A 64-position run otherwise
    ends with 2 x 64 widening launches on the sink that nobody may ever look at (0.8 ms of rank 0's 10 ms share: bench.py
    rank_share, round 5); main.run's host copies are widened on the host in either case."""

    def __init__(self, packed, per_img, shape, ready, owner=None):
        super().__init__(ready)
        self._packed, self._per_img, self._shape = dict(packed), per_img, tuple(shape)
        # the packed bytes live in the gatherer's pooled receive buckets, which its next run overwrites: a position still packed
        # when PositionGatherer.reset() is called can no longer be read -- and says so instead of widening the next run's data
        self._owner, self._generation = owner, getattr(owner, "generation", 0)

    def _widen(self, p):
        if self._owner is not None and self._owner.generation != self._generation:
            raise RuntimeError("PackedPositions: position %d was still packed when its gatherer was reset for another run; its "
                               "receive buffer has been reused (read, or copy(), a result before PositionGatherer.reset())" % p)
        b = self._packed.pop(p)
        img = _CountsWire(2 * self._per_img, b.device, like=b).unpack().view((2,) + self._shape)
        super().__setitem__(p, (img[0], img[1]))

    def __getitem__(self, p):
        if p in self._packed and not super().__contains__(p):
            self._widen(p)
        return super().__getitem__(p)

    def __setitem__(self, p, v):
        self._packed.pop(p, None)
        super().__setitem__(p, v)

    def __delitem__(self, p):
        if p in self._packed and not super().__contains__(p):
            del self._packed[p]
        else:
            self._packed.pop(p, None)
            super().__delitem__(p)

    def __contains__(self, p):
        return super().__contains__(p) or p in self._packed

    def __len__(self):
        return len(set(super().keys()) | set(self._packed))

    def __iter__(self):
        return iter(sorted(set(super().keys()) | set(self._packed)))

    def keys(self):
        return list(iter(self))

    def values(self):
        return [self[p] for p in self]

    def items(self):
        return [(p, self[p]) for p in self]

    def get(self, p, default=None):
        return self[p] if p in self else default

    _missing = object()

    def pop(self, p, default=_missing):
        if p in self:
            v = self[p]
            del self[p]
            return v
        if default is PackedPositions._missing:
            raise KeyError(p)
        return default

    def setdefault(self, p, default=None):
        if p not in self:
            self[p] = default
        return self[p]

    def update(self, *args, **kw):
        for p, v in dict(*args, **kw).items():
            self[p] = v

    def copy(self):
        """A plain dict of float32 stacks (every position widened): what to keep across PositionGatherer.reset()."""
        return {p: self[p] for p in self}

    def clear(self):
        self._packed.clear()
        super().clear()

    def popitem(self):
        for p in reversed(self.keys()):
            return p, self.pop(p)
        raise KeyError("popitem(): PackedPositions is empty")

    def __repr__(self):
        return "PackedPositions(%d positions, %d still packed)" % (len(self), len(self._packed))

    def stack_numel(self):
        """Elements of one position's pair of stacks (no widening)."""
        return 2 * self._per_img


class PositionGatherer:
    """gather_positions() in pieces, overlapped with the computation: as soon as a rank has finished its position of
    round t (positions t*world .. t*world + world - 1, one per rank) it packs the two stacks and issues the gather of that
    round WITHOUT waiting for it (async_op); RCCL moves round t while the GPU computes round t + 1, and only the last
    round's transfer is exposed.  With the work queue of the Fresnel plan on (psx_fresnel_plan_work_queue), the copy
    kernels of the transfer cost the computation a few per cent (tools/contention_probe.py).

    Every round crosses as 16-bit photon counts (see gather_positions); a round that cannot (non-integer images, too many
    bright pixels) is flagged, and finish() then repeats the whole gather in float32 -- all ranks together, after one
    all_reduce of the flag.  Every rank issues the same collectives in the same order whatever positions it owns."""

    def __init__(self, n_positions, rank, world, dst=0, to_host=False, shape=None, force_collectives=False):
        """shape: (nbins, n0, n1) of a detector stack -- known to every rank from the experiment; a rank that owns no
        position cannot learn it any other way (a collective for it would come in a different order on the other ranks)."""
        self.P, self.rank, self.world, self.dst, self.to_host = n_positions, rank, world, dst, to_host
        self.rounds = (n_positions + world - 1) // world
        self.results, self.work, self.wires, self.buckets = {}, [], [], []
        self.local = world == 1 and not force_collectives             # nothing to move
        self.force = force_collectives
        self.shape = tuple(int(v) for v in shape) if shape is not None else None
        self.flag = None
        self.next_round = 0
        self.pool = None               # (wires, buckets) of every round, allocated by prepare()
        self.issued = 0                # collectives issued so far: an exception after the first one is not recoverable
        self.generation = 0            # runs started on these buffers: a PackedPositions of an earlier run refuses to widen

    def prepare(self):
        """Allocates the wire buffer of every round and, on dst, the `world` receive buckets per round (~2 GiB for 64
        positions of 2048^2 on 8 ranks) WITHOUT issuing a collective.  This is the step most likely to fail on one rank
        only (dst alone holds the buckets): wrap it in try/except and put the outcome through any_rank_failed() -- every rank
        can then fall back to the one-gather form together.  Without prepare() the buffers are allocated round by round."""
        if self.local or self.pool is not None:
            return
        if self.shape is None:
            raise ValueError("PositionGatherer.prepare: needs the stack shape (constructor argument)")
        dev = _dev()
        per_img = self.shape[0] * self.shape[1] * self.shape[2]
        wires = [_CountsWire(2 * per_img, dev) for _ in range(self.rounds)]
        buckets = [[torch.empty_like(w.bytes) for _ in range(self.world)] if self.rank == self.dst else None for w in wires]
        self.flag = torch.zeros(1, dtype=torch.int32, device=dev)
        self.pool = (wires, buckets)

    def reset(self):
        """Ready for another run over the same positions with the buffers of prepare() (the bench: an untimed run first).
        The receive buckets are reused: positions of the PREVIOUS run's result that were never read (dist.PackedPositions
        widens on first access) can no longer be read -- the result raises instead of widening the next run's bytes; keep
        `result.copy()` if it is wanted."""
        self.generation += 1
        self.results, self.work, self.wires, self.buckets = {}, [], [], []
        self.next_round = 0
        self.issued = 0
        if self.flag is not None:
            self.flag.zero_()

    def _issue(self, t):
        dev = _dev()
        per_img = self.shape[0] * self.shape[1] * self.shape[2]
        wire = self.pool[0][t] if self.pool is not None else _CountsWire(2 * per_img, dev)
        wire.head.zero_()
        p = t * self.world + self.rank
        if p in self.results:
            wire.pack(self.results[p][0], 0, self.flag)
            wire.pack(self.results[p][1], per_img, self.flag)
        else:
            wire.counts.zero_()
        if self.pool is not None:
            bucket = self.pool[1][t]
        else:
            bucket = [torch.empty_like(wire.bytes) for _ in range(self.world)] if self.rank == self.dst else None
        self.issued += 1
        try:
            self.work.append(td.gather(wire.bytes, bucket, dst=self.dst, async_op=True))
        except RuntimeError as exc:           # a peer is gone (gloo: connection reset; RCCL: communicator aborted)
            raise DistError("PositionGatherer: gather of round %d could not be issued on rank %d: %s" % (t, self.rank, exc)) from exc
        self.wires.append(wire)
        self.buckets.append(bucket)

    def add(self, p, images):
        """images: the tuple computeSampleAndReferenceImages returned for position p (owned by this rank).  Only position 0's
        extras (Propag, White, the RT chain's displacement and dark-field maps) are ever gathered: the other positions keep
        their two detector stacks and nothing else alive (a scattering sample returns a fresh study-grid map per position)."""
        self.results[p] = images if p == 0 else tuple(images[:2])
        if self.local:
            return
        if self.shape is None:
            self.shape = tuple(images[0].shape)
        if tuple(images[0].shape) != self.shape or tuple(images[1].shape) != self.shape:
            raise ValueError("PositionGatherer: stack of shape %s, expected %s" % (tuple(images[0].shape), self.shape))
        if self.flag is None:
            self.flag = torch.zeros(1, dtype=torch.int32, device=_dev())
        while self.next_round < self.rounds and self.next_round * self.world + self.rank <= p:
            self._issue(self.next_round)
            self.next_round += 1

    def _move_extras(self, timeout):
        if self.P <= 0:
            return None
        return move_extras(self.results, self.rank, self.world, self.dst, self.to_host, timeout=timeout)

    def finish(self, timeout=None):
        """Waits for the rounds in flight -- at most `timeout` seconds each (default: timeout_s()), then DistError: a rank
        that died or left the sequence of collectives must not hang the others until the process-group timeout -- and returns
        on dst {position: images} (every position), {} elsewhere."""
        timeout = timeout_s() if timeout is None else float(timeout)
        host = lambda tup: tuple(t.detach().cpu() if isinstance(t, torch.Tensor) and self.to_host else t for t in tup)
        last_gather.clear()
        if self.local:
            return {p: host(v) for p, v in self.results.items()}
        if self.shape is None:
            raise ValueError("PositionGatherer: a rank without positions needs the stack shape (constructor argument)")
        if self.flag is None:
            self.flag = torch.zeros(1, dtype=torch.int32, device=_dev())
        while self.next_round < self.rounds:                            # rounds this rank has no position in
            self._issue(self.next_round)
            self.next_round += 1
        try:
            for t, w in enumerate(self.work):
                # polled, not wait(): gloo's wait() blocks the host without bound and nccl's only orders the streams (the
                # host would then hang in the first .item() below instead).  The bound is per round -- a long healthy run
                # with many rounds in flight is not declared dead by the sum of its transfers -- and the poll backs off
                # from 50 us to 1 ms, so a waiting rank does not keep a host core from the packing / launch thread.
                deadline = time.monotonic() + timeout
                nap = 0.00005
                while not w.is_completed():
                    if time.monotonic() > deadline:
                        raise DistError("PositionGatherer.finish: the gather of round %d did not complete within %.0f s on rank "
                                        "%d (a rank failed or left the sequence of collectives)" % (t, timeout, self.rank))
                    time.sleep(nap)
                    nap = min(0.001, nap * 1.5)
                w.wait()
            td.all_reduce(self.flag, op=td.ReduceOp.MAX)
            if int(self.flag.item()):                                   # something was not photon counts: float32, all together
                # (gather_positions moves position 0's extras itself: they cross ONCE, after the decision -- ADVICE r4)
                return gather_positions(self.results, self.P, self.rank, self.world, dst=self.dst, to_host=self.to_host,
                                        pack=False, force_collectives=self.force, timeout=timeout)
            extras = self._move_extras(timeout)
        except RuntimeError as exc:           # includes DistError; a peer that died surfaces here as a transport error
            if isinstance(exc, DistError):
                raise
            raise DistError("PositionGatherer.finish: rank %d lost a peer: %s" % (self.rank, exc)) from exc
        out, lazy = {}, {}
        wire_bytes = 0
        if self.rank == self.dst:
            per_img = self.shape[0] * self.shape[1] * self.shape[2]
            for t in range(self.rounds):
                for r in range(self.world):
                    q = t * self.world + r
                    if q >= self.P:
                        continue
                    b = self.buckets[t][r]
                    wire_bytes += b.numel() if r != self.rank else 0
                    if not self.to_host and b.is_cuda:
                        lazy[q] = b              # stays in the sink's HBM as it crossed: widened when (if) somebody reads it
                        continue
                    if self.to_host:
                        b = b.cpu()
                    img = _CountsWire(2 * per_img, b.device, like=b).unpack().view((2,) + self.shape)
                    out[q] = (img[0], img[1])
            if lazy:
                out = PackedPositions(lazy, per_img, self.shape, out, owner=self)
            if extras is not None:                                       # Propag / White / Dx,Dy exist for position 0 only
                out[0] = extras
        last_gather.update(packed=True, wire_bytes=wire_bytes, overlapped=True, lazy=bool(lazy))
        return out if self.rank == self.dst else {}


def _wait_p2p(work, timeout, what, rank):
    """A point-to-point transfer under the same regime as the gather rounds: polled with a deadline, not wait() -- a sink
    waiting for a stalled owner of position 0 must not sit there until the process-group timeout."""
    if td.get_backend() == "gloo":
        # gloo's send / receive requests report completion only from inside wait(): bounded there (waitSend / waitRecv take the
        # timeout and raise when it passes)
        try:
            work.wait(datetime.timedelta(seconds=timeout))
        except RuntimeError as exc:
            raise DistError("move_extras: %s did not complete within %.0f s on rank %d: %s" % (what, timeout, rank, exc)) from exc
        return
    deadline = time.monotonic() + timeout
    nap = 0.00005
    while not work.is_completed():
        if time.monotonic() > deadline:
            raise DistError("move_extras: %s did not complete within %.0f s on rank %d" % (what, timeout, rank))
        time.sleep(nap)
        nap = min(0.001, nap * 1.5)
    work.wait()


def move_extras(results, rank, world, dst, to_host, timeout=None):
    """Position 0's full tuple (Sample, Reference, Propag, White[, Dx, Dy, DF]) on `dst`, None elsewhere.  Propag / White and
    the displacement maps exist for position 0 only (EXP:363-375, 488-498) and live on its owner, rank 0.  With dst == 0
    nothing moves.  With another sink (the straggler that computes the extras and the rank that receives everybody's images
    are then two different GPUs) they cross in ONE point-to-point transfer per tensor: a small int64 header (count, ndim and
    shape of each) first -- the sink cannot know the shapes of the RT chain's padded maps -- then float32 payloads, each
    transfer polled against `timeout` seconds (default: timeout_s()).  An owner that never computed position 0 says so in
    the header (count -1) before it raises, so the sink raises too instead of waiting.
    Every rank calls this at the same point of its sequence; ranks other than 0 and dst do nothing."""
    timeout = timeout_s() if timeout is None else float(timeout)
    host = lambda tup: tuple(t.detach().cpu() if isinstance(t, torch.Tensor) and to_host else t for t in tup)
    src = owner(0, world)
    if dst == src:
        return host(results[0]) if (rank == dst and 0 in results) else None
    dev = _dev()
    HDR = 64
    if rank == src:
        hdr = torch.zeros(HDR, dtype=torch.int64)
        if 0 not in results:
            hdr[0] = -1
            _wait_p2p(td.isend(hdr.to(dev), dst=dst), timeout, "the header to rank %d" % dst, rank)
            raise DistError("move_extras: rank %d owns position 0 but never computed it" % rank)
        tens = [t for t in results[0]]
        hdr[0] = len(tens)
        k = 1
        for t in tens:
            if not isinstance(t, torch.Tensor):          # e.g. a displacement map that was never computed: travels as None
                hdr[k] = -1
                k += 1
                continue
            hdr[k] = t.dim()
            hdr[k + 1:k + 1 + t.dim()] = torch.tensor(list(t.shape), dtype=torch.int64)
            k += 1 + t.dim()
        _wait_p2p(td.isend(hdr.to(dev), dst=dst), timeout, "the header to rank %d" % dst, rank)
        for i, t in enumerate(tens):
            if isinstance(t, torch.Tensor):
                _wait_p2p(td.isend(t.detach().to(dev, torch.float32).contiguous(), dst=dst), timeout,
                          "tensor %d of position 0 to rank %d" % (i, dst), rank)
        return None
    if rank == dst:
        hdr = torch.zeros(HDR, dtype=torch.int64, device=dev)
        _wait_p2p(td.irecv(hdr, src=src), timeout, "the header from rank %d" % src, rank)
        hdr = hdr.cpu().tolist()
        if hdr[0] < 0:
            raise DistError("move_extras: rank %d, the owner of position 0, never computed it" % src)
        out, k = [], 1
        for i in range(hdr[0]):
            nd = hdr[k]
            if nd < 0:
                out.append(None)
                k += 1
                continue
            shape = tuple(hdr[k + 1:k + 1 + nd])
            k += 1 + nd
            t = torch.empty(shape, dtype=torch.float32, device=dev)
            _wait_p2p(td.irecv(t, src=src), timeout, "tensor %d of position 0 from rank %d" % (i, src), rank)
            out.append(t)
        return host(tuple(out))
    return None


def _dev():
    if td.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")
