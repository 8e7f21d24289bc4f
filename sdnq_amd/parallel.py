"""Column-sharded (tensor-parallel) quantized Linear over the GPUs of one node: RCCL all-gather over xGMI.

The reference has no inference parallelism; this is the new capability BASELINE.json's north_star
asks for.  The path shards on OUTPUT CHANNELS: weight rows, scale[n], zero_point[n], bias[n],
svd_up[n,:] and the re-quantized row scale are all per-output-channel, while x, its row scale, the Hadamard
rotation and x @ svd_down depend only on the replicated activation.  So rank r quantizes the full activation
(cheap, M*K) and multiplies it with its N/W slab of the weight; ONE collective per layer re-assembles the output:

    y_r [M, N/W]  --ncclAllGather-->  [W, M, N/W]  --strided copy-->  y [M, N]

One process per GPU; ``torch.distributed`` backend "nccl" is RCCL on ROCm.  The message per rank is 2*M*N/W bytes
(bf16); xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a fully connected all-gather is bound by one link:
t ~ 2*M*N/W / 153 GB/s + launch latency -- at SDXL bs=1 sizes that is longer than the sharded GEMM itself, which is
why the default multi-GPU mode of bench.py is independent replicas and TP is opt-in.

``column_shard_module`` slices an ALREADY QUANTIZED layer (a loaded checkpoint): rank r takes rows [a, b) of the stored weight
-- a contiguous slab of the physical [N][K] buffer in every storage format (the transposed matmul layout, plain [N, K] /
[N, G, g], the packed sub-byte codecs, which hold whole rows) -- and the matching slices of scale / zero_point / bias /
svd_up; svd_down (shared by all channels) is replicated.  The slab's tensors are VIEWS of the full layer's parameters (no
copy) and the slab is an ordinary SDNQLinear with the reference's state_dict layout, so its output is bit-identical to columns
[a, b) of the unsharded layer for every format: all quantization statistics, the re-quantization and the low-rank term are
per output row (N / W stays a multiple of 16, utils.py:96-97).  ``column_shard_linear`` (slice a FLOAT layer, then quantize the
slab) is kept for building sharded models from float checkpoints; with SVD it derives per-shard factors, the module slicer does not.

Why the gather is not "in place": RCCL collectives move one contiguous buffer per rank, and rank r's columns of a row-major
[M, N] matrix are M strided pieces, so the gathered [W, M, N/W] buffer needs one transposing copy: ``sdnq_hip_unshard_columns``
(csrc/parallel.hip), one HBM-bound pass that also trims the padding of uneven shards.

The COPY-FREE variant (round 4, ``PeerArena`` / ``column_shard_module(..., peer=arena)``): every rank owns an arena that all ranks
of the node have mapped through IPC handles (exchanged once); rank r pushes its slab straight into columns [a_r, b_r) of EVERY
rank's row-major [M, N] output with P2P stores over xGMI (``sdnq_hip_push_columns``): no staging buffer, no RCCL call, no
re-assembly pass -- the one read of y_r feeds W writes.  The receiver chooses where it receives (a ring allocator over its arena that
never reuses memory a live tensor still views) and announces the offset per gather through a peer-mapped mailbox, so the ranks'
allocators never have to agree; completion is signalled through peer-mapped flags (no collective).  Falls back to the RCCL path
when peer mapping is unavailable.
"""
from __future__ import annotations

import torch

from .quantizer import SDNQConfig, sdnq_quantize_layer


def shard_bounds(n: int, rank: int, world: int, multiple: int = 16) -> tuple[int, int]:
    """Even split of N output channels in units of `multiple` (the reference needs N % 16 == 0 for quantized matmul)."""
    units = n // multiple
    if n % multiple or units < world:
        raise ValueError(f"N={n} cannot be column-sharded {world} ways in multiples of {multiple}")
    base, extra = divmod(units, world)
    start = (rank * base + min(rank, extra)) * multiple
    size = (base + (1 if rank < extra else 0)) * multiple
    return start, start + size


def chunk_rows(m: int, chunks: int) -> list[tuple[int, int]]:
    """Row ranges of the M-chunked gather pipeline: 32-row-aligned steps, and NO chunk shorter than 32 rows -- a shorter tail would
    take the layer's small-batch (M < 32) float branch instead of the quantized matmul (linear_int8.py:102-103) and differ from
    the unchunked layer; such a tail is merged into the chunk before it."""
    step = (-(-m // max(1, chunks)) + 31) // 32 * 32
    starts = list(range(0, m, step))
    if len(starts) > 1 and m - starts[-1] < 32:
        starts.pop()
    return [(a, starts[i + 1] if i + 1 < len(starts) else m) for i, a in enumerate(starts)]


class _Region:
    """A byte range of the arena exported through ``__cuda_array_interface__``: a tensor made from it (``torch.as_tensor``) has its OWN
    storage object whose deleter drops this Python object -- so `weakref(region)` is dead exactly when no tensor, and no view of one,
    still uses the range (every view shares that storage).  That is what lets the ring allocator reuse memory safely."""

    def __init__(self, ptr: int, nbytes: int, owner):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2, "strides": None}
        self._owner = owner  # keeps the arena allocation alive as long as any output tensor lives


class PeerUnavailable(RuntimeError):
    """The copy-free gather cannot run on this set of ranks (another node, or a device pair without peer access): use the RCCL gather."""


class PeerArena:
    """The per-rank half of the copy-free gather: an IPC-shared arena (a ring of output matrices) and a separate block of control words,
    both mapped by every rank of `group` (one node), and the rendezvous protocol of ``sdnq_hip_push_post`` / ``sdnq_hip_push_columns``
    (include/sdnq_hip.h).

    The control words -- post[16] u64 at bytes [0, 128), done[16] u64 at [128, 256) of a block of their own -- are written by REMOTE
    kernels (P2P stores over xGMI) while a local kernel spins on them, so they live in SIGNAL memory: fine-grained / uncached device
    memory from ``sdnq_hip_signal_alloc`` (`self.ctrl_kind`: "uncached" | "finegrained"), not in the coarse-grained arena, whose
    coherence is only guaranteed at kernel boundaries (round-4 verdict: two processes on ONE GPU share a coherence point and cannot
    see the difference).  The status word (a rendezvous that timed out) is host-coherent memory that `poll()` reads WITHOUT
    synchronizing: every gather looks at it first, so a rank out of step raises at the next layer instead of handing out partially
    filled matrices for the rest of the step (advisor, round 4).

    `gather(y, n_total, col0)` returns this rank's complete row-major [rows, n_total] matrix: a tensor over arena memory (no copy) that
    stays valid as long as it (or any view of it) is referenced -- the ring never hands out a range that a live tensor still uses; when
    the ring is full of live tensors it raises (size it with SDNQ_HIP_TP_ARENA_MB, default 2048).  All ranks must call gather() the same
    number of times in the same order (SPMD), like any collective.  The constructor raises `PeerUnavailable` -- on EVERY rank -- when
    some pair of devices has no peer access or the ranks are not on one host; `PeerArena.try_create` returns None then (the callers
    fall back to the RCCL gather)."""

    CTRL = 256
    CTRL_BYTES = 4096

    @classmethod
    def try_create(cls, *args, **kwargs):
        try:
            return cls(*args, **kwargs)
        except PeerUnavailable:
            return None

    def __init__(self, rank: int, world: int, group=None, device=None, arena_bytes: int | None = None, timeout_ms: int | None = None):
        import ctypes
        import os
        import socket
        import torch.distributed as dist
        from torch.multiprocessing.reductions import reduce_tensor
        from . import _lib
        if world > 16:
            raise ValueError("PeerArena: at most 16 ranks (one node)")
        if timeout_ms is None:  # generous: a peer may sit in a first-call JIT or a host hiccup; a DEAD peer still ends the wait
            timeout_ms = int(os.environ.get("SDNQ_HIP_TP_TIMEOUT_MS", "30000"))
        self.rank, self.world, self.group, self.timeout_ms = rank, world, group, int(timeout_ms)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        lib = self._lib = _lib.load()
        self._imported, self._ctrl, self._status_ptr = [], None, None
        # ---- can every rank reach every other rank's memory?  (decided together: all ranks raise, or none)
        me = (rank, socket.gethostname(), int(self.device.index), os.getpid())
        everyone = [None] * world
        dist.all_gather_object(everyone, me, group=group)
        ok = len({h for (_, h, _, _) in everyone}) == 1
        if ok:
            for (_, _, d, _) in everyone:
                if d != self.device.index and not torch.cuda.can_device_access_peer(self.device.index, d):
                    ok = False
        verdicts = [None] * world
        dist.all_gather_object(verdicts, bool(ok), group=group)
        if not all(verdicts):
            raise PeerUnavailable("PeerArena: the ranks are not on one host with peer access between every pair of devices")
        if arena_bytes is None:
            arena_bytes = int(os.environ.get("SDNQ_HIP_TP_ARENA_MB", "2048")) << 20
        self.size = (int(arena_bytes) + 255) // 256 * 256
        self.buf = torch.zeros(self.CTRL + self.size, dtype=torch.uint8, device=self.device)  # (the first CTRL bytes stay unused: offsets as before)
        if self.buf.data_ptr() % 256:
            raise _lib.SdnqHipError("PeerArena: allocation is not 256-byte aligned")
        # ---- control words in signal memory, the status word in host-coherent memory.  Every step that can fail on ONE rank (allocation, IPC
        # export, IPC import) is followed by an agreement step: a rank that raised alone would leave the others blocked in the next collective
        # until the process-group timeout (advisor, round 5)
        pp, granted = ctypes.c_void_p(), ctypes.c_int()
        handle = ctypes.create_string_buffer(64)
        err = None
        try:
            with torch.cuda.device(self.device):
                _lib.check(lib.sdnq_hip_signal_alloc(self.CTRL_BYTES, 0, ctypes.byref(pp), ctypes.byref(granted)), "signal_alloc")
                self._ctrl = int(pp.value)
                self.ctrl_kind = {1: "uncached", 2: "finegrained"}[granted.value]
                _lib.check(lib.sdnq_hip_signal_alloc(64, 1, ctypes.byref(pp), ctypes.byref(granted)), "signal_alloc(host)")
                self._status_ptr = int(pp.value)
                _lib.check(lib.sdnq_hip_ipc_export(self._ctrl, handle), "ipc_export")
            torch.cuda.synchronize(self.device)  # the zeroed words are in memory before any peer maps them
        except (_lib.SdnqHipError, RuntimeError, KeyError) as e:
            err = f"rank {rank}: {e!r}"
        # exchange the IPC handles (plain data: picklable through any backend's all_gather_object)
        fn, args = reduce_tensor(self.buf)
        gathered = [None] * world
        dist.all_gather_object(gathered, (rank, args, bytes(handle.raw), os.getpid(), err), group=group)
        failed = [g[4] for g in gathered if g[4]]
        if failed:
            raise PeerUnavailable("PeerArena: control memory could not be set up on every rank: " + "; ".join(failed))
        self.peers = [None] * world
        ctrl_ptrs = [0] * world
        err = None
        try:
            for r, a, h, pid, _ in gathered:
                if r == rank:
                    self.peers[r], ctrl_ptrs[r] = self.buf, self._ctrl
                    continue
                self.peers[r] = fn(*a)
                with torch.cuda.device(self.device):
                    _lib.check(lib.sdnq_hip_ipc_import(ctypes.create_string_buffer(h, 64), ctypes.byref(pp)), "ipc_import")
                ctrl_ptrs[r] = int(pp.value)
                self._imported.append(ctrl_ptrs[r])
        except (_lib.SdnqHipError, RuntimeError) as e:
            err = f"rank {rank}: {e!r}"
        verdicts = [None] * world
        dist.all_gather_object(verdicts, err, group=group)
        failed = [v for v in verdicts if v]
        if failed:
            raise PeerUnavailable("PeerArena: a peer's memory could not be mapped on every rank: " + "; ".join(failed))
        ptrs = [int(t.data_ptr()) for t in self.peers]
        arr = ctypes.c_void_p * world
        self._arena = arr(*[p for p in ptrs])
        self._post = arr(*[p for p in ctrl_ptrs])
        self._done = arr(*[p + 128 for p in ctrl_ptrs])
        self._ticket = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.seq = 0
        self.head = self.CTRL
        self.live = []  # [(start, end, weakref(_Region))] in ring order
        dist.barrier(group=group)  # every rank has mapped every arena before the first push

    def __del__(self):
        try:
            lib = self._lib
            for p in self._imported:
                lib.sdnq_hip_ipc_close(p)
            if self._ctrl:
                lib.sdnq_hip_signal_free(self._ctrl, 0)
            if self._status_ptr:
                lib.sdnq_hip_signal_free(self._status_ptr, 1)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    def poll(self):
        """Raise if a rendezvous of this rank has timed out -- read from host-coherent memory, no synchronization: cheap enough to run
        in front of every gather."""
        import ctypes
        if self._status_ptr and ctypes.c_int.from_address(self._status_ptr).value != 0:
            raise RuntimeError("PeerArena: a peer did not arrive within the timeout (ranks out of step, or a rank died); every gather since "
                               "then is invalid -- rebuild the arena (or fall back to the RCCL gather)")

    # ---- ring allocator over [CTRL, CTRL + size) ----------------------------------------------------------------------------------
    def _alloc(self, nbytes: int):
        """Next free range of the ring: dead ranges in the way are recycled, LIVE ones (a tensor or a view of one still references
        them) are stepped over; raises only when one full turn of the ring finds no room."""
        import weakref
        n = (nbytes + 255) // 256 * 256
        if n > self.size:
            raise RuntimeError(f"PeerArena: output of {nbytes} bytes exceeds the arena ({self.size} bytes; SDNQ_HIP_TP_ARENA_MB)")
        self.live = [e for e in self.live if e[2]() is not None]  # only live ranges matter
        lo, hi = self.CTRL, self.CTRL + self.size
        start, wraps = self.head, 0
        while True:
            if start + n > hi:
                start, wraps = lo, wraps + 1
                if wraps > 1:
                    raise RuntimeError("PeerArena: the ring is full of live output tensors; raise SDNQ_HIP_TP_ARENA_MB or drop references "
                                       f"({sum(b - a for a, b, _ in self.live)} of {self.size} bytes are in use, {n} needed)")
            clash = max((b for (a, b, _) in self.live if a < start + n and start < b), default=None)
            if clash is None:
                break
            start = clash  # step over the live range
        region = _Region(self.buf.data_ptr() + start, n, self.buf)
        self.live.append((start, start + n, weakref.ref(region)))
        self.head = start + n
        return start, region

    def gather(self, y: torch.Tensor, n_total: int, col0: int) -> torch.Tensor:
        """This rank's slab y [rows, w] goes to columns [col0, col0 + w) of every rank's [rows, n_total] output; returns this rank's."""
        from . import _lib, ops
        lib = _lib.load()
        self.poll()
        rows, w = y.shape
        es = y.element_size()
        if y.stride(1) != 1:
            y = y.contiguous()
        off, region = self._alloc(rows * n_total * es)
        self.seq = (self.seq + 1) & 0xFFFFFF
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ops.check(lib.sdnq_hip_push_post(self._post, self.world, self.rank, self.seq, off, stream), "push_post")
        ops.check(lib.sdnq_hip_push_columns(y.data_ptr(), es, rows, w, y.stride(0), self._arena, self._post, self._done, self.world, self.rank,
                                            self.seq, n_total, col0, 0, self._ticket.data_ptr(), self._status_ptr, self.timeout_ms,
                                            stream), "push_columns")
        out = torch.as_tensor(region, device=self.device).view(y.dtype)[: rows * n_total].view(rows, n_total)
        return out

    def post(self, rows: int, n_total: int, dtype: torch.dtype):
        """Two-step form: announce the destination BEFORE the layer's matmul (the peers' pushes then never wait for it); returns a
        token for `push`."""
        from . import _lib, ops
        self.poll()
        es = torch.empty((), dtype=dtype).element_size()
        off, region = self._alloc(rows * n_total * es)
        self.seq = (self.seq + 1) & 0xFFFFFF
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ops.check(_lib.load().sdnq_hip_push_post(self._post, self.world, self.rank, self.seq, off, stream), "push_post")
        return (self.seq, off, region, rows, n_total, dtype)

    def push(self, token, y: torch.Tensor, col0: int) -> torch.Tensor:
        from . import _lib, ops
        seq, off, region, rows, n_total, dtype = token
        assert y.shape[0] == rows and y.dtype == dtype
        if y.stride(1) != 1:
            y = y.contiguous()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        ops.check(_lib.load().sdnq_hip_push_columns(y.data_ptr(), y.element_size(), rows, y.shape[1], y.stride(0), self._arena, self._post,
                                                    self._done, self.world, self.rank, seq, n_total, col0, 0, self._ticket.data_ptr(),
                                                    self._status_ptr, self.timeout_ms, stream), "push_columns")
        return torch.as_tensor(region, device=self.device).view(dtype)[: rows * n_total].view(rows, n_total)

    def check(self):
        """Synchronizes; raises if any rendezvous of this rank timed out (a peer that never arrived)."""
        torch.cuda.synchronize(self.device)
        self.poll()


class ColumnShardedLinear(torch.nn.Module):
    """Holds this rank's slab of a Linear; forward = local forward + all-gather along the channel axis.

    On GPU tensors the re-assembly [W, M, w] -> [M, N] is ONE HIP pass (``sdnq_hip_unshard_columns``; uneven shards included: the
    local slab is written into a padded gather buffer, no zeros / cat), and with ``chunks > 1`` the M rows are processed as a
    pipeline: the all-gather of chunk i runs on a side stream while the local matmul of chunk i + 1 runs on the caller's stream
    (row-wise activation quantization, the matmul and its epilogue are all per activation row, so a chunked int8 layer is
    bit-identical to the unchunked one -- `chunk_rows` never leaves a chunk below the 32-row small-batch threshold; fp8 / float
    matmuls may pick another tile for a chunk's M, i.e. another fp32 summation order: equal within the usual float tolerance).  The gather itself stays the bound -- 2 M N / W bytes per rank over one 153 GB/s xGMI link -- the
    pipeline only hides the matmul under it.  CPU tensors (the gloo tests) take the torch path."""

    def __init__(self, local: torch.nn.Module, n_total: int, rank: int, world: int, group=None, chunks: int = 1, peer: "PeerArena | None" = None):
        super().__init__()
        self.local = local
        self.n_total, self.rank, self.world, self.group = n_total, rank, world, group
        self.peer = peer  # copy-free gather over peer-mapped memory instead of RCCL all-gather + re-assembly
        self.bounds = [shard_bounds(n_total, r, world) for r in range(world)]
        self.even = len({b - a for a, b in self.bounds}) == 1
        self.chunks = max(1, int(chunks))
        self._side = None  # side stream of the pipelined gather (per device)

    def _forward_torch(self, y2: torch.Tensor, m: int) -> torch.Tensor:
        import torch.distributed as dist
        if self.even:
            gathered = torch.empty((self.world * m, y2.shape[1]), device=y2.device, dtype=y2.dtype)  # rank-major concat
            dist.all_gather_into_tensor(gathered, y2, group=self.group)
            return gathered.view(self.world, m, y2.shape[1]).permute(1, 0, 2).reshape(m, self.n_total)
        # uneven shards: pad every slab to the widest one (collectives want equal messages), gather, trim
        wmax = max(b - a for a, b in self.bounds)
        padded = torch.zeros((m, wmax), device=y2.device, dtype=y2.dtype)
        padded[:, : y2.shape[1]] = y2
        gathered = torch.empty((self.world * m, wmax), device=y2.device, dtype=y2.dtype)
        dist.all_gather_into_tensor(gathered, padded, group=self.group)
        g3 = gathered.view(self.world, m, wmax)
        return torch.cat([g3[r, :, : b - a] for r, (a, b) in enumerate(self.bounds)], dim=-1)

    def _gather_chunk(self, y2: torch.Tensor, out: torch.Tensor, m0: int):
        """All-gather the slab rows y2 [rows, w_r] and scatter them into out[m0 : m0 + rows] (current stream)."""
        import torch.distributed as dist
        from . import ops
        rows = y2.shape[0]
        wmax = max(b - a for a, b in self.bounds)
        if y2.shape[1] == wmax and y2.is_contiguous():
            send = y2
        else:  # the narrower slabs of an uneven split travel padded (the pad columns are never read back)
            send = torch.empty((rows, wmax), device=y2.device, dtype=y2.dtype)
            send[:, : y2.shape[1]] = y2
        gathered = torch.empty((self.world, rows, wmax), device=y2.device, dtype=y2.dtype)
        dist.all_gather_into_tensor(gathered.view(self.world * rows, wmax), send, group=self.group)
        ops.unshard_columns(gathered, out, [a for a, _ in self.bounds] + [self.n_total], m0)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        k = x.shape[-1]
        lead = x.shape[:-1]
        x2 = x.reshape(-1, k)
        m = x2.shape[0]
        es = x.element_size()
        if not x.is_cuda or m == 0 or any((a * es) % 16 for a, _ in self.bounds):
            y_local = self.local(x)
            return self._forward_torch(y_local.reshape(-1, y_local.shape[-1]).contiguous(), m).view(*lead, self.n_total)
        if self.peer is not None:
            # destination announced first, so that no peer's push ever waits for it; then the local matmul; then the push into every
            # rank's matrix (the kernel returns when this rank's matrix is complete)
            token = self.peer.post(m, self.n_total, x.dtype)
            y = self.local(x2)
            return self.peer.push(token, y, self.bounds[self.rank][0]).view(*lead, self.n_total)
        out = torch.empty((m, self.n_total), device=x.device, dtype=x.dtype)
        chunks = min(self.chunks, max(1, m // 256))
        if chunks == 1:
            y = self.local(x2)
            self._gather_chunk(y, out, 0)
            return out.view(*lead, self.n_total)
        # pipeline over M: matmul of chunk i + 1 (caller's stream) overlaps gather + scatter of chunk i (side stream)
        cur = torch.cuda.current_stream(x.device)
        if self._side is None or self._side.device != x.device:
            self._side = torch.cuda.Stream(device=x.device)
        side = self._side
        side.wait_stream(cur)  # `out` was allocated on the caller's stream
        for m0, m1 in chunk_rows(m, chunks):
            y = self.local(x2[m0:m1])
            ev = torch.cuda.Event()
            ev.record(cur)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                self._gather_chunk(y, out, m0)
            y.record_stream(side)
        cur.wait_stream(side)
        return out.view(*lead, self.n_total)


def _slab_of_rows(t: torch.Tensor, n: int, a: int, b: int, what: str) -> torch.Tensor:
    """Rows [a, b) of a per-output-channel tensor whose FIRST dimension covers the N channels in order, possibly several
    storage rows per channel (packed codecs: [N * K / G, words] or 1-D [N * K * bits / 8])."""
    rows = t.shape[0]
    if rows % n:
        raise ValueError(f"{what}: first dimension {rows} is not a multiple of N={n}")
    per = rows // n
    return t[a * per:b * per]


@torch.no_grad()
def shard_quantized_module(mod: torch.nn.Module, a: int, b: int) -> torch.nn.Module:
    """SDNQLinear holding output channels [a, b) of the quantized layer `mod`, its tensors being views of mod's parameters."""
    from .dequantizer import SDNQDequantizer
    from .forward import get_forward_func
    from .layers import SDNQLinear
    from .loader import _DQ_FIELDS, adopt_dequantizer
    dq0 = adopt_dequantizer(mod.sdnq_dequantizer)
    if dq0.layer_class_name not in ("Linear", "SDNQLinear") or dq0.use_codebook:
        raise NotImplementedError("column sharding is built for quantized Linear layers")
    n, k = dq0.out_features, dq0.in_features
    if not (0 <= a < b <= n) or a % 16 or (b - a) % 16:
        raise ValueError(f"channel range [{a}, {b}) of N={n} must be 16-aligned")
    w = mod.weight
    transposed = dq0.weight_is_transposed
    if transposed:  # logical [K, N], strides (1, K) (or contiguous [K, N] straight from safetensors): columns = channels
        if tuple(w.shape) != (k, n):
            raise ValueError(f"transposed weight must be [K, N] = ({k}, {n}), got {tuple(w.shape)}")
        w_s = w[:, a:b]
    else:
        w_s = _slab_of_rows(w, n, a, b, "weight")

    def per_channel(t, what):
        if t is None:
            return None
        if t.shape[0] == n:
            return t[a:b]
        if t.shape[-1] == n:  # [1, N] of the transposed layout
            return t[..., a:b]
        return _slab_of_rows(t, n, a, b, what)

    scale = per_channel(mod.scale, "scale")
    zp = per_channel(getattr(mod, "zero_point", None), "zero_point")
    svd_up, svd_down = getattr(mod, "svd_up", None), getattr(mod, "svd_down", None)
    if svd_up is not None:  # [R, N] / [K, R] with use_quantized_matmul (quantizer.py:164-167), else [N, R] / [R, K]
        svd_up = svd_up[:, a:b] if dq0.use_quantized_matmul else svd_up[a:b]
    bias = None if mod.bias is None else mod.bias[a:b]
    fields = {f: getattr(dq0, f) for f in _DQ_FIELDS}
    dq = SDNQDequantizer(**fields)
    dq.original_shape = torch.Size((b - a, *dq0.original_shape[1:]))
    dq.original_stride = list(torch.empty(dq.original_shape, device="meta").stride())
    if dq0.result_shape is not None:
        dq.result_shape = torch.Size((b - a, *dq0.result_shape[1:]))
    qs = list(dq0.quantized_weight_shape)
    qs[1 if transposed else 0] = b - a
    dq.quantized_weight_shape = torch.Size(qs)
    skeleton = torch.nn.Linear(8, 8, bias=False)
    skeleton.in_features, skeleton.out_features = k, b - a
    skeleton.sdnq_dequantizer = dq
    slab = SDNQLinear(skeleton, get_forward_func("Linear", dq.quantized_matmul_dtype, dq.use_quantized_matmul))
    P = lambda t: None if t is None else torch.nn.Parameter(t, requires_grad=False)  # noqa: E731
    slab.weight, slab.scale, slab.zero_point = P(w_s), P(scale), P(zp)
    slab.svd_up, slab.svd_down, slab.bias = P(svd_up), P(svd_down), P(bias)
    return slab


@torch.no_grad()
def column_shard_module(mod: torch.nn.Module, rank: int, world: int, group=None, chunks: int = 1, peer: "PeerArena | None" = None) -> ColumnShardedLinear:
    """Tensor-parallel shard of a PRE-QUANTIZED SDNQLinear (checkpoint layout untouched): this rank's slab (views of mod's
    parameters) + the RCCL all-gather of the outputs.  Bit-identical to `mod` for every storage format.  chunks > 1 pipelines the
    gather of one M chunk under the matmul of the next (ColumnShardedLinear)."""
    n = int(mod.sdnq_dequantizer.original_shape[0])
    a, b = shard_bounds(n, rank, world)
    return ColumnShardedLinear(shard_quantized_module(mod, a, b), n, rank, world, group, chunks=chunks, peer=peer)


@torch.no_grad()
def column_shard_linear(linear: torch.nn.Linear, config: SDNQConfig, rank: int, world: int, group=None,
                        quantize=sdnq_quantize_layer) -> ColumnShardedLinear:
    """Slice a float nn.Linear on its output channels, quantize the slab, wrap it. `quantize` is injectable so the
    CPU (gloo) tests can exercise the sharding + collective plumbing without a GPU."""
    n = linear.out_features
    a, b = shard_bounds(n, rank, world)
    slab = torch.nn.Linear(linear.in_features, b - a, bias=linear.bias is not None, device=linear.weight.device,
                           dtype=linear.weight.dtype)
    slab.weight.copy_(linear.weight[a:b])
    if linear.bias is not None:
        slab.bias.copy_(linear.bias[a:b])
    local, _ = quantize(slab, config)
    return ColumnShardedLinear(local, n, rank, world, group)
