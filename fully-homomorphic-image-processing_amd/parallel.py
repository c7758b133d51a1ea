"""Sharding of the three circuits across the GPUs of one node: one process per GPU, torch.distributed.

| circuit | reference loop | unit | shard of rank r of R | replicated per rank | exchange |
|---|---|---|---|---|---|
| DCT + quant | homo/server_jpeg.cpp:113-138 (blocks) | 8x8 block | contiguous block range (block_range) | twiddles, 76 constants | none (optional output gather) |
| resize | homo/fhe_resize.h:350-388 (rows y, pixels x) | destination row | contiguous row range (row_range) + source rows +- halo (source_rows) | xfract per column, constants | none (rows land in one file / optional gather) |
| decode | homo/server_decode.cpp:120-137, homo/fhe_decode.h:224 | (channel, position) | contiguous unit range (decode_shards) | runs, the `index` chain (pairs additions), the offset chain, the sine polynomials | one broadcast of the three Enc(0) `index` ciphertexts |

The reference is a single-threaded loop over blocks (homo/server_jpeg.cpp:113); blocks are
independent, so rank r of R owns the contiguous range [r*N/R, (r+1)*N/R) and no collective is
needed on the data path.  The only exchanges are a barrier, an all-reduce of 64-bit output digests
(cheap verification that every shard was produced) and -- when the caller asks for the ciphertexts
on one rank -- a per-wave point-to-point gather of the output shards (WaveGather: RCCL send/recv over xGMI on
GPUs, gloo on CPU in the tests).

Everything here is backend-agnostic: `compute(blocks)` is any callable mapping a shard of input
blocks to output blocks (the HIP evaluator on GPUs).
"""
import torch


def block_range(rank, world, n_blocks):
    """Contiguous shard of global block indices owned by `rank` (sizes differ by at most one)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, rem = divmod(n_blocks, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def row_range(rank, world, dst_h):
    """Destination rows [y0, y1) of ResizeImage's outer loop (homo/fhe_resize.h:350) owned by `rank`."""
    return block_range(rank, world, dst_h)


def source_rows(src_h, dst_h, row0, row1, bicubic=True):
    """(first, count) of the source rows destination rows [row0, row1) read -- the shard's rows plus the sampler's halo
    (yi - 1 .. yi + 2 for SampleBicubic, yi .. yi + 1 for SampleLinear, clamped to the image: homo/fhe_resize.h:264-290,
    229-240,215-220) -- with v in float32 exactly as the reference computes it (:351).  The same figure as the C ABI's
    fhe_resize_source_rows (tests/test_cabi.py compares them)."""
    import numpy as np
    if not (0 <= row0 < row1 <= dst_h) or dst_h < 2 or src_h < 1:
        raise ValueError("bad row range [%d, %d) of %d" % (row0, row1, dst_h))
    f32 = np.float32
    lo, hi = src_h, -1
    for y in range(row0, row1):
        v = f32(f32(y) / f32(dst_h - 1) * f32(src_h)) - f32(0.5)
        yi = int(v)
        a, z = (yi - 1, yi + 2) if bicubic else (yi, yi + 1)
        lo = min(lo, min(max(a, 0), src_h - 1))
        hi = max(hi, min(max(z, 0), src_h - 1))
    return lo, hi - lo + 1


def decode_shards(rank, world, npos, channels=3):
    """The (channel, pos0, pos1) pieces of `rank`'s contiguous range of the channels * npos (channel, position) units of
    homo/server_decode.cpp:120-137 (channel-major).  A rank whose range crosses a channel boundary gets two pieces."""
    u0, u1 = block_range(rank, world, channels * npos)
    out = []
    for ch in range(channels):
        a, z = max(u0, ch * npos), min(u1, (ch + 1) * npos)
        if a < z:
            out.append((ch, a - ch * npos, z - ch * npos))
    return out


def broadcast_from_root(t, src=0, group=None):
    """One small broadcast (the decode path's three Enc(0) `index` ciphertexts, drawn once on the root so that every
    shard of a channel continues the same chain); a no-op without a process group."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        gsrc = src if group is None else dist.get_global_rank(group, src)
        dist.broadcast(t, src=gsrc, group=group)
    return t


def rank_world(group=None):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def run_resize_sharded(load_rows, sample_rows, src_h, dst_h, digest, bicubic=True, gather=False, group=None):
    """Shard ResizeImage's destination rows over the process group.  pixels = load_rows(first, count) brings in the
    source rows [first, first + count) (the shard's rows +- halo; every rank reads its own, inputs are read-only);
    out = sample_rows(pixels, first, count, y0, y1) -> tensor [y1 - y0, dst_w, ...]; digest(out, y0) is the
    order-independent digest with global indices.  Returns ((y0, y1), local_out, global_digest, gathered_or_None).
    A rank without rows (world > dst_h) contributes an empty band."""
    rank, world = rank_world(group)
    y0, y1 = row_range(rank, world, dst_h)
    if y1 > y0:
        first, count = source_rows(src_h, dst_h, y0, y1, bicubic)
        local_out = sample_rows(load_rows(first, count), first, count, y0, y1)
        local_digest = digest(local_out, y0)
    else:
        local_out, local_digest = sample_rows(None, 0, 0, y0, y0), 0
    total = combine_digests(local_digest, group)
    method = gather if isinstance(gather, str) else "collective"
    gathered = gather_outputs(local_out, dst_h, 0, group, method=method) if (gather and world > 1) else (local_out if gather else None)
    return (y0, y1), local_out, total, gathered


def run_decode_sharded(decode_piece, npos, digest, channels=3, group=None):
    """Shard the (channel, position) units of the decode driver loop.  decode_piece(ch, pos0, pos1) -> tensor
    [pos1 - pos0, S, k, n] (it replays the channel's `index` chain itself); digest(out, ch, pos0).  Returns
    (pieces [(ch, pos0, pos1, out)], global_digest)."""
    rank, world = rank_world(group)
    pieces, local = [], 0
    for ch, p0, p1 in decode_shards(rank, world, npos, channels):
        out = decode_piece(ch, p0, p1)
        local = (local + digest(out, ch, p0)) & 0xFFFFFFFFFFFFFFFF
        pieces.append((ch, p0, p1, out))
    return pieces, combine_digests(local, group)


def words_per_block(k, n, size=2):
    return 64 * size * k * n


def shard_first_index(rank, world, n_blocks, k, n):
    """Linear index of the first u64 of this rank's shard in the global synthetic input stream, so
    that any GPU count generates byte-identical blocks."""
    start, _ = block_range(rank, world, n_blocks)
    return start * words_per_block(k, n)


def combine_digests(local_digest, group=None):
    """Sum (mod 2^64) of the per-rank order-independent digests == digest of the whole output."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local_digest & 0xFFFFFFFFFFFFFFFF
    # two 32-bit halves in int64 so that the SUM all-reduce cannot overflow
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    parts = torch.tensor([local_digest & 0xFFFFFFFF, (local_digest >> 32) & 0xFFFFFFFF], dtype=torch.int64, device=dev)
    dist.all_reduce(parts, op=dist.ReduceOp.SUM, group=group)
    lo, hi = int(parts[0].item()), int(parts[1].item())
    return (lo + (hi << 32)) & 0xFFFFFFFFFFFFFFFF


class WaveGather:
    """The final ciphertext gather (BASELINE.json north_star: "RCCL over xGMI only for the final ciphertext
    gather"), done per WAVE of blocks with point-to-point transfers instead of one collective at the end.

    Why per wave and point to point: one step leaves 12 MiB of output per block (n = 4096, k = 3); config[4]'s
    shard is 96 GiB per GPU and cannot be gathered in one piece, and xGMI is point-to-point (each peer reaches
    the root over its own link, ~153 GB/s; the root takes in at most 7 links at once), so a ring collective would
    be bound by a single link.  Each non-root rank sends wave w to the root (ncclSend on RCCL / gloo send in the
    CPU tests) from a ring of `slots` output buffers while it computes wave w + 1 into the next slot; the root
    posts the matching receives (grouped, one per peer) into its own ring and hands every received wave -- and
    its own -- to `consume(src_rank, wave_index, tensor)`, e.g. a digest or a stream writer.  Sends and receives
    of a pair are issued in wave order on both sides, which is what NCCL/RCCL p2p matching requires, and BOTH
    sides go through `batch_isend_irecv`: since torch 2.2 batched point-to-point operations run on the group's
    collective communicator while a plain `isend` / `irecv` lazily creates a two-rank communicator, so mixing
    the two forms leaves each side waiting in a communicator the other never joins.  Receives are posted when a
    pass starts (first `acquire`), never ahead of it, so no unmatched receive sits in front of a later barrier
    or all-reduce.

    Usage on every rank:
        g = WaveGather(wave_shape, dtype, device, n_waves, consume=...)
        for w in range(n_waves):
            buf = g.acquire()            # an output buffer that is free again (its send has completed)
            compute wave w into buf
            g.commit(w)                  # non-root: isend;  root: consume own wave, receive + consume the peers'
        g.finish()
    """

    def __init__(self, wave_shape, dtype, device, n_waves, dst=0, group=None, slots=2, consume=None):
        import torch.distributed as dist
        self.dist, self.group, self.dst = dist, group, dst
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.n_waves, self.slots = n_waves, max(2, slots)
        self.consume = consume or (lambda src, w, t: None)
        self.ring = [torch.empty(wave_shape, dtype=dtype, device=device) for _ in range(self.slots)]
        self.sends = [None] * self.slots                   # in-flight send per slot (non-root)
        self.peers = [r for r in range(self.world) if r != dst]
        self.is_cuda = torch.device(device).type == "cuda"
        self.side = torch.cuda.Stream(device=device) if self.is_cuda else None
        # root: receive ring, `slots` waves deep per peer; receives for wave w are posted as soon as slot w % slots is free
        self.rx = {r: [torch.empty(wave_shape, dtype=dtype, device=device) for _ in range(self.slots)] for r in self.peers} if self.rank == dst else {}
        # TEST MODE only (gloo with device tensors: the world > 1 path on a box with fewer devices than ranks): gloo moves host
        # memory, so a wave goes device -> page-locked host -> gloo -> page-locked host -> device.  RCCL takes device pointers.
        self.stage = self.is_cuda and dist.get_backend(group) != "nccl"
        if self.stage:
            self.tx_host = [torch.empty(wave_shape, dtype=dtype).pin_memory() for _ in range(self.slots)] if self.rank != dst else []
            self.rx_host = {r: [torch.empty(wave_shape, dtype=dtype).pin_memory() for _ in range(self.slots)] for r in self.peers} if self.rank == dst else {}
        self.rx_work = {}                                   # wave -> list of (src, work)
        self.next, self.posted, self.consumed = 0, 0, 0

    def reset(self):
        """start another pass of n_waves waves with the same buffers (call after finish()); the pass's receives are
        posted by its first acquire(), so a reset that is not followed by a pass leaves nothing in flight"""
        assert not self.rx_work and all(w is None for w in self.sends)
        self.next, self.posted, self.consumed = 0, 0, 0

    def _global(self, r):
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def _post_receives(self):
        """keep up to `slots` waves of receives in flight (each slot is reposted once its wave was consumed)"""
        while self.posted < self.n_waves and self.posted - self.consumed < self.slots and self.peers:
            w = self.posted
            into = self.rx_host if self.stage else self.rx
            ops = [self.dist.P2POp(self.dist.irecv, into[r][w % self.slots], self._global(r), self.group) for r in self.peers]
            works = self.dist.batch_isend_irecv(ops)
            self.rx_work[w] = list(zip(self.peers, works if len(works) == len(ops) else [works[0]] * len(ops)))
            self.posted += 1

    def _send(self, buf):
        """one batched send to the root (same primitive as the root's batched receives, see the class comment)"""
        works = self.dist.batch_isend_irecv([self.dist.P2POp(self.dist.isend, buf, self._global(self.dst), self.group)])
        return works[0]

    def acquire(self):
        if self.rank == self.dst and self.next == 0 and self.posted == 0:
            self._post_receives()                           # the pass starts here
        slot = self.next % self.slots
        if self.sends[slot] is not None:                    # the buffer is free once its send has completed
            self.sends[slot].wait()
            self.sends[slot] = None
        return self.ring[slot]

    def commit(self, wave_index):
        slot = self.next % self.slots
        assert wave_index == self.next
        buf = self.ring[slot]
        if self.rank != self.dst:
            if self.stage:                                  # test mode: the wave leaves through host memory
                self.tx_host[slot].copy_(buf, non_blocking=True)
                torch.cuda.current_stream().synchronize()
                self.sends[slot] = self._send(self.tx_host[slot])
            elif self.is_cuda:                              # the transfer waits for the compute of this wave only;
                self.side.wait_stream(torch.cuda.current_stream())   # the caller's stream goes on with the next wave
                with torch.cuda.stream(self.side):
                    self.sends[slot] = self._send(buf)
            else:
                self.sends[slot] = self._send(buf)
        else:
            self.consume(self.dst, wave_index, buf)
            self._drain(wave_index)
        self.next += 1

    def _drain(self, upto):
        while self.consumed <= upto and self.consumed < self.n_waves and self.peers:
            w = self.consumed
            for src, work in self.rx_work.pop(w):
                work.wait()
                if self.stage:
                    self.rx[src][w % self.slots].copy_(self.rx_host[src][w % self.slots], non_blocking=True)
                    torch.cuda.current_stream().synchronize()        # the host slot is handed to the next receive right after
                self.consume(src, w, self.rx[src][w % self.slots])
            self.consumed += 1
            self._post_receives()

    def finish(self):
        if self.rank == self.dst:
            if self.next == 0 and self.posted == 0 and self.n_waves:
                self._post_receives()                       # a root that computed nothing itself still receives
            self._drain(self.n_waves - 1)
        for i, wk in enumerate(self.sends):
            if wk is not None:
                wk.wait()
                self.sends[i] = None
        if self.is_cuda:
            torch.cuda.current_stream().wait_stream(self.side)


def gather_outputs(local_out, n_blocks, dst=0, group=None, wave_blocks=64, method="collective"):
    """The output shards on rank `dst` in global block order (None elsewhere).  Meant for results that fit one device;
    larger jobs pass their own `consume` to WaveGather (digest, stream writer) instead of materialising everything.

    method="collective" (default): one `dist.gather` of shards padded to the longest one -- the plain RCCL collective,
    kept as the default until a multi-GPU RCCL run of tests/test_gpu_multi.py has exercised the wave path.
    method="wave": waves of `wave_blocks` blocks through WaveGather (point-to-point, overlappable with compute)."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = [block_range(r, world, n_blocks) for r in range(world)]
    longest = max(e - s for s, e in per)
    result = torch.empty((n_blocks,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=local_out.device) if rank == dst else None
    if method == "collective":
        padded = local_out
        if local_out.shape[0] < longest:
            padded = torch.zeros((longest,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=local_out.device)
            padded[:local_out.shape[0]].copy_(local_out)
        parts = [torch.empty_like(padded) for _ in range(world)] if rank == dst else None
        gdst = dst if group is None else dist.get_global_rank(group, dst)
        dist.gather(padded.contiguous(), parts, dst=gdst, group=group)
        if rank == dst:
            for r, (s, e) in enumerate(per):
                result[s:e].copy_(parts[r][:e - s])
        return result
    if method != "wave":
        raise ValueError("gather method must be 'collective' or 'wave'")
    wave_blocks = max(1, min(wave_blocks, max(1, longest)))
    n_waves = (longest + wave_blocks - 1) // wave_blocks
    shape = (wave_blocks,) + tuple(local_out.shape[1:])

    def consume(src, w, t):
        s, e = per[src]
        lo = s + w * wave_blocks
        cnt = max(0, min(wave_blocks, e - lo))
        if cnt:
            result[lo:lo + cnt].copy_(t[:cnt])

    g = WaveGather(shape, local_out.dtype, local_out.device, n_waves, dst=dst, group=group, consume=consume)
    for w in range(n_waves):
        buf = g.acquire()
        lo = w * wave_blocks
        cnt = max(0, min(wave_blocks, local_out.shape[0] - lo))
        if cnt:
            buf[:cnt].copy_(local_out[lo:lo + cnt])
        g.commit(w)
    g.finish()
    return result


class LocalDrain:
    """A consumer for sharded outputs that scales with the GPU count: every rank moves ITS OWN waves to pinned host
    memory over its own PCIe link (a side stream, overlapped with the next wave's compute) and hands the host buffer to
    `consume(wave_index, host_tensor)` -- the shape of a per-GPU stream writer.  No inter-GPU traffic at all; the
    per-GPU ceiling is the link (PCIe Gen5 x16, ~55-60 GB/s sustained = ~4.5 k blocks/s of 12 MiB outputs), identical
    on every rank.  Same acquire / commit / finish protocol as WaveGather."""

    def __init__(self, wave_shape, dtype, device, slots=2, consume=None):
        self.ring = [torch.empty(wave_shape, dtype=dtype, device=device) for _ in range(max(2, slots))]
        self.host = [torch.empty(wave_shape, dtype=dtype).pin_memory() for _ in self.ring]
        self.done = [None] * len(self.ring)                 # (event, wave) of the copy in flight per slot
        self.side = torch.cuda.Stream(device=device)
        self.consume = consume or (lambda w, t: None)
        self.next = 0

    def _retire(self, slot):
        if self.done[slot] is not None:
            ev, w = self.done[slot]
            ev.synchronize()
            self.consume(w, self.host[slot])
            self.done[slot] = None

    def acquire(self):
        slot = self.next % len(self.ring)
        self._retire(slot)
        return self.ring[slot]

    def commit(self, wave_index):
        slot = self.next % len(self.ring)
        self.side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.side):
            self.host[slot].copy_(self.ring[slot], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.done[slot] = (ev, wave_index)
        self.next += 1

    def finish(self):
        for i in range(len(self.ring)):
            self._retire((self.next + i) % len(self.ring))
        torch.cuda.current_stream().wait_stream(self.side)

    def reset(self):
        self.next = 0


def run_sharded(compute, make_inputs, n_blocks, digest, gather=False, group=None):
    """Shard `n_blocks` over the process group: inputs = make_inputs(start, end), out = compute(inputs).
    gather: False, True (= "collective") or a gather_outputs method name ("collective", "wave").
    Returns (local_out, global_digest, gathered_or_None)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    start, end = block_range(rank, world, n_blocks)
    local_out = compute(make_inputs(start, end))
    total = combine_digests(digest(local_out, start), group)
    method = gather if isinstance(gather, str) else "collective"
    gathered = gather_outputs(local_out, n_blocks, 0, group, method=method) if (gather and world > 1) else (local_out if gather else None)
    return local_out, total, gathered
