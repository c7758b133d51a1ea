"""Block sharding across the GPUs of one node: one process per GPU, torch.distributed.

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
            ops = [self.dist.P2POp(self.dist.irecv, self.rx[r][w % self.slots], self._global(r), self.group) for r in self.peers]
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
            if self.is_cuda:                                # the transfer waits for the compute of this wave only;
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
