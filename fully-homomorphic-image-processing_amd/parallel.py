"""Block sharding across the GPUs of one node: one process per GPU, torch.distributed.

The reference is a single-threaded loop over blocks (homo/server_jpeg.cpp:113); blocks are
independent, so rank r of R owns the contiguous range [r*N/R, (r+1)*N/R) and no collective is
needed on the data path.  The only exchanges are a barrier, an all-reduce of 64-bit output digests
(cheap verification that every shard was produced) and -- when the caller asks for the ciphertexts
on one rank -- a gather of the output shards (RCCL over xGMI on GPUs, gloo on CPU in the tests).

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
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    parts = torch.tensor([local_digest & 0xFFFFFFFF, (local_digest >> 32) & 0xFFFFFFFF], dtype=torch.int64, device=dev)
    dist.all_reduce(parts, op=dist.ReduceOp.SUM, group=group)
    lo, hi = int(parts[0].item()), int(parts[1].item())
    return (lo + (hi << 32)) & 0xFFFFFFFFFFFFFFFF


def gather_outputs(local_out, n_blocks, dst=0, group=None):
    """Gather the output shards on rank `dst` in global block order.  Shards may differ in size by
    one block, so they are exchanged as padded equal-size tensors."""
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = [block_range(r, world, n_blocks) for r in range(world)]
    longest = max(e - s for s, e in per)
    pad = torch.zeros((longest,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=local_out.device)
    pad[: local_out.shape[0]] = local_out
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, bufs, dst=dst, group=group)
        return torch.cat([b[: e - s] for b, (s, e) in zip(bufs, per)], dim=0)
    dist.gather(pad, None, dst=dst, group=group)
    return None


def run_sharded(compute, make_inputs, n_blocks, digest, gather=False, group=None):
    """Shard `n_blocks` over the process group: inputs = make_inputs(start, end), out = compute(inputs).
    Returns (local_out, global_digest, gathered_or_None)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    start, end = block_range(rank, world, n_blocks)
    local_out = compute(make_inputs(start, end))
    total = combine_digests(digest(local_out, start), group)
    gathered = gather_outputs(local_out, n_blocks, 0, group) if (gather and world > 1) else (local_out if gather else None)
    return local_out, total, gathered
