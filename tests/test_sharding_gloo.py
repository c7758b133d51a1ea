"""CPU, world_size 2 over gloo: the N>1 path (block sharding, digest all-reduce, output gather).

The per-shard compute is the CPU oracle at a test-sized ring (the GPU evaluator is the same
callable shape); the assertions are about the orchestration: shards tile the block range, any
world size generates identical input bytes, the combined digest equals the single-process digest
and the gathered output equals the single-process output."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, Q, T, NB = 64, [0xFFFFEE001, 0xFFFFC4001, 0x1FFFFE0001], 1 << 14, 5
M64 = (1 << 64) - 1


def _sm(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)


def _digest(arr, index0):
    """same definition as fhe_digest: sum of splitmix64(value ^ splitmix64(index)) mod 2^64"""
    tot = 0
    for i, v in enumerate(arr.ravel()):
        tot = (tot + _sm(int(v) ^ _sm(index0 + i))) & M64
    return tot


def _setup():
    sys.path.insert(0, ROOT)
    import fhip_amd as fhe
    from oracle import oracle as om
    return fhe, om, om.Oracle(N, Q, T)


def _pipeline(fhe, om, orc):
    wpb = fhe.parallel.words_per_block(len(Q), N)

    def make_inputs(s, e):
        return orc.random_ct((e - s) * 64, seed=om.SEED, first_index=s * wpb).reshape(e - s, 64, 2, len(Q), N)

    def compute(blocks):
        return torch.from_numpy(np.stack([orc.encrypted_dct(b) for b in blocks]).view(np.int64)) if len(blocks) else torch.zeros((0, 64, 2, len(Q), N), dtype=torch.int64)

    def digest(out, start):
        return _digest(out.numpy().view(np.uint64), start * wpb)

    return make_inputs, compute, digest


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fhe, om, orc = _setup()
    make_inputs, compute, digest = _pipeline(fhe, om, orc)
    local, total, gathered = fhe.parallel.run_sharded(compute, make_inputs, NB, digest, gather=True)
    q.put((rank, tuple(local.shape), total, None if gathered is None else gathered.numpy().view(np.uint64).copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_block_range_tiles_everything():
    sys.path.insert(0, ROOT)
    import fhip_amd as fhe
    for n_blocks in (0, 1, 5, 8, 1024, 65536):
        for world in (1, 2, 3, 8):
            rs = [fhe.parallel.block_range(r, world, n_blocks) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n_blocks
            assert all(a[1] == b[0] for a, b in zip(rs, rs[1:]))
            assert max(e - s for s, e in rs) - min(e - s for s, e in rs) <= 1
    with pytest.raises(ValueError):
        fhe.parallel.block_range(2, 2, 10)


def test_two_ranks_equal_one_rank():
    fhe, om, orc = _setup()
    make_inputs, compute, digest = _pipeline(fhe, om, orc)
    ref_out, ref_digest, ref_gather = fhe.parallel.run_sharded(compute, make_inputs, NB, digest, gather=True)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in range(2)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1][0] == 3 and res[1][1][0] == 2            # 5 blocks -> 3 + 2
    assert res[0][2] == res[1][2] == ref_digest               # combined digest == single-process digest
    assert res[1][3] is None
    assert np.array_equal(res[0][3], ref_gather.numpy().view(np.uint64))


def _wave_worker(rank, world, port, q, n_waves, slots):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    import fhip_amd as fhe
    seen = []
    g = fhe.parallel.WaveGather((4, 3), torch.int64, "cpu", n_waves, slots=slots,
                                consume=lambda src, w, t: seen.append((src, w, t.clone())))
    for rep in range(2):                                    # two passes over the same rings (reset)
        for w in range(n_waves):
            buf = g.acquire()
            buf.copy_(torch.arange(12).reshape(4, 3) + 1000 * rank + 100 * w + 7 * rep)
            g.commit(w)
        g.finish()
        g.reset() if rep == 0 else None
    q.put((rank, [(s, w, t.numpy().copy()) for s, w, t in seen]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_waves,slots", [(1, 2), (5, 2), (4, 3)])
def test_wave_gather_delivers_every_wave_in_order(n_waves, slots):
    """WaveGather over gloo, world size 2: the root consumes its own and the peer's waves, each exactly once per
    pass, in wave order per source, with the bytes the sender wrote (also when there are more waves than ring slots)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_wave_worker, args=(r, 2, port, q, n_waves, slots)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] == []                                       # only the root consumes
    for src in (0, 1):
        got = [(w, t) for s_, w, t in res[0] if s_ == src]
        assert [w for w, _ in got] == list(range(n_waves)) * 2
        for i, (w, t) in enumerate(got):
            rep = i // n_waves
            assert np.array_equal(t, np.arange(12).reshape(4, 3) + 1000 * src + 100 * w + 7 * rep)


@pytest.mark.parametrize("method,nb", [("wave", NB), ("collective", NB), ("wave", 1), ("collective", 1)])
def test_gather_outputs_in_single_block_waves(method, nb):
    """ragged shards (3 + 2 blocks): moved one block per wave through WaveGather (both sides batched point-to-point, the
    pass's receives posted by its first acquire), and through the dist.gather fallback with padding; nb = 1: ONE block over two
    ranks -- the second rank's shard is empty and still takes part in every transfer"""
    fhe, om, orc = _setup()
    make_inputs, compute, digest = _pipeline(fhe, om, orc)
    ref = compute(make_inputs(0, nb))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q, method, nb)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[1] is None and np.array_equal(res[0], ref.numpy())


def _gather_worker(rank, world, port, q, method, nb=NB):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fhe, om, orc = _setup()
    make_inputs, compute, _ = _pipeline(fhe, om, orc)
    s, e = fhe.parallel.block_range(rank, world, nb)
    g = fhe.parallel.gather_outputs(compute(make_inputs(s, e)), nb, wave_blocks=1, method=method)
    q.put((rank, None if g is None else g.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()
