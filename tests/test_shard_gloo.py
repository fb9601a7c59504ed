"""N > 1 path on CPU: two gloo ranks shard a batch by sample index and all-gather logits."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mquant_amd import shard


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_samples, vocab, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mine = shard.shard_indices(n_samples, rank, world)
        # "logits" of sample i are a function of i only, so every rank can check the gather
        local = torch.stack([torch.arange(vocab, dtype=torch.float32) + 1000.0 * i for i in mine]) \
            if mine else torch.zeros((0, vocab))
        full = shard.gather_logits(local, n_samples)
        expect = torch.stack([torch.arange(vocab, dtype=torch.float32) + 1000.0 * i for i in range(n_samples)])
        ok = bool(torch.equal(full, expect))
        scales = shard.broadcast_scales([0.5 + rank, 0.25], src=0)
        ok = ok and scales == [0.5, 0.25]
        q.put((rank, ok, mine))
    finally:
        dist.destroy_process_group()


def _run(n_samples, world=2, vocab=17):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_samples, vocab, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res)


def test_shard_indices_cover_every_sample_once():
    for n in (0, 1, 5, 32):
        for world in (1, 2, 8):
            parts = [shard.shard_indices(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert max(len(p) for p in parts) == shard.padded_local_count(n, world) or n == 0


def test_two_ranks_even_batch():
    res = _run(4)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == [0, 2] and res[1][2] == [1, 3]


def test_two_ranks_ragged_batch():
    res = _run(5)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == [0, 2, 4] and res[1][2] == [1, 3]
