"""Multi-GPU path, covered on CPU with gloo and world_size 2 (the driver runs the real 8-GPU bench).

The path shards over the batch with no data-path collective (dirt_amd/sharding.py); what there is to
test without GPUs is the host logic: round-robin ownership, shard extraction, broadcast of shared
topology and the optional gather that undoes the placement."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_scenes, results):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dirt_amd import sharding
        full = torch.arange(n_scenes * 6, dtype=torch.float32).reshape(n_scenes, 2, 3)
        mine = sharding.shard_batch(full, rank, world)
        assert mine.shape[0] == len(sharding.scenes_for_rank(n_scenes, rank, world))
        # stand-in for the per-scene render: something only the owner computes
        local = mine * 2 + 1
        out = sharding.gather_batch(local, n_scenes, dst=0)
        faces = torch.arange(12, dtype=torch.int32).reshape(4, 3) if rank == 0 else torch.zeros(4, 3, dtype=torch.int32)
        sharding.broadcast_shared(faces, src=0)
        assert torch.equal(faces, torch.arange(12, dtype=torch.int32).reshape(4, 3))
        if rank == 0:
            assert torch.equal(out, full * 2 + 1)
            results.put('ok')
        else:
            assert out is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_scenes', [5, 8])
def test_round_robin_shard_and_gather_gloo(n_scenes):
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_scenes, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get() == 'ok'


def test_ownership_is_a_partition():
    from dirt_amd import sharding
    for n in (0, 1, 7, 64):
        for world in (1, 2, 8):
            owned = sorted(sum((sharding.scenes_for_rank(n, r, world) for r in range(world)), []))
            assert owned == list(range(n))
    assert sharding.scenes_for_rank(64, 3, 8) == list(range(3, 64, 8))   # K4: 8 scenes per GPU


def test_single_process_is_identity():
    from dirt_amd import sharding
    x = torch.rand(3, 4)
    assert sharding.gather_batch(x, 3) is x
    assert torch.equal(sharding.shard_batch(x, 0, 1), x)
