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


def _sharded_worker(rank, world, port, n_scenes, shared_faces, results):
    """rasterise_batch_sharded end to end over gloo with a CPU stand-in for the per-scene render (the op itself is GPU-only):
    which scenes a rank renders, what it hands the op (an EMPTY batch where it owns nothing), the gather that undoes the
    round-robin placement with uneven shards."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dirt_amd import sharding, rasterise_ops
        seen = []

        def fake_rasterise_batch(background, vertices, vertex_colors, faces, *a, **k):
            assert background.shape[0] == vertices.shape[0] == vertex_colors.shape[0]
            assert faces.dim() == 2 or faces.shape[0] == background.shape[0]
            seen.append(int(background.shape[0]))
            w = vertices.sum(dim=(1, 2)).view(-1, 1, 1, 1) + vertex_colors.sum(dim=(1, 2)).view(-1, 1, 1, 1)
            tag = faces.sum().float() if faces.dim() == 2 else faces.sum(dim=(1, 2)).float().view(-1, 1, 1, 1)
            return background * 2 + w + tag

        rasterise_ops.rasterise_batch = fake_rasterise_batch
        g = torch.Generator().manual_seed(1234)   # every rank holds the same replicated batch
        bg = torch.rand(n_scenes, 3, 4, 2, generator=g)
        v = torch.rand(n_scenes, 5, 4, generator=g)
        vc = torch.rand(n_scenes, 5, 2, generator=g)
        f = torch.randint(0, 5, (6, 3), generator=g, dtype=torch.int32) if shared_faces else torch.randint(0, 5, (n_scenes, 6, 3), generator=g, dtype=torch.int32)
        out = sharding.rasterise_batch_sharded(bg, v, vc, f, gather=True)
        assert seen == [len(sharding.scenes_for_rank(n_scenes, rank, world))]
        if rank == 0:
            assert out.shape == bg.shape and torch.allclose(out, fake_rasterise_batch(bg, v, vc, f))
            results.put('ok')
        else:
            assert out is None
        local = sharding.rasterise_batch_sharded(bg, v, vc, f)   # outputs stay sharded by default
        assert local.shape[0] == len(sharding.scenes_for_rank(n_scenes, rank, world))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_scenes,shared_faces', [(5, False), (7, True), (3, False)])
def test_sharded_render_and_gather_gloo_world4_uneven(n_scenes, shared_faces):
    """World size 4 with 5, 7 and 3 scenes: ranks own two, one or NO scene (tests/multi_gpu_test.py:22-29 of the reference
    places whole graphs on devices; here scene s lives on rank s mod N)."""
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 4, port, n_scenes, shared_faces, q)) for r in range(4)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get() == 'ok'
