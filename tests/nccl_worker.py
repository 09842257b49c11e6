"""Worker of tests/test_gpu_configs.py::test_two_gpus_over_rccl (launched under torch.distributed.run, one rank per GPU):
the analogue of the reference's tests/multi_gpu_test.py:22-29 -- the same op on two devices -- on the sharded path:
rank 0 broadcasts the shared topology over RCCL, every rank renders and differentiates its scenes on its own GPU, rank 0
gathers the pixels and compares them with the whole batch rendered on its own."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # dmabuf IPC: what RCCL needs on this pool's host driver
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from dirt_amd import sharding
from tests import scenes  # noqa: E402
from dirt_amd import rasterise_ops as ops  # noqa: E402


def main():
    rank, world, local = int(os.environ['RANK']), int(os.environ['WORLD_SIZE']), int(os.environ['LOCAL_RANK'])
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    import datetime
    dist.init_process_group('nccl', device_id=dev, timeout=datetime.timedelta(seconds=180))
    n_scenes, H, W, C = 5, 96, 128, 3
    base = scenes.rand_scene(300, H, W, C, 31, 0.03, 0.25, True)
    rng = np.random.default_rng(7)   # the same on every rank: a replicated batch, of which each rank takes its share
    verts = np.stack([base['vertices'] * (1 + 0.05 * rng.standard_normal(base['vertices'].shape)).astype(np.float32) for _ in range(n_scenes)])
    cols = rng.uniform(0, 1, (n_scenes,) + base['vertex_colors'].shape).astype(np.float32)
    bg = rng.uniform(0, 1, (n_scenes, H, W, C)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    faces = t(base['faces']) if rank == 0 else torch.zeros(base['faces'].shape, dtype=torch.int32, device=dev)
    sharding.broadcast_shared(faces, src=0)
    assert torch.equal(faces.cpu(), torch.from_numpy(base['faces']))
    v_local = sharding.shard_batch(t(verts), rank, world).clone().requires_grad_(True)
    local_px = ops.rasterise_batch(sharding.shard_batch(t(bg), rank, world), v_local, sharding.shard_batch(t(cols), rank, world), faces)
    local_px.sum().backward()
    assert torch.isfinite(v_local.grad).all() and bool((v_local.grad[..., 2] == 0).all())
    full = sharding.gather_batch(local_px.detach(), n_scenes, dst=0)
    if rank == 0:
        want = ops.rasterise_batch(t(bg), t(verts), t(cols), faces)
        assert torch.equal(full, want), 'gathered shards differ from the single-GPU batch'
        print('nccl_worker ok: %d ranks, %d scenes' % (world, n_scenes))
    else:
        assert full is None
    # the collectives themselves on device tensors, whatever the world size (a world of ONE still initialises RCCL, its IPC
    # path and the device-side kernels for real: what the one-GPU boxes can exercise)
    ones = torch.ones(4, device=dev)
    dist.all_reduce(ones)
    assert torch.equal(ones.cpu(), torch.full((4,), float(world)))
    dist.broadcast(ones, src=0)
    piece = torch.full((3,), float(rank), device=dev)
    bufs = [torch.empty_like(piece) for _ in range(world)] if rank == 0 else None
    dist.gather(piece, bufs, dst=0)
    if rank == 0:
        assert all(torch.equal(b.cpu(), torch.full((3,), float(r))) for r, b in enumerate(bufs))
        print('nccl_worker collectives ok: RCCL %s' % '.'.join(str(x) for x in torch.cuda.nccl.version()))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
