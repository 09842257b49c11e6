"""Batch-dimension sharding of `rasterise_batch` across the GPUs of one node.

Scenes of a batch are independent (dirt/rasterise_ops.py:56-63: `rasterise_batch` is defined as a
stack of per-scene renders; `iib` indexes every access in csrc/rasterise_grad_egl.cu:107-230), so the
path shards with NO data-path collective: one process per GPU, scene s lives on rank s mod N, inputs
and outputs stay sharded.  RCCL (torch.distributed backend "nccl") is used only for the optional
convenience of broadcasting shared inputs from rank 0 and gathering results to rank 0; a pixel
gather is per-link bound over xGMI and is never part of the timed path (DESIGN.md "Multi-GPU").
"""
import torch
import torch.distributed as dist


def scenes_for_rank(n_scenes, rank, world_size):
    """Indices of the scenes rank `rank` owns: s mod world_size == rank (round-robin)."""
    return list(range(rank, n_scenes, world_size))


def shard_batch(tensor, rank, world_size):
    """The local shard [n_local, ...] of a replicated batch tensor [n_scenes, ...]."""
    idx = scenes_for_rank(tensor.shape[0], rank, world_size)
    return tensor[idx] if idx else tensor[:0]


def broadcast_shared(tensor, src=0, group=None):
    """Broadcast a tensor every scene shares (e.g. one `faces` topology) from rank `src`."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(tensor, src=src, group=group)
    return tensor


def gather_batch(local, n_scenes, dst=0, group=None):
    """Collect per-rank shards [n_local, ...] into the full batch [n_scenes, ...] on rank `dst`
    (None elsewhere), undoing the round-robin placement.  Convenience only: see module docstring."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    n_max = (n_scenes + world - 1) // world
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.empty((n_scenes,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = scenes_for_rank(n_scenes, r, world)
        if idx:
            out[idx] = bufs[r][:len(idx)]
    return out


def rasterise_batch_sharded(background, vertices, vertex_colors, faces, group=None, gather=False, rank=None, world_size=None):
    """Render this rank's share of a replicated batch.  Returns the local pixels [n_local,H,W,C]
    (or, with gather=True, the full batch on rank 0 and None elsewhere).  `faces` may be [n_scenes,F,3] (sharded like
    the other inputs) or one [F,3] topology shared by every scene (handed on unchanged).  `rank` / `world_size`
    default to the process group's; given explicitly they select the share without any process group (tests)."""
    from .rasterise_ops import rasterise_batch
    if rank is None or world_size is None:
        if dist.is_available() and dist.is_initialized():
            world_size, rank = dist.get_world_size(group), dist.get_rank(group)
        else:
            world_size, rank = 1, 0
    n = background.shape[0]
    local_faces = shard_batch(faces, rank, world_size) if faces.dim() == 3 else faces
    local = rasterise_batch(shard_batch(background, rank, world_size), shard_batch(vertices, rank, world_size),
                            shard_batch(vertex_colors, rank, world_size), local_faces)
    return gather_batch(local, n, 0, group) if gather else local
