"""Texture look-up for deferred shaders: the helpers of the reference's samples/textured.py:16-61 over torch
tensors.  Gather-style dense math that runs after the rasteriser, inside `shader_fn`;
differentiable with respect to the texture and (bilinear mode) the coordinates through torch autograd."""
import torch


def uvs_to_pixel_indices(uvs, texture_shape, mode='repeat'):
    """[*, 2] (u, v) with (0, 0) at the TOP-LEFT of the image -> [*, 2] fractional (row, column) indices
    (samples/textured.py:16-26).  `texture_shape` = (height, width)."""
    uvs = uvs.flip(-1)  # x, y coordinates -> y, x indices
    shape = torch.as_tensor(texture_shape, dtype=uvs.dtype, device=uvs.device)
    if mode == 'repeat':
        return torch.remainder(uvs, 1.) * shape
    elif mode == 'clamp':
        return torch.clamp(uvs, 0., 1.) * shape
    raise NotImplementedError(mode)


def sample_texture(texture, indices, mode='bilinear'):
    """texture [Ht, Wt, C], fractional (row, column) `indices` [*, 2] -> [*, C] (samples/textured.py:29-60).

    Bilinear weights are the reference's (fraction of the index, no half-texel shift).  Where the reference's
    `gather_nd` would read row Ht or column Wt (an index in the last texel), the last texel is used instead."""
    ht, wt = texture.shape[0], texture.shape[1]

    def fetch(rows, cols):
        return texture[rows.clamp(0, ht - 1), cols.clamp(0, wt - 1)]

    if mode == 'nearest':
        idx = indices.to(torch.int64)  # truncation, as tf.cast
        return fetch(idx[..., 0], idx[..., 1])
    elif mode == 'bilinear':
        floor = torch.floor(indices)
        frac = indices - floor
        r, c = floor[..., 0].to(torch.int64), floor[..., 1].to(torch.int64)
        fr, fc = frac[..., :1], frac[..., 1:]
        return (fetch(r, c) * (1. - fc) * (1. - fr) + fetch(r, c + 1) * fc * (1. - fr)
                + fetch(r + 1, c) * (1. - fc) * fr + fetch(r + 1, c + 1) * fc * fr)
    raise NotImplementedError(mode)
