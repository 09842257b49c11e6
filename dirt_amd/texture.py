"""Texture look-up for deferred shaders: the helpers of the reference's samples/textured.py:16-61.

`sample_texture_uv` is the fused path -- one HIP kernel for `sample_texture(texture, uvs_to_pixel_indices(uvs, shape,
mode), filter)` and one for its gradient (include/dirt_hip.h: dirt_texture_sample_forward / _backward), reading the
(u, v) pairs in place from a G-buffer slice.  `uvs_to_pixel_indices` and `sample_texture` are the reference's two
functions over torch tensors (same names and arguments), kept for scripts that call them separately; both routes give
the same values bit for bit."""
import torch

from . import _lib
from . import rasterise_ops as _ops


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


def _pairs_in_place(uvs):
    """(tensor to pass, element stride between pairs) such that pair i starts at data_ptr + 4 * i * stride: a slice
    `gbuffer[..., a:a+2]` of a contiguous G-buffer is read in place, anything else is made contiguous."""
    if uvs.stride(-1) == 1 and uvs.dim() >= 2:
        step = uvs.stride(-2)
        ok = step >= 2
        expect = step
        for d in range(uvs.dim() - 2, -1, -1):
            if uvs.stride(d) != expect:
                ok = False
                break
            expect *= uvs.shape[d]
        if ok:
            return uvs, step
    return uvs.contiguous(), 2


class _SampleTextureUV(torch.autograd.Function):
    @staticmethod
    def forward(ctx, texture, uvs, flags):
        lib = _lib.load()
        if not (texture.is_cuda and uvs.is_cuda):
            raise RuntimeError('dirt_amd.texture.sample_texture_uv runs on an MI355X only; there is no CPU fallback')
        texture = texture.contiguous()
        src, stride = _pairs_in_place(uvs)
        n = uvs.numel() // 2
        ht, wt, ct = (int(d) for d in texture.shape)
        out = torch.empty(tuple(uvs.shape[:-1]) + (ct,), dtype=torch.float32, device=texture.device)
        with _ops._on_device(texture.device):
            rc = lib.dirt_texture_sample_forward(texture.data_ptr(), src.data_ptr(), out.data_ptr(), n, ht, wt, ct, stride, flags,
                                                 _ops._stream_handle(texture.device))
        if rc:
            raise ValueError(lib.dirt_texture_last_error().decode())
        ctx.save_for_backward(texture, src)
        ctx.meta = (stride, flags, tuple(uvs.shape))
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        lib = _lib.load()
        texture, src = ctx.saved_tensors
        stride, flags, uv_shape = ctx.meta
        ht, wt, ct = (int(d) for d in texture.shape)
        n = 1
        for d in uv_shape[:-1]:
            n *= int(d)
        grad_out = grad_out.contiguous().to(torch.float32)
        grad_texture = torch.empty_like(texture)
        grad_uvs = torch.empty(uv_shape, dtype=torch.float32, device=texture.device) if ctx.needs_input_grad[1] else None
        # the look-ups as an image: the last pixel axis is a row, everything before it stacks rows ([H, W, 2]; [B, H, W, 2]);
        # a flat [n, 2] list is one row
        cols = int(uv_shape[-2]) if len(uv_shape) >= 3 else n
        rows = n // cols if cols else 0
        with _ops._on_device(texture.device):
            rc = lib.dirt_texture_sample_backward_image(texture.data_ptr(), src.data_ptr(), grad_out.data_ptr(), grad_texture.data_ptr(),
                                                        grad_uvs.data_ptr() if grad_uvs is not None else None, rows, cols, ht, wt, ct, stride, 2,
                                                        flags, _ops._stream_handle(texture.device))
        if rc:
            raise ValueError(lib.dirt_texture_last_error().decode())
        return grad_texture, grad_uvs, None


def sample_texture_uv(texture, uvs, mode='repeat', filter='bilinear'):
    """Fused `sample_texture(texture, uvs_to_pixel_indices(uvs, texture.shape[:2], mode), filter)`
    (samples/textured.py:16-61, as its shader_fn uses them, :116-141): texture [Ht, Wt, C] float32, uvs [*, 2] with
    (0, 0) at the top-left of the image -> [*, C].  Differentiable with respect to the texture and the coordinates."""
    flags = {'repeat': 0, 'clamp': _lib.TEX_CLAMP}[mode] | {'bilinear': 0, 'nearest': _lib.TEX_NEAREST}[filter]
    if texture.dim() != 3:
        raise ValueError('sample_texture_uv expects texture to be 3D [height, width, channels], got shape %s' % (tuple(texture.shape),))
    if uvs.dim() < 1 or uvs.shape[-1] != 2:
        raise ValueError('sample_texture_uv expects uvs of shape [..., 2], got %s' % (tuple(uvs.shape),))
    if texture.device != uvs.device:
        raise ValueError('texture and uvs must be on the same device (%s vs %s)' % (texture.device, uvs.device))
    return _SampleTextureUV.apply(texture.to(torch.float32), uvs.to(torch.float32), flags)
