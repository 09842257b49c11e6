"""CPU restatement (numpy) of the texture look-up of the reference's samples/textured.py -- TEST INFRASTRUCTURE: only
tests/ may import it; dirt_amd never does.

`sample_texture_uv` follows `uvs_to_pixel_indices` (samples/textured.py:16-26) and `sample_texture`
(samples/textured.py:29-60) operation for operation in float32; `sample_texture_uv_grad` is the analytic gradient of that
expression, accumulated in float64.  Where the reference's gather_nd would read row Ht / column Wt (an index inside the
last texel) the last texel is used (a documented choice of this build; TF's GPU gather_nd returns zeros there, its CPU
kernel raises).  Parity: the reference ships no expected values for its samples; these functions are pinned by the
reference's own two functions executed over a numpy stand-in for TensorFlow (tests/test_helpers_ref.py, committed
vectors tests/golden/helpers_ref.npz: equal wherever the reference's gather_nd stays inside the texture) and by the
analytic cases in tests/test_texture.py (texel centres, linear ramps)."""
import numpy as np


def _indices(uvs, ht, wt, mode):
    uvs = np.asarray(uvs, np.float32)[..., ::-1]                       # :20 x, y coordinates -> y, x indices
    shape = np.array([ht, wt], np.float32)
    if mode == 'repeat':
        return ((uvs - np.floor(uvs)).astype(np.float32) * shape).astype(np.float32)   # :22 uvs % 1. * texture_shape
    if mode == 'clamp':
        return (np.clip(uvs, np.float32(0), np.float32(1)) * shape).astype(np.float32)  # :24
    raise NotImplementedError(mode)


def sample_texture_uv(texture, uvs, mode='repeat', filter='bilinear'):
    texture = np.asarray(texture, np.float32)
    ht, wt = texture.shape[:2]
    idx = _indices(uvs, ht, wt, mode)
    if filter == 'nearest':
        r = np.clip(idx[..., 0].astype(np.int64), 0, ht - 1)            # :33 tf.cast(indices, tf.int32): truncation
        c = np.clip(idx[..., 1].astype(np.int64), 0, wt - 1)
        return texture[r, c]
    fl = np.floor(idx)                                                   # :37
    frac = (idx - fl).astype(np.float32)                                 # :38
    r0 = np.clip(fl[..., 0].astype(np.int64), 0, ht - 1)
    c0 = np.clip(fl[..., 1].astype(np.int64), 0, wt - 1)
    r1, c1 = np.minimum(r0 + 1, ht - 1), np.minimum(c0 + 1, wt - 1)
    fr, fc = frac[..., :1], frac[..., 1:]
    one = np.float32(1)
    tl, tr, bl, br = texture[r0, c0], texture[r0, c1], texture[r1, c0], texture[r1, c1]
    return (((tl * (one - fc)) * (one - fr) + (tr * fc) * (one - fr)) + (bl * (one - fc)) * fr) + (br * fc) * fr   # :53-57


def sample_texture_uv_grad(texture, uvs, grad_out, mode='repeat'):
    """-> (grad_texture [Ht,Wt,C], grad_uvs [*,2]) of the bilinear look-up, float64 accumulation."""
    texture = np.asarray(texture, np.float64)
    ht, wt, ct = texture.shape
    uvs32 = np.asarray(uvs, np.float32)
    idx = _indices(uvs32, ht, wt, mode).astype(np.float64)
    g = np.asarray(grad_out, np.float64).reshape(-1, ct)
    idx2 = idx.reshape(-1, 2)
    fl = np.floor(idx2)
    fr, fc = idx2[:, 0] - fl[:, 0], idx2[:, 1] - fl[:, 1]
    r0 = np.clip(fl[:, 0].astype(np.int64), 0, ht - 1); c0 = np.clip(fl[:, 1].astype(np.int64), 0, wt - 1)
    r1, c1 = np.minimum(r0 + 1, ht - 1), np.minimum(c0 + 1, wt - 1)
    gt = np.zeros_like(texture)
    for (rr, cc, w) in ((r0, c0, (1 - fc) * (1 - fr)), (r0, c1, fc * (1 - fr)), (r1, c0, (1 - fc) * fr), (r1, c1, fc * fr)):
        np.add.at(gt, (rr, cc), g * w[:, None])
    tl, tr, bl, br = texture[r0, c0], texture[r0, c1], texture[r1, c0], texture[r1, c1]
    d_fr = (g * ((bl - tl) * (1 - fc)[:, None] + (br - tr) * fc[:, None])).sum(-1)
    d_fc = (g * ((tr - tl) * (1 - fr)[:, None] + (br - bl) * fr[:, None])).sum(-1)
    u, v = uvs32.reshape(-1, 2)[:, 0], uvs32.reshape(-1, 2)[:, 1]
    if mode == 'clamp':
        du = np.where((u >= 0) & (u <= 1), wt, 0.0); dv = np.where((v >= 0) & (v <= 1), ht, 0.0)
    else:
        du = np.full_like(d_fc, wt); dv = np.full_like(d_fr, ht)
    guv = np.stack([d_fc * du, d_fr * dv], -1).reshape(uvs32.shape)
    return gt.astype(np.float32), guv.astype(np.float32)
