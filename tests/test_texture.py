"""The fused texture look-up (dirt_amd.texture.sample_texture_uv; include/dirt_hip.h dirt_texture_sample_*) against the
numpy restatement of the reference's samples/textured.py:16-61 (oracle/texture_oracle.py) and against the composed torch
helpers; the oracle itself against analytic cases on the CPU."""
import numpy as np
import pytest
import torch

from oracle import texture_oracle as tex_oracle


def test_oracle_texel_centres_and_ramps():
    """Sampling at integer indices returns the texel; a texture linear in (row, column) is reproduced exactly between
    texels (no half-texel shift: the reference blends by the fraction of the index)."""
    rng = np.random.default_rng(0)
    ht, wt = 6, 9
    tex = rng.uniform(0, 1, (ht, wt, 3)).astype(np.float32)
    rows, cols = np.meshgrid(np.arange(ht), np.arange(wt), indexing='ij')
    uv = np.stack([cols / wt, rows / ht], -1).astype(np.float32)          # (u, v) = (column, row) / size, top-left origin
    got = tex_oracle.sample_texture_uv(tex, uv)
    assert np.allclose(got, tex, atol=1e-6)
    ramp = (2.0 * rows + 0.5 * cols)[..., None].astype(np.float32) * np.ones(3, np.float32)
    uvf = np.stack([(cols[:-1, :-1] + 0.25) / wt, (rows[:-1, :-1] + 0.75) / ht], -1).astype(np.float32)
    got = tex_oracle.sample_texture_uv(ramp, uvf)
    want = (2.0 * (rows[:-1, :-1] + 0.75) + 0.5 * (cols[:-1, :-1] + 0.25))[..., None] * np.ones(3)
    assert np.allclose(got, want, atol=1e-4)
    # repeat wraps, clamp saturates
    assert np.allclose(tex_oracle.sample_texture_uv(tex, uv + 3.0), tex_oracle.sample_texture_uv(tex, uv), atol=1e-5)
    edge = tex_oracle.sample_texture_uv(tex, np.array([[5.0, -2.0]], np.float32), mode='clamp')
    assert np.allclose(edge[0], tex[0, wt - 1])


def test_oracle_gradient_is_the_finite_difference():
    rng = np.random.default_rng(1)
    tex = rng.uniform(0, 1, (5, 7, 2)).astype(np.float32)
    uv = rng.uniform(0.05, 0.8, (11, 2)).astype(np.float32)
    g = rng.standard_normal((11, 2)).astype(np.float32)
    gt, guv = tex_oracle.sample_texture_uv_grad(tex, uv, g)
    eps = 1e-3
    for i in range(3):
        for k in range(2):
            up, dn = uv.copy(), uv.copy()
            up[i, k] += eps; dn[i, k] -= eps
            fd = ((tex_oracle.sample_texture_uv(tex, up).astype(np.float64) - tex_oracle.sample_texture_uv(tex, dn)) * g).sum() / (2 * eps)
            assert abs(fd - guv[i, k]) <= 2e-2 * max(1.0, abs(fd)), (i, k, fd, guv[i, k])
    t2 = tex.copy(); t2[2, 3, 1] += 0.5
    fd = ((tex_oracle.sample_texture_uv(t2, uv).astype(np.float64) - tex_oracle.sample_texture_uv(tex, uv)) * g).sum() / 0.5
    assert abs(fd - gt[2, 3, 1]) <= 1e-4 * max(1.0, abs(fd))


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['repeat', 'clamp'])
@pytest.mark.parametrize('filt', ['bilinear', 'nearest'])
def test_fused_lookup_matches_oracle_and_composition(gpu, mode, filt):
    from dirt_amd import texture
    rng = np.random.default_rng(3)
    tex = rng.uniform(0, 1, (37, 53, 3)).astype(np.float32)
    uv = rng.uniform(-1.5, 2.5, (48, 64, 2)).astype(np.float32)
    uv[0, :8] = [[0.0, 0.0], [1.0, 1.0], [0.999999, 0.5], [-1e-9, 0.3], [0.5, -1e-9], [1.0, 0.0], [0.0, 1.0], [2.0, -3.0]]
    want = tex_oracle.sample_texture_uv(tex, uv, mode, filt)
    t, u = torch.from_numpy(tex).to(gpu), torch.from_numpy(uv).to(gpu)
    got = texture.sample_texture_uv(t, u, mode, filt)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), 'differs from the oracle'
    composed = texture.sample_texture(t, texture.uvs_to_pixel_indices(u, t.shape[:2], mode), filt)
    assert torch.equal(got, composed), 'differs from the composed torch helpers'


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['repeat', 'clamp'])
def test_fused_lookup_gradients_in_place_from_a_gbuffer(gpu, mode):
    """(u, v) read in place from channels 1:3 of a 6-channel G-buffer (samples/textured.py:120-122), gradients against the
    oracle and against autograd through the composed helpers."""
    from dirt_amd import texture
    rng = np.random.default_rng(4)
    tex = rng.uniform(0, 1, (20, 31, 3)).astype(np.float32)
    gbuf = rng.uniform(-0.4, 1.4, (40, 56, 6)).astype(np.float32)
    g = rng.standard_normal((40, 56, 3)).astype(np.float32)
    t = torch.from_numpy(tex).to(gpu).requires_grad_(True)
    gb = torch.from_numpy(gbuf).to(gpu).requires_grad_(True)
    out = texture.sample_texture_uv(t, gb[..., 1:3], mode)
    out.backward(torch.from_numpy(g).to(gpu))
    want_t, want_uv = tex_oracle.sample_texture_uv_grad(tex, gbuf[..., 1:3], g, mode)
    assert float(np.abs(t.grad.cpu().numpy() - want_t).max()) <= 1e-4 * max(1.0, float(np.abs(want_t).max()))
    got_uv = gb.grad.cpu().numpy()
    assert float(np.abs(got_uv[..., 1:3] - want_uv).max()) <= 1e-4 * max(1.0, float(np.abs(want_uv).max()))
    assert not got_uv[..., 0].any() and not got_uv[..., 3:].any()
    t2 = torch.from_numpy(tex).to(gpu).requires_grad_(True)
    gb2 = torch.from_numpy(gbuf).to(gpu).requires_grad_(True)
    texture.sample_texture(t2, texture.uvs_to_pixel_indices(gb2[..., 1:3], t2.shape[:2], mode)).backward(torch.from_numpy(g).to(gpu))
    assert torch.allclose(t.grad, t2.grad, atol=1e-4, rtol=1e-4)
    assert torch.allclose(gb.grad, gb2.grad, atol=1e-3, rtol=1e-4)


def test_argument_shapes_are_checked():
    """`[..., 3]` coordinates or a 2-D texture are refused instead of being read as the wrong pairs (no GPU needed: the
    checks come before any device work)."""
    from dirt_amd import texture as tex
    with pytest.raises(ValueError):
        tex.sample_texture_uv(torch.zeros(4, 4, 3), torch.zeros(5, 3))
    with pytest.raises(ValueError):
        tex.sample_texture_uv(torch.zeros(4, 4), torch.zeros(5, 2))


@pytest.mark.gpu
@pytest.mark.parametrize('ct', [1, 3, 4, 5])
@pytest.mark.parametrize('mode', ['repeat', 'clamp'])
def test_gradient_of_a_smooth_uv_image_takes_the_patch_path(gpu, ct, mode):
    """The backward kernel works on 16 x 16-pixel tiles of the look-up image and sums each tile's contributions in an LDS copy of
    the texture patch they fall into (dirt_texture.hip, round 6); tiles whose patch is too large -- the `repeat` seam of this
    field, random coordinates -- scatter float atomics as before.  A smooth, rotated (u, v) field with a seam across the
    frame exercises both, for every channel-count specialisation (1, 3, 4; 5 = the generic passes), an odd image size and a
    batch of two images; values against the oracle's float64 gradient."""
    from dirt_amd import texture
    rng = np.random.default_rng(11 + ct)
    H, W, Ht, Wt = 75, 93, 64, 48
    tex = rng.uniform(0, 1, (Ht, Wt, ct)).astype(np.float32)
    ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
    uvs = []
    for b, (ang, scale) in enumerate(((0.3, 1.7), (-0.2, 0.6))):
        c, s = np.cos(ang), np.sin(ang)
        u = (c * xs / W + s * ys / H) * scale - 0.2
        v = (-s * xs / W + c * ys / H) * scale + 0.1
        uvs.append(np.stack([u, v], -1))
    uv = np.stack(uvs).astype(np.float32)                     # [2, H, W, 2]
    g = rng.standard_normal((2, H, W, ct)).astype(np.float32)
    t = torch.from_numpy(tex).to(gpu).requires_grad_(True)
    u_t = torch.from_numpy(uv).to(gpu).requires_grad_(True)
    out = texture.sample_texture_uv(t, u_t, mode)
    assert np.array_equal(out.detach().cpu().numpy().view(np.uint32), tex_oracle.sample_texture_uv(tex, uv, mode).view(np.uint32))
    out.backward(torch.from_numpy(g).to(gpu))
    want_t, want_uv = tex_oracle.sample_texture_uv_grad(tex, uv, g, mode)
    assert float(np.abs(t.grad.cpu().numpy() - want_t).max()) <= 2e-5 * max(1.0, float(np.abs(want_t).max()))
    assert float(np.abs(u_t.grad.cpu().numpy() - want_uv).max()) <= 1e-4 * max(1.0, float(np.abs(want_uv).max()))
    # nearest: the gradient of a gather
    t2 = torch.from_numpy(tex).to(gpu).requires_grad_(True)
    texture.sample_texture_uv(t2, torch.from_numpy(uv).to(gpu), mode, 'nearest').backward(torch.from_numpy(g).to(gpu))
    t3 = torch.from_numpy(tex).to(gpu).requires_grad_(True)
    texture.sample_texture(t3, texture.uvs_to_pixel_indices(torch.from_numpy(uv).to(gpu), t3.shape[:2], mode), 'nearest').backward(torch.from_numpy(g).to(gpu))
    assert torch.allclose(t2.grad, t3.grad, atol=2e-5 * max(1.0, float(t3.grad.abs().max())), rtol=1e-5)
