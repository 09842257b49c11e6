"""GPU parity: the HIP path (through the C ABI, via dirt_amd.rasterise_ops) against the CPU oracle on
the same seeded inputs.  Forward is bit-exact by specification (DESIGN.md "Numeric specification");
gradients accumulated with float atomics are compared PER ELEMENT within 1e-4 of the L1 mass of the terms the
reference adds into that element (tests/parity.py)."""
import numpy as np
import pytest
import torch

from tests import scenes
from dirt_amd import rasterise_ops as ops
from tests import parity

pytestmark = pytest.mark.gpu

GRAD_TOL = 1e-4  # BASELINE.json north_star: "fp32 gradients within 1e-4"


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _batched(s):
    return {k: (v[None] if isinstance(v, np.ndarray) and v.ndim in (2, 3) and k != 'x' else v) for k, v in s.items()}


def _fwd_gpu(s, dev, flags=0):
    return ops._op_rasterise(_t(s['background'], dev), _t(s['vertices'], dev), _t(s['vertex_colors'], dev),
                             _t(s['faces'], dev), s['height'], s['width'], s['channels'], flags=flags).cpu().numpy()


# The library picks the kernels' tile shape from the frame size and the face density; small test frames would
# only ever see the small shape, so the parity tests pin each shape in turn (DIRT_FLAG_TILES_*).
# (the tile-shape flags of the forward kernels, each with one of the gradient kernel's face-loop shapes pinned as well)
TILE_SHAPES = [pytest.param(0, id='auto'), pytest.param(0x200 | 0x2000, id='large-tiles'), pytest.param(0x400 | 0x1000, id='small-tiles'),
               pytest.param(0x400 | 0x4000, id='small-tiles-px1'), pytest.param(0x200 | 0x8000, id='large-tiles-px2'),
               pytest.param(0x10000, id='px4'), pytest.param(0x600 | 0x2000, id='large-tiles-eight-waves')]


def _assert_grad_close(got, ow, key, what, index=None):
    parity.grad_close(got, ow, key, what, index, tol=parity.TIGHT_TOL)   # 5e-6 of the element's own terms (the specification: 1e-4)


def test_square_all_pixels_agree(gpu):
    """tests/square_test.py:54-57 of the reference: exact equality with the analytic mask."""
    s = scenes.square_scene()
    px = ops.rasterise(_t(s['background'], gpu), _t(s['vertices'], gpu), _t(s['vertex_colors'], gpu),
                       _t(s['faces'], gpu), height=128, width=128, channels=1)[:, :, 0].cpu().numpy()
    assert np.all(px == scenes.square_expected()), 'failed: %d pixels disagree' % np.sum(px != scenes.square_expected())


@pytest.mark.parametrize('name,F,H,W,C,seed,rlo,rhi,shared', [
    ('tiny1', 40, 64, 64, 1, 3, 0.05, 0.3, False),
    ('tiny3', 60, 48, 36, 3, 4, 0.05, 0.3, False),       # non multiple-of-32 frame, as tests/rasterise_tests.py:55-56
    ('c4', 500, 128, 160, 4, 5, 0.02, 0.15, False),
    ('c5', 300, 96, 96, 5, 6, 0.02, 0.2, True),
    ('c16', 800, 128, 128, 16, 7, 0.01, 0.1, False),
    ('dense', 3000, 256, 256, 4, 8, 0.005, 0.04, False),  # more faces than one scan round lists
    ('shared', 2000, 200, 120, 3, 9, 0.0, 0.0, True),
])
@pytest.mark.parametrize('tiles', TILE_SHAPES)
def test_forward_bit_exact_and_gradients(gpu, oracle, name, F, H, W, C, seed, rlo, rhi, shared, tiles):
    s = _batched(scenes.rand_scene(F, H, W, C, seed, rlo, rhi, shared))
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    got = _fwd_gpu(s, gpu, tiles)
    assert got.shape == want.shape
    nbad = int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
    assert nbad == 0, '%s: %d of %d pixel values differ from the oracle' % (name, nbad, got.size)

    for flags in (0, 1):
        ow = oracle.backward(s['vertices'], s['faces'], want, s['grad_pixels'], flags=flags, want_debug=True)
        gb, gv, gvc, dbg = ops._op_rasterise_grad(_t(s['vertices'], gpu), _t(s['faces'], gpu), _t(want, gpu),
                                                  _t(s['grad_pixels'], gpu), H, W, C, flags=flags | tiles, want_debug=True)
        assert np.array_equal(gb.cpu().numpy(), ow['grad_background']), name
        _assert_grad_close(gvc.cpu().numpy(), ow, 'grad_vertex_colors', name + ' grad_vertex_colors')
        _assert_grad_close(gv.cpu().numpy(), ow, 'grad_vertices', name + ' grad_vertices')
        assert np.array_equal(dbg.cpu().numpy(), ow['debug_thingy']), name + ' debug_thingy'


def test_visibility_matches_oracle(gpu, oracle):
    s = scenes.rand_scene(700, 150, 130, 1, 11, 0.02, 0.2)
    want, _, _ = oracle.visibility(s['vertices'], s['faces'], 150, 130)
    got = ops._op_visibility(_t(s['vertices'][None], gpu), _t(s['faces'][None], gpu), 150, 130)[0].cpu().numpy()
    assert np.array_equal(got, want)


def test_batch_matches_per_scene(gpu, oracle):
    s = scenes.batch_scene(200, 64, 96, 3, seeds=[21, 22, 23], r_lo=0.03, r_hi=0.2)
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    got = _fwd_gpu(s, gpu)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ow = oracle.backward(s['vertices'], s['faces'], want, s['grad_pixels'])
    gb, gv, gvc, _ = ops._op_rasterise_grad(_t(s['vertices'], gpu), _t(s['faces'], gpu), _t(want, gpu),
                                            _t(s['grad_pixels'], gpu), 64, 96, 3)
    assert np.array_equal(gb.cpu().numpy(), ow['grad_background'])
    _assert_grad_close(gvc.cpu().numpy(), ow, 'grad_vertex_colors', 'batch gvc')
    _assert_grad_close(gv.cpu().numpy(), ow, 'grad_vertices', 'batch gv')


def test_cube_k2(gpu, oracle):
    s = _batched(scenes.cube_scene(256, 256))
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    got = _fwd_gpu(s, gpu)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_autograd_wiring(gpu, oracle):
    """torch.autograd through `rasterise` returns the RasteriseGrad outputs in the reference's
    order: [background, vertices, vertex_colors] (dirt/rasterise_ops.py:124-129)."""
    s = scenes.rand_scene(120, 64, 64, 3, 31, 0.05, 0.3)
    bg = _t(s['background'], gpu).requires_grad_(True)
    v = _t(s['vertices'], gpu).requires_grad_(True)
    vc = _t(s['vertex_colors'], gpu).requires_grad_(True)
    px = ops.rasterise(bg, v, vc, _t(s['faces'], gpu))
    g = _t(s['grad_pixels'], gpu)
    px.backward(g)
    ow = oracle.backward(s['vertices'][None], s['faces'][None], px.detach().cpu().numpy()[None], s['grad_pixels'][None])
    assert np.array_equal(bg.grad.cpu().numpy(), ow['grad_background'][0])
    _assert_grad_close(v.grad.cpu().numpy(), ow, 'grad_vertices', 'autograd gv', 0)
    _assert_grad_close(vc.grad.cpu().numpy(), ow, 'grad_vertex_colors', 'autograd gvc', 0)


def test_autograd_gradients_are_dense(gpu):
    """The reference's grad op returns dense tensors; so does the autograd path here (the state's interleaved accumulators
    are an internal layout): `.view(-1)` works and the gradients do not alias each other or a workspace."""
    s = scenes.rand_scene(80, 40, 56, 4, 32, 0.05, 0.3)
    bg = _t(s['background'], gpu).requires_grad_(True)
    v = _t(s['vertices'], gpu).requires_grad_(True)
    vc = _t(s['vertex_colors'], gpu).requires_grad_(True)
    px = ops.rasterise(bg, v, vc, _t(s['faces'], gpu))
    gb, gv, gvc = torch.autograd.grad(px, [bg, v, vc], _t(s['grad_pixels'], gpu))
    for t_ in (gb, gv, gvc):
        assert t_.is_contiguous()
        assert t_.view(-1).numel() == t_.numel()
    assert gv.untyped_storage().nbytes() == gv.numel() * 4 and gvc.untyped_storage().nbytes() == gvc.numel() * 4


def test_empty_inputs(gpu):
    bg = torch.rand(2, 16, 16, 3, device=gpu)
    out = ops.rasterise_batch(bg, torch.zeros(2, 0, 4, device=gpu), torch.zeros(2, 0, 3, device=gpu),
                              torch.zeros(2, 0, 3, dtype=torch.int32, device=gpu))
    assert torch.equal(out, bg)
    # faces but all degenerate / out of range
    v = torch.zeros(1, 3, 4, device=gpu)
    out = ops.rasterise_batch(bg[:1], v, torch.zeros(1, 3, 3, device=gpu),
                              torch.tensor([[[0, 1, 2], [0, 1, 7]]], dtype=torch.int32, device=gpu))
    assert torch.equal(out, bg[:1])


@pytest.mark.parametrize('tiles', TILE_SHAPES[1:])
def test_golden_fixtures_on_gpu(gpu, tiles):
    """The committed fixtures (tests/golden/make_golden.py) reproduced by the HIP path alone."""
    import glob
    import os
    from tests.golden.make_golden import CASES, make_inputs
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    files = sorted(f for f in glob.glob(os.path.join(here, '*.npz')) if os.path.basename(f) not in ('ref_grads.npz', 'helpers_ref.npz'))
    assert files
    for path in files:
        name = os.path.splitext(os.path.basename(path))[0]
        z = np.load(path)
        s = make_inputs(CASES[name])
        B, H, W, C = s['background'].shape
        s.update(height=H, width=W, channels=C)
        got = _fwd_gpu(s, gpu, tiles)
        assert np.array_equal(got.view(np.uint32), z['pixels'].view(np.uint32)), name
        vis = ops._op_visibility(_t(s['vertices'], gpu), _t(s['faces'], gpu), H, W).cpu().numpy()
        assert np.array_equal(vis, z['face_id']), name
        gb, gv, gvc, _ = ops._op_rasterise_grad(_t(s['vertices'], gpu), _t(s['faces'], gpu), _t(z['pixels'], gpu),
                                                _t(s['grad_pixels'], gpu), H, W, C, flags=tiles)
        assert np.array_equal(gb.cpu().numpy(), z['grad_background']), name
        _assert_grad_close(gv.cpu().numpy(), z, 'grad_vertices', name + ' gv')
        _assert_grad_close(gvc.cpu().numpy(), z, 'grad_vertex_colors', name + ' gvc')


def test_reference_kernel_vectors_on_gpu(gpu):
    """The HIP path against the REFERENCE'S OWN `assemble_grads` (csrc/rasterise_grad_egl.cu:93-236 compiled for the
    host, oracle/make_ref.py): tests/golden/ref_grads.npz holds its outputs for the golden cases (the reference's
    cylinder / bent-square scenes among them).  grad_background and debug_thingy exactly; the atomically summed
    gradients per element within 1e-4 of the terms' L1 mass (the reference's own float32 sum is one ordering)."""
    import os
    from tests.golden.make_golden import CASES, make_inputs
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    ref = np.load(os.path.join(here, 'ref_grads.npz'))
    for name in sorted(CASES):
        z = np.load(os.path.join(here, name + '.npz'))
        s = make_inputs(CASES[name])
        B, H, W, C = s['background'].shape
        gb, gv, gvc, dbg = ops._op_rasterise_grad(_t(s['vertices'], gpu), _t(s['faces'], gpu), _t(z['pixels'], gpu),
                                                  _t(s['grad_pixels'], gpu), H, W, C, want_debug=True)
        assert np.array_equal(gb.cpu().numpy(), ref[name + '/grad_background']), name
        assert np.array_equal(dbg.cpu().numpy(), ref[name + '/debug_thingy']), name
        want = {'grad_vertices': ref[name + '/grad_vertices'], 'grad_vertex_colors': ref[name + '/grad_vertex_colors'],
                'mass_vertices': z['mass_vertices'], 'mass_vertex_colors': z['mass_vertex_colors'], 'cond_vertices': z['cond_vertices']}
        parity.grads_close(gv, gvc, want, name + ' vs reference kernel')


@pytest.mark.parametrize('tiles', TILE_SHAPES[1:])
def test_state_reuse_is_identical(gpu, oracle, tiles):
    """DIRT_FLAG_KEEP_STATE / DIRT_FLAG_REUSE_STATE: the backward pass fed with the forward's records +
    visibility gives the same result as the stateless one that renders again."""
    s = _batched(scenes.rand_scene(600, 100, 140, 4, 41, 0.02, 0.2))
    args = [_t(s[k], gpu) for k in ('background', 'vertices', 'vertex_colors', 'faces')]
    px, state = ops._op_rasterise(*args, 100, 140, 4, keep_state=True, flags=tiles)
    px2 = ops._op_rasterise(*args, 100, 140, 4, flags=tiles)
    assert torch.equal(px, px2)
    g = _t(s['grad_pixels'], gpu)
    a = ops._op_rasterise_grad(args[1], args[3], px, g, 100, 140, 4, state=state, flags=tiles)
    b = ops._op_rasterise_grad(args[1], args[3], px, g, 100, 140, 4, flags=tiles)
    assert torch.equal(a[0], b[0])
    ow = oracle.backward(s['vertices'], s['faces'], px.cpu().numpy(), s['grad_pixels'])
    for got in (a, b):
        _assert_grad_close(got[1].cpu().numpy(), ow, 'grad_vertices', 'gv')
        _assert_grad_close(got[2].cpu().numpy(), ow, 'grad_vertex_colors', 'gvc')


@pytest.mark.parametrize('H,W,C,seed,n_small', [
    (96, 80, 4, 1, 200),
    (70, 50, 3, 2, 1200),   # > 64 faces in one gradient tile: the slot table overflows into the direct-atomic path
    (33, 65, 1, 3, 400),    # 1-channel group: quirk Q1 on hostile geometry
    (64, 64, 5, 4, 300),
])
@pytest.mark.parametrize('tiles', TILE_SHAPES[1:])
def test_hostile_geometry(gpu, oracle, H, W, C, seed, n_small, tiles):
    """Near-plane / w = 0 crossings, faces behind the eye, degenerate and invalid faces, NaN / inf vertices,
    frame-filling and enormous triangles, exact depth ties, sub-pixel clusters: forward bit-exact, visibility
    identical, gradients within tolerance (the cases tests/test_oracle.py pins one by one on the CPU)."""
    s = _batched(scenes.hostile_scene(H, W, C, seed, n_small))
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    got = _fwd_gpu(s, gpu, tiles)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    fid = ops._op_visibility(_t(s['vertices'], gpu), _t(s['faces'], gpu), H, W).cpu().numpy()
    assert np.array_equal(fid[0], oracle.visibility(s['vertices'][0], s['faces'][0], H, W)[0])
    for flags in (0, 1):
        ow = oracle.backward(s['vertices'], s['faces'], want, s['grad_pixels'], flags=flags)
        gb, gv, gvc, _ = ops._op_rasterise_grad(_t(s['vertices'], gpu), _t(s['faces'], gpu), _t(want, gpu),
                                                _t(s['grad_pixels'], gpu), H, W, C, flags=flags | tiles)
        assert np.array_equal(gb.cpu().numpy(), ow['grad_background'])
        _assert_grad_close(gvc.cpu().numpy(), ow, 'grad_vertex_colors', 'grad_vertex_colors')
        _assert_grad_close(gv.cpu().numpy(), ow, 'grad_vertices', 'grad_vertices')
        # ... and at the measured margin (worst element 5.4e-7 of its mass: profiles/r04_tolerance_probe.txt) x 10
        parity.grads_close(gv, gvc, ow, 'hostile tight', tol=5e-6)


@pytest.mark.parametrize('H,W', [(1, 1), (1, 40), (40, 1), (2, 2), (31, 33)])
def test_thin_frames(gpu, oracle, H, W):
    """Frames smaller than a tile / a Scharr stencil: every tap is edge clamped (csrc/rasterise_grad_egl.cu:113-124)
    and no pixel is interior."""
    s = _batched(scenes.rand_scene(30, H, W, 4, 21, 0.2, 0.9))
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    got = _fwd_gpu(s, gpu)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ow = oracle.backward(s['vertices'], s['faces'], want, s['grad_pixels'])
    gb, gv, gvc, _ = ops._op_rasterise_grad(_t(s['vertices'], gpu), _t(s['faces'], gpu), _t(want, gpu),
                                            _t(s['grad_pixels'], gpu), H, W, 4)
    assert np.array_equal(gb.cpu().numpy(), ow['grad_background'])
    _assert_grad_close(gvc.cpu().numpy(), ow, 'grad_vertex_colors', 'grad_vertex_colors')
    _assert_grad_close(gv.cpu().numpy(), ow, 'grad_vertices', 'grad_vertices')


def test_shared_topology(gpu, oracle):
    """One [F,3] `faces` for the whole batch (DIRT_FLAG_SHARED_FACES; the reference tiles it,
    tests/rasterise_tests.py:89) gives exactly what the tiled [B,F,3] tensor gives."""
    B, H, W, C = 3, 60, 84, 3
    base = scenes.rand_scene(300, H, W, C, 31, 0.03, 0.25, True)
    rng = np.random.default_rng(7)
    verts = np.stack([base['vertices'] * (1 + 0.05 * rng.standard_normal(base['vertices'].shape)).astype(np.float32) for _ in range(B)])
    cols = rng.uniform(0, 1, (B,) + base['vertex_colors'].shape).astype(np.float32)
    bg = rng.uniform(0, 1, (B, H, W, C)).astype(np.float32)
    g = rng.standard_normal((B, H, W, C)).astype(np.float32)
    faces = base['faces']
    tiled = np.ascontiguousarray(np.broadcast_to(faces, (B,) + faces.shape))

    def run(f):
        b_, v_, c_ = (_t(a, gpu).requires_grad_(True) for a in (bg, verts, cols))
        px = ops.rasterise_batch(b_, v_, c_, _t(f, gpu))
        px.backward(_t(g, gpu))
        return px.detach().cpu().numpy(), b_.grad.cpu().numpy(), v_.grad.cpu().numpy(), c_.grad.cpu().numpy()

    shared, ref = run(faces), run(tiled)
    assert np.array_equal(shared[0], ref[0]) and np.array_equal(shared[1], ref[1])
    want = oracle.forward(bg, verts, cols, tiled)
    assert np.array_equal(shared[0].view(np.uint32), want.view(np.uint32))
    ow = oracle.backward(verts, tiled, want, g)
    _assert_grad_close(shared[2], ow, 'grad_vertices', 'grad_vertices')
    _assert_grad_close(shared[3], ow, 'grad_vertex_colors', 'grad_vertex_colors')
    vis = ops._op_visibility(_t(verts, gpu), _t(faces, gpu), H, W).cpu().numpy()
    assert np.array_equal(vis, ops._op_visibility(_t(verts, gpu), _t(tiled, gpu), H, W).cpu().numpy())


@pytest.mark.parametrize('H,W', [(48, 16384), (16384, 40)])
def test_maximum_frame_dimension(gpu, oracle, H, W):
    """A frame as wide / as tall as the library accepts (DIRT_MAX_DIM = 16384): bin grid, tile indexing and the
    16-bit boxes at their limits; forward bit-exact, gradients within tolerance."""
    s = _batched(scenes.rand_scene(500, H, W, 3, 51, 0.02, 0.6))
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    got = _fwd_gpu(s, gpu)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ow = oracle.backward(s['vertices'], s['faces'], want, s['grad_pixels'])
    gb, gv, gvc, _ = ops._op_rasterise_grad(_t(s['vertices'], gpu), _t(s['faces'], gpu), _t(want, gpu),
                                            _t(s['grad_pixels'], gpu), H, W, 3)
    assert np.array_equal(gb.cpu().numpy(), ow['grad_background'])
    _assert_grad_close(gvc.cpu().numpy(), ow, 'grad_vertex_colors', 'grad_vertex_colors')
    _assert_grad_close(gv.cpu().numpy(), ow, 'grad_vertices', 'grad_vertices')


@pytest.mark.parametrize('C', [1, 3, 4, 6])
def test_many_dilated_pairs_per_wave(gpu, oracle, C):
    """The gradient kernel lists the dilated (pixel, channel group) pairs of a wave -- which take a neighbour's face
    and barycentrics -- in LDS, 27 to 64 of them depending on the channel count; beyond that they take direct float
    atomics.  A stack of thin occluders at alternating depths makes nearly every pixel of a 32 x 8 region a
    silhouette pixel, far more pairs than the list holds."""
    H, W = 64, 96
    rng = np.random.default_rng(5)
    verts, faces = [], []
    for i in range(0, W, 3):            # vertical slivers, 3 pixels wide, alternating near / far, slightly slanted
        z, w = (0.2, 1.0) if (i // 3) % 2 else (-0.3, 2.0)
        xa, xb = 2.0 * i / W - 1.0, 2.0 * (i + 3) / W - 1.0
        base = len(verts)
        verts += [(xa * w, -1.1 * w, z * w, w), (xb * w, -1.1 * w, z * w, w), (xb * w + 0.01, 1.1 * w, z * w, w), (xa * w + 0.01, 1.1 * w, z * w, w)]
        faces += [(base, base + 1, base + 2), (base, base + 2, base + 3)]
    s = {'vertices': np.asarray(verts, np.float32)[None], 'faces': np.asarray(faces, np.int32)[None]}
    V = s['vertices'].shape[1]
    s['vertex_colors'] = rng.uniform(0, 1, (1, V, C)).astype(np.float32)
    s['background'] = rng.uniform(0, 1, (1, H, W, C)).astype(np.float32)
    s['grad_pixels'] = rng.standard_normal((1, H, W, C)).astype(np.float32)
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    for flags in (0, 1):
        ow = oracle.backward(s['vertices'], s['faces'], want, s['grad_pixels'], flags=flags, want_debug=True)
        assert float((ow['debug_thingy'][..., 0] > 0).mean()) > 0.3   # the scene does what it is meant to
        gb, gv, gvc, dbg = ops._op_rasterise_grad(_t(s['vertices'], gpu), _t(s['faces'], gpu), _t(want, gpu),
                                                  _t(s['grad_pixels'], gpu), H, W, C, flags=flags, want_debug=True)
        assert np.array_equal(gb.cpu().numpy(), ow['grad_background'])
        assert np.array_equal(dbg.cpu().numpy(), ow['debug_thingy'])
        _assert_grad_close(gvc.cpu().numpy(), ow, 'grad_vertex_colors', 'grad_vertex_colors')
        _assert_grad_close(gv.cpu().numpy(), ow, 'grad_vertices', 'grad_vertices')


@pytest.mark.parametrize('W', [32, 33, 34, 35, 61, 64, 65, 67])
@pytest.mark.parametrize('C', [1, 4, 5, 7, 10, 16])
def test_right_border_alias_taps(gpu, oracle, W, C):
    """Quirk Q1 at the right image border: the aliased "channels" 1, 2 of a 1-channel group are the next two elements
    of the flattened slice, which for the last interior columns lie in the NEXT image row (and past the end of the
    tensor for the last rows of the last scene).  Frame widths around the tile width put those columns at every
    position of a strip and of a tile; batch of 2 so that "the next scene" is also exercised."""
    H = 37
    s = scenes.batch_scene(80, H, W, C, seeds=[81, 82], r_lo=0.05, r_hi=0.4)
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    ow = oracle.backward(s['vertices'], s['faces'], want, s['grad_pixels'], want_debug=True)
    gb, gv, gvc, dbg = ops._op_rasterise_grad(_t(s['vertices'], gpu), _t(s['faces'], gpu), _t(want, gpu),
                                              _t(s['grad_pixels'], gpu), H, W, C, want_debug=True)
    assert np.array_equal(gb.cpu().numpy(), ow['grad_background'])
    assert np.array_equal(dbg.cpu().numpy(), ow['debug_thingy'])
    _assert_grad_close(gvc.cpu().numpy(), ow, 'grad_vertex_colors', 'grad_vertex_colors')
    _assert_grad_close(gv.cpu().numpy(), ow, 'grad_vertices', 'grad_vertices')


@pytest.mark.parametrize('C', [2, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 20])
def test_channel_group_passes(gpu, oracle, C):
    """Channel counts other than 1, 3, 4 are cut into passes of whole channel groups (dirt/rasterise_ops.py:148-152: triples
    while >= 3 channels remain, then singles): pairs of triples ({3,3}), whose last pass may carry a triple and the first
    single ({3,1}) or a lone triple, then the remaining singles -- every combination of C // 3 odd / even and C % 3, on a
    frame that is no multiple of the tile, with and without the debug output (one instantiation each), a batch of two.
    grad_background -- written by the first launch's passes in shares where C % 4 == 0, by every pass otherwise -- exact."""
    H, W = 70, 91
    s = scenes.batch_scene(90, H, W, C, seeds=[31 + C, 32 + C], r_lo=0.03, r_hi=0.3)
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    for flags in (0, 1):
        ow = oracle.backward(s['vertices'], s['faces'], want, s['grad_pixels'], flags=flags, want_debug=True)
        for dbg_on in (False, True):
            gb, gv, gvc, dbg = ops._op_rasterise_grad(_t(s['vertices'], gpu), _t(s['faces'], gpu), _t(want, gpu),
                                                      _t(s['grad_pixels'], gpu), H, W, C, flags=flags, want_debug=dbg_on)
            assert np.array_equal(gb.cpu().numpy().view(np.uint32), ow['grad_background'].view(np.uint32)), 'C=%d grad_background' % C
            if dbg_on:
                assert np.array_equal(dbg.cpu().numpy(), ow['debug_thingy']), 'C=%d debug_thingy' % C
            parity.grads_close(gv, gvc, ow, 'C=%d flags=%d' % (C, flags), tol=5e-6)

@pytest.mark.parametrize('C', [1, 3, 4, 5, 16])
def test_dense_outputs_from_the_state(gpu, oracle, C):
    """DIRT_FLAG_DENSE_FROM_STATE (what the autograd path uses): the gradients are summed in the state's interleaved
    accumulators and copied out into dense tensors by the same call -- dense, contiguous, and equal to the strided views
    the plain REUSE_STATE call returns (same kernel, same accumulators: bit for bit up to the atomics' order) and to the
    oracle.  Without REUSE_STATE the flag is ignored."""
    H, W = 75, 100
    s = scenes.batch_scene(120, H, W, C, seeds=[5, 6, 7], r_lo=0.03, r_hi=0.3)
    d = {k: _t(s[k], gpu) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    ow = oracle.backward(s['vertices'], s['faces'], want, s['grad_pixels'])
    px, state = ops._op_rasterise(d['background'], d['vertices'], d['vertex_colors'], d['faces'], H, W, C, keep_state=True)
    gb, gv, gvc, _ = ops._op_rasterise_grad(d['vertices'], d['faces'], px, d['grad_pixels'], H, W, C, state=state, state_outputs='dense')
    assert gv.is_contiguous() and gvc.is_contiguous() and gv.shape == (3, s['vertices'].shape[1], 4) and gvc.shape[-1] == C
    assert gv.data_ptr() < state.data_ptr() or gv.data_ptr() >= state.data_ptr() + state.numel()   # not a view of the state
    assert np.array_equal(gb.cpu().numpy().view(np.uint32), ow['grad_background'].view(np.uint32))
    parity.grads_close(gv, gvc, ow, 'dense from state, C=%d' % C, tol=5e-6)
    from dirt_amd import _lib
    gb2, gv2, gvc2, _ = ops._op_rasterise_grad(d['vertices'], d['faces'], px, d['grad_pixels'], H, W, C, flags=_lib.FLAG_DENSE_FROM_STATE)
    parity.grads_close(gv2, gvc2, ow, 'flag without a state, C=%d' % C, tol=5e-6)


def test_stream_handle_fallback_and_device_guard(gpu, oracle, monkeypatch):
    """The wrappers take the current stream's handle from torch's C binding where it exists and guard the device only
    when it is not current (host-side cost); the public-API fallbacks of both must give the same result."""
    s = _batched(scenes.rand_scene(50, 40, 56, 3, 12, 0.05, 0.3))
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    args = (_t(s['background'], gpu), _t(s['vertices'], gpu), _t(s['vertex_colors'], gpu), _t(s['faces'], gpu), 40, 56, 3)
    assert ops._stream_handle(gpu) == torch.cuda.current_stream(gpu).cuda_stream
    side = torch.cuda.Stream(device=gpu)
    with torch.cuda.stream(side):
        assert ops._stream_handle(gpu) == side.cuda_stream
    monkeypatch.setattr(ops, '_get_raw_stream', None)
    monkeypatch.setattr(ops, '_on_device', lambda dev: torch.cuda.device(dev))
    assert ops._stream_handle(gpu) == torch.cuda.current_stream(gpu).cuda_stream
    got = ops._op_rasterise(*args).cpu().numpy()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_duplicate_index_triples(gpu, oracle):
    """Distinct faces over the SAME three vertex indices count as one face for the dilation test
    (csrc/rasterise_grad_egl.cu:86-89 compares the index triples, not the primitives)."""
    H, W, C = 48, 64, 3
    s = scenes.rand_scene(60, H, W, C, 91, 0.1, 0.5, shared=True)
    faces = np.concatenate([s['faces'], s['faces'][:20], s['faces'][5:15][:, [1, 2, 0]]], 0)  # exact repeats + rotated repeats
    s = _batched(dict(s, faces=faces))
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    got = _fwd_gpu(s, gpu)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    ow = oracle.backward(s['vertices'], s['faces'], want, s['grad_pixels'], want_debug=True)
    gb, gv, gvc, dbg = ops._op_rasterise_grad(_t(s['vertices'], gpu), _t(s['faces'], gpu), _t(want, gpu),
                                              _t(s['grad_pixels'], gpu), H, W, C, want_debug=True)
    assert np.array_equal(gb.cpu().numpy(), ow['grad_background'])
    assert np.array_equal(dbg.cpu().numpy(), ow['debug_thingy'])
    _assert_grad_close(gvc.cpu().numpy(), ow, 'grad_vertex_colors', 'grad_vertex_colors')
    _assert_grad_close(gv.cpu().numpy(), ow, 'grad_vertices', 'grad_vertices')


def test_retain_graph_double_backward_is_pure(gpu, oracle):
    """Two backward passes over one forward (retain_graph=True) each return the RasteriseGrad outputs of their own
    grad_pixels: the reference's grad op is pure (csrc/rasterise_grad_egl.cu:244-250 clears its outputs)."""
    s = scenes.rand_scene(150, 64, 80, 3, 33, 0.05, 0.3)
    bg = _t(s['background'], gpu).requires_grad_(True)
    v = _t(s['vertices'], gpu).requires_grad_(True)
    vc = _t(s['vertex_colors'], gpu).requires_grad_(True)
    px = ops.rasterise(bg, v, vc, _t(s['faces'], gpu))
    g1 = _t(s['grad_pixels'], gpu)
    g2 = torch.flip(g1, dims=(0,)) * 0.5
    a = torch.autograd.grad(px, [bg, v, vc], g1, retain_graph=True)
    a_copy = [t.clone() for t in a]
    b = torch.autograd.grad(px, [bg, v, vc], g2, retain_graph=True)
    c = torch.autograd.grad(px, [bg, v, vc], g1)
    for t, t0 in zip(a, a_copy):
        assert torch.equal(t, t0), 'a later backward changed gradients already returned'
    pxn = px.detach().cpu().numpy()[None]
    for got, g in ((a, g1), (b, g2), (c, g1)):
        ow = oracle.backward(s['vertices'][None], s['faces'][None], pxn, g.cpu().numpy()[None])
        assert np.array_equal(got[0].cpu().numpy(), ow['grad_background'][0])
        _assert_grad_close(got[1].cpu().numpy(), ow, 'grad_vertices', 'gv', 0)
        _assert_grad_close(got[2].cpu().numpy(), ow, 'grad_vertex_colors', 'gvc', 0)


def test_misaligned_views_are_accepted(gpu, oracle):
    """Slices such as x[1:] of a 5x5x3 batch start at addresses that are not multiples of 16; the reference accepts any
    tensor, the C ABI wants 16-byte alignment: the wrapper copies."""
    s = scenes.batch_scene(7, 5, 5, 3, seeds=[1, 2, 3], r_lo=0.2, r_hi=0.9)
    t = {k: _t(s[k], gpu) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    bg, v, vc = (t[k][1:].detach().requires_grad_(True) for k in ('background', 'vertices', 'vertex_colors'))
    assert bg.data_ptr() % 16 != 0
    px = ops.rasterise_batch(bg, v, vc, t['faces'][1:])
    px.backward(t['grad_pixels'][1:])
    want = oracle.forward(s['background'][1:], s['vertices'][1:], s['vertex_colors'][1:], s['faces'][1:])
    assert np.array_equal(px.detach().cpu().numpy().view(np.uint32), want.view(np.uint32))
    ow = oracle.backward(s['vertices'][1:], s['faces'][1:], want, s['grad_pixels'][1:])
    assert np.array_equal(bg.grad.cpu().numpy(), ow['grad_background'])
    _assert_grad_close(v.grad.cpu().numpy(), ow, 'grad_vertices', 'gv')
    _assert_grad_close(vc.grad.cpu().numpy(), ow, 'grad_vertex_colors', 'gvc')


@pytest.mark.parametrize('C', [1, 3, 4, 5, 9, 10, 16])
@pytest.mark.parametrize('shape', [pytest.param(0x2000, id='pairs'), pytest.param(0x1000, id='rows'), pytest.param(0x4000, id='px1'), pytest.param(0x8000, id='px2')])
def test_non_finite_grad_pixels_stay_with_their_own_face(gpu, oracle, C, shape):
    """The reference adds a pixel's terms to the vertices of that pixel's face only (csrc/rasterise_grad_egl.cu:140,228-230):
    a NaN / Inf in grad_pixels makes exactly those vertices' gradients non-finite.  The wave-level reductions here
    multiply by zeroed factors where a pixel is not of the face at hand -- 0 * NaN -- so non-finite pixels take a path of
    their own (dirt_grad.hip: "non-finite factors"; dirt_grad_small.hip selects products).  Every gradient element must
    be non-finite exactly where the oracle's is, and within the tolerance elsewhere; covered, uncovered, border and
    dilated pixels are hit."""
    H, W = 48, 80
    s = scenes.rand_scene(70, H, W, C, 23, 0.05, 0.25)
    g = s['grad_pixels'].copy()
    rng = np.random.default_rng(5)
    for n, val in enumerate([np.nan, np.inf, -np.inf, np.nan, np.inf, np.nan, -np.inf, np.nan]):
        y, x = (0, 3) if n == 0 else (int(rng.integers(0, H)), int(rng.integers(0, W)))
        g[y, x, int(rng.integers(0, C))] = val
    g[17, 40, :] = np.nan          # every channel of one pixel
    b = {k: s[k][None] for k in ('background', 'vertices', 'vertex_colors', 'faces')}
    b['grad_pixels'] = g[None]
    want = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    ow = oracle.backward(b['vertices'], b['faces'], want, b['grad_pixels'])
    assert not np.isfinite(ow['grad_vertices']).all() and not np.isfinite(ow['grad_vertex_colors']).all()
    assert np.isfinite(ow['grad_vertices']).mean() > 0.5            # ... and most of the mesh stays finite
    d = {k: _t(b[k], gpu) for k in b}
    px, state = ops._op_rasterise(d['background'], d['vertices'], d['vertex_colors'], d['faces'], H, W, C, keep_state=True)
    if C >= 5 and shape != 0x2000:
        pytest.skip('many-channel frames have one kernel shape (passes of channel groups)')
    for st in (state, None):
        gb, gv, gvc, _ = ops._op_rasterise_grad(d['vertices'], d['faces'], px, d['grad_pixels'], H, W, C, flags=shape, state=st)
        assert np.array_equal(gb.cpu().numpy().view(np.uint32), ow['grad_background'].view(np.uint32)), 'grad_background'
        parity.grads_close(gv, gvc, ow, 'non-finite grad_pixels, C=%d' % C)
