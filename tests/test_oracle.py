"""Pins the CPU oracle (oracle/dirt_oracle.c) -- runs without a GPU.

The reference holds exactly one known-answer test for this path, tests/square_test.py (forward only);
it has no gradient test and no golden vectors.  The oracle is pinned against:
  * that known answer and the facts that follow from it;
  * exact rational arithmetic for coverage, analytic formulas for perspective interpolation;
  * an independent numpy restatement of assemble_grads (csrc/rasterise_grad_egl.cu:93-236);
  * invariants that follow from the CUDA source, and finite differences for the derivatives that
    are exact (colour, background);
  * the committed fixtures under tests/golden/ (drift detection).
"""
from fractions import Fraction
import os

import numpy as np
import pytest

from tests import scenes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _fwd(oracle, s):
    return oracle.forward(s['background'][None], s['vertices'][None], s['vertex_colors'][None], s['faces'][None])[0]


# ---------------------------------------------------------------------------------------------- forward

def test_square_known_answer(oracle):
    """tests/square_test.py:11-17,54-57: exact equality, 256 pixels, rows 56-71 x cols 24-39."""
    s = scenes.square_scene()
    px = _fwd(oracle, s)[:, :, 0]
    exp = scenes.square_expected()
    assert np.all(px == exp), 'failed: %d pixels disagree' % np.sum(px != exp)
    rows, cols = np.nonzero(px)
    assert px.sum() == 256 and rows.min() == 56 and rows.max() == 71 and cols.min() == 24 and cols.max() == 39
    assert set(np.unique(px)) == {0.0, 1.0}


def test_square_diagonal_is_watertight(oracle):
    """16 pixel centres lie exactly on the shared diagonal v0->v2 (App. B): each is drawn by exactly
    one of the two faces, whichever way round the faces are listed or wound."""
    s = scenes.square_scene()
    for faces in ([[0, 1, 2], [0, 2, 3]], [[0, 2, 3], [0, 1, 2]], [[2, 1, 0], [3, 2, 0]], [[1, 2, 0], [2, 3, 0]]):
        f = np.array(faces, np.int32)
        fid, _, _ = oracle.visibility(s['vertices'], f, 128, 128)
        assert (fid >= 0).sum() == 256
        a = oracle.visibility(s['vertices'], f[:1], 128, 128)[0] >= 0
        b = oracle.visibility(s['vertices'], f[1:], 128, 128)[0] >= 0
        assert not np.any(a & b), 'a sample on the shared edge was drawn twice'
        assert (a | b).sum() == 256


def test_orientation_y_up_x_right(oracle):
    """Clip y=+1 is the TOP row of the image, x=+1 the right column (README.md:183; the vertical flip
    of csrc/rasterise_egl.cu:23,80).  The square test cannot see the flip (its mask is v-symmetric)."""
    v = np.array([[0.5, 0.5, 0, 1], [0.9, 0.5, 0, 1], [0.7, 0.9, 0, 1]], np.float32)
    fid, _, _ = oracle.visibility(v, np.array([[0, 1, 2]], np.int32), 40, 60)
    rows, cols = np.nonzero(fid >= 0)
    assert rows.max() < 10 + 1 and rows.min() >= 2      # y in [0.5,0.9] -> rows (1-y)/2*40 in [2,10]
    assert cols.min() >= 45 and cols.max() <= 57        # x in [0.5,0.9] -> cols (x+1)/2*60 in [45,57]


def test_depth_less_and_tie_break(oracle):
    """GL_LESS (GL default; only DEPTH_TEST is enabled, csrc/rasterise_egl.cpp:213): the nearer face
    wins whatever the draw order; on exactly equal depth the earlier face wins."""
    quad = lambda z: [[-0.5, -0.5, z, 1], [0.5, -0.5, z, 1], [0.5, 0.5, z, 1], [-0.5, 0.5, z, 1]]
    v = np.array(quad(0.3) + quad(-0.2), np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3], [4, 5, 6], [4, 6, 7]], np.int32)
    fid, _, _ = oracle.visibility(v, f, 32, 32)
    assert set(np.unique(fid)) == {-1, 2, 3}            # z=-0.2 is nearer (depth range [0,1], z_win=(z+1)/2)
    fid, _, _ = oracle.visibility(v, f[[2, 3, 0, 1]], 32, 32)
    assert set(np.unique(fid)) == {-1, 0, 1}
    v2 = np.array(quad(0.1) + quad(0.1), np.float32)
    fid, _, _ = oracle.visibility(v2, f, 32, 32)
    assert set(np.unique(fid)) == {-1, 0, 1}            # tie -> earlier faces


def test_depth_clip_and_behind_eye(oracle):
    quad = lambda z, w: [[-0.5 * w, -0.5 * w, z, w], [0.5 * w, -0.5 * w, z, w], [0.5 * w, 0.5 * w, z, w], [-0.5 * w, 0.5 * w, z, w]]
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    for z, w, visible in ((0.0, 1.0, True), (1.5, 1.0, False), (-1.5, 1.0, False), (0.0, -1.0, False), (0.99, 1.0, True)):
        fid, _, _ = oracle.visibility(np.array(quad(z, w), np.float32), f, 16, 16)
        assert bool((fid >= 0).any()) == visible, (z, w)
    # a triangle crossing the near plane / w=0: only the part in front with -w <= z <= w is drawn
    v = np.array([[-0.5, -0.8, -0.5, 1.0], [0.5, -0.8, -0.5, 1.0], [0.0, 3.0, 2.0, -1.0]], np.float32)
    fid, bary, cw = oracle.visibility(v, np.array([[0, 1, 2]], np.int32), 64, 64)
    assert (fid >= 0).any() and np.all(cw[fid >= 0] > 0)


def _exact_cover(verts, face, H, W, i, r):
    """Exact rational evaluation of the coverage rule of the specification at pixel (i, r)."""
    P = [tuple(Fraction(float(c)) for c in verts[k]) for k in face]
    X = [(p[0] + p[3]) * Fraction(W, 2) for p in P]
    Y = [(p[1] + p[3]) * Fraction(H, 2) for p in P]
    Wc = [p[3] for p in P]
    px, py = Fraction(2 * i + 1, 2), Fraction(2 * (H - 1 - r) + 1, 2)
    coef = []
    for k in range(3):
        p, q = (k + 1) % 3, (k + 2) % 3
        coef.append((Y[p] * Wc[q] - Wc[p] * Y[q], Wc[p] * X[q] - X[p] * Wc[q], X[p] * Y[q] - Y[p] * X[q]))
    det = X[0] * coef[0][0] + Y[0] * coef[0][1] + Wc[0] * coef[0][2]
    if det == 0:
        return False
    sg = 1 if det > 0 else -1
    for a, b, c in coef:
        a, b, c = sg * a, sg * b, sg * c
        E = a * px + b * py + c
        if not (E > 0 or (E == 0 and (a > 0 or (a == 0 and b > 0)))):
            return False
    return True


def test_coverage_matches_exact_rational_arithmetic(oracle):
    """The f64 edge functions decide coverage exactly as infinite-precision arithmetic does."""
    H, W = 24, 20
    verts, faces = scenes.rand_mesh(12, seed=2, r_lo=0.2, r_hi=0.6)
    for fi in range(len(faces)):
        got = oracle.visibility(verts, faces[fi:fi + 1], H, W)[0] >= 0
        # depth clip is not part of _exact_cover: rand_mesh keeps |z/w| <= 0.9, so nothing is clipped
        want = np.array([[_exact_cover(verts, faces[fi], H, W, i, r) for i in range(W)] for r in range(H)])
        assert np.array_equal(got, want), 'face %d' % fi


def test_exact_lattice_mesh_has_no_cracks_or_overlaps(oracle):
    """A grid whose vertices sit exactly on pixel centres puts hundreds of samples exactly on shared
    edges and vertices: every pixel must be drawn exactly once (the GL watertightness guarantee)."""
    n, H, W = 9, 64, 64
    xs = (np.arange(n) * 8 + 0.5) / W * 2 - 1
    ys = (np.arange(n) * 8 + 0.5) / H * 2 - 1
    gx, gy = np.meshgrid(xs, ys)
    v = np.stack([gx, gy, np.zeros_like(gx), np.ones_like(gx)], -1).reshape(-1, 4).astype(np.float32)
    idx = np.arange(n * n).reshape(n, n)
    a, b, c, d = idx[:-1, :-1].ravel(), idx[:-1, 1:].ravel(), idx[1:, 1:].ravel(), idx[1:, :-1].ravel()
    f = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 0).astype(np.int32)
    count = np.zeros((H, W), int)
    for fi in range(len(f)):
        count += oracle.visibility(v, f[fi:fi + 1], H, W)[0] >= 0
    inside = count[1:-1, 1:-1]   # pixel centres 0.5 .. 64.5-> columns 0..63; the mesh spans [0.5, 64.5)
    assert inside.max() == 1, 'a sample was claimed by two faces'
    fid = oracle.visibility(v, f, H, W)[0]
    rows, cols = np.nonzero(fid >= 0)
    assert (fid[rows.min():rows.max() + 1, cols.min():cols.max() + 1] >= 0).all(), 'crack inside the mesh'
    assert count.sum() == (fid >= 0).sum()


def test_perspective_correct_interpolation(oracle):
    """`smooth` varyings (csrc/shaders.cpp:22,35): b_k = (beta_k/w_k)/sum_j(beta_j/w_j) with beta the
    screen-space barycentrics; clip_w = 1/gl_FragCoord.w (csrc/shaders.cpp:74)."""
    ndc = np.array([[-0.8, -0.7], [0.9, -0.5], [0.1, 0.8]])
    w = np.array([1.0, 3.0, 0.5])
    v = np.concatenate([ndc * w[:, None], np.zeros((3, 1)), w[:, None]], 1).astype(np.float32)
    H, W = 50, 70
    fid, bary, cw = oracle.visibility(v, np.array([[0, 1, 2]], np.int32), H, W)
    vd = v.astype(np.float64)
    sx = (vd[:, 0] / vd[:, 3] + 1) * W / 2
    sy = (vd[:, 1] / vd[:, 3] + 1) * H / 2
    rows, cols = np.nonzero(fid >= 0)
    assert len(rows) > 500
    px, py = cols + 0.5, (H - 1 - rows) + 0.5
    def edge(ax, ay, bx, by):
        return (bx - ax) * (py - ay) - (by - ay) * (px - ax)
    area = (sx[1] - sx[0]) * (sy[2] - sy[0]) - (sy[1] - sy[0]) * (sx[2] - sx[0])
    beta = np.stack([edge(sx[1], sy[1], sx[2], sy[2]), edge(sx[2], sy[2], sx[0], sy[0]), edge(sx[0], sy[0], sx[1], sy[1])], 1) / area
    q = beta / vd[:, 3][None]
    want_b = q / q.sum(1, keepdims=True)
    want_w = 1.0 / q.sum(1)
    assert np.allclose(bary[rows, cols], want_b, rtol=0, atol=2e-6)
    assert np.allclose(cw[rows, cols], want_w, rtol=1e-5)
    # and the forward colours are that interpolation of the vertex colours over the background
    cols_v = np.random.default_rng(0).uniform(0, 1, (3, 3)).astype(np.float32)
    bgd = np.full((H, W, 3), 0.25, np.float32)
    out = oracle.forward(bgd[None], v[None], cols_v[None], np.array([[[0, 1, 2]]], np.int32))[0]
    assert np.allclose(out[rows, cols], want_b @ cols_v.astype(np.float64), atol=3e-6)
    assert np.all(out[fid < 0] == 0.25)


def test_forward_group_invariance_and_batch(oracle):
    """C=4 equals the concatenation of a 3-channel and a 1-channel render (dirt/rasterise_ops.py:86-108),
    and a batch equals the stack of its scenes (dirt/rasterise_ops.py:56-63)."""
    b = scenes.batch_scene(80, 40, 56, 4, seeds=[1, 2], r_lo=0.05, r_hi=0.3)
    full = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    a3 = oracle.forward(b['background'][..., :3], b['vertices'], b['vertex_colors'][..., :3], b['faces'])
    a1 = oracle.forward(b['background'][..., 3:], b['vertices'], b['vertex_colors'][..., 3:], b['faces'])
    assert np.array_equal(full, np.concatenate([a3, a1], -1))
    for i in range(2):
        one = oracle.forward(b['background'][i:i + 1], b['vertices'][i:i + 1], b['vertex_colors'][i:i + 1], b['faces'][i:i + 1])
        assert np.array_equal(full[i], one[0])


def test_degenerate_and_invalid_faces_are_skipped(oracle):
    s = scenes.rand_scene(30, 32, 32, 3, 5, 0.1, 0.4)
    base = _fwd(oracle, s)
    extra = np.array([[0, 0, 1], [0, 1, 10 ** 6], [-1, 0, 1]], np.int32)       # zero area, out of range, negative
    s2 = dict(s, faces=np.concatenate([s['faces'], extra], 0))
    assert np.array_equal(_fwd(oracle, s2), base)
    v = s['vertices'].copy()
    v[0, 0] = np.nan                                                           # face 0 uses vertex 0 (split mesh)
    s3 = dict(s, vertices=v)
    s4 = dict(s, faces=s['faces'][1:])
    assert np.array_equal(_fwd(oracle, s3), _fwd(oracle, s4))


# --------------------------------------------------------------------------------------------- backward

def _numpy_assemble_grads(verts, faces, fid, bary, cw, pixels, g, flat_group=None, iib=0, q1_intended=False):
    """Independent restatement of assemble_grads (csrc/rasterise_grad_egl.cu:93-236) for ONE scene and ONE channel
    group (pixels / g: [H,W,G], G = 3 or 1), vectorised numpy in float64, from the oracle's visibility arrays.

    G = 1 reproduces quirk Q1 unless `q1_intended`: the L1 norms of :185 take "channels" 0, 1, 2 of a 1-channel
    tensor, i.e. elements base, base + 1, base + 2 of the flattened [B,H,W,1] slice `flat_group` (this scene is number
    `iib` in it), where base is the tap's (edge-clamped) pixel; past the end of the slice: its last element."""
    H, W, G = pixels.shape
    V = verts.shape[0]
    pad = np.pad(pixels.astype(np.float64), ((1, 1), (1, 1), (0, 0)), mode='edge')
    at = lambda ox, oy: pad[1 - oy:H + 1 - oy, 1 + ox:W + 1 + ox]   # offset_y is UP in the image (GL y)
    scharr = lambda a: ((a(-1, -1) + a(-1, 1) - a(1, -1) - a(1, 1)) * (3 / 32) + (a(-1, 0) - a(1, 0)) * (10 / 32),
                        (a(-1, -1) + a(1, -1) - a(-1, 1) - a(1, 1)) * (3 / 32) + (a(0, -1) - a(0, 1)) * (10 / 32))
    sx, sy = scharr(at)
    if G == 1 and not q1_intended:
        flat = np.asarray(flat_group, np.float64).reshape(-1)
        rr_, cc_ = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')

        def at_alias(ch):
            def a(ox, oy):
                base = (iib * H + np.clip(rr_ - oy, 0, H - 1)) * W + np.clip(cc_ + ox, 0, W - 1)
                return flat[np.minimum(base + ch, flat.size - 1)][..., None]
            return a
        l1x = sum(np.abs(scharr(at_alias(ch))[0][..., 0]) for ch in range(3))
        l1y = sum(np.abs(scharr(at_alias(ch))[1][..., 0]) for ch in range(3))
    else:
        l1x, l1y = np.abs(sx).sum(-1), np.abs(sy).sum(-1)
    tri = np.where(fid[..., None] >= 0, faces[np.maximum(fid, 0)], -1)        # [H,W,3] vertex indices
    covered = fid >= 0
    gv = np.zeros((V, 4)); gvc = np.zeros((V, G)); gb = np.where(covered[..., None], 0, g).astype(np.float32)
    for k in range(3):
        np.add.at(gvc, tri[covered][:, k], (g[covered] * bary[covered][:, k:k + 1]).astype(np.float64))
    # dilation
    b2, t2, w2 = bary.copy(), tri.copy(), cw.copy()
    rr, cc = np.meshgrid(np.arange(H), np.arange(W), indexing='ij')
    interior = (cc > 0) & (rr > 0) & (cc < W - 1) & (rr < H - 1)
    horiz = l1x > l1y
    ox = np.where(horiz, 1, 0); oy = np.where(horiz, 0, 1)
    flip = ((cc + rr) % 2) == 1
    ox = np.where(flip, -ox, ox); oy = np.where(flip, -oy, oy)
    done = np.zeros((H, W), bool)
    for sign in (1, -1):
        nr = np.clip(rr - sign * oy, 0, H - 1); nc = np.clip(cc + sign * ox, 0, W - 1)
        n_tri, n_w, n_b = tri[nr, nc], cw[nr, nc], bary[nr, nc]
        ok = interior & ~done & (n_tri[..., 0] != -1) & np.any(n_tri != t2, -1) & (w2 > n_w)
        b2[ok], t2[ok], w2[ok] = n_b[ok], n_tri[ok], n_w[ok]
        done |= ok
    cov2 = t2[..., 0] != -1
    dLdx = (g * sx).sum(-1); dLdy = (g * sy).sum(-1)
    vx = verts[np.maximum(t2, 0)][..., 0]; vy = verts[np.maximum(t2, 0)][..., 1]
    clip_x = (b2 * vx).sum(-1); clip_y = (b2 * vy).sum(-1)
    with np.errstate(divide='ignore', invalid='ignore'):
        for k in range(3):
            gx = dLdx * b2[..., k] * (0.5 * W / w2)
            gy = dLdy * b2[..., k] * (0.5 * H / w2)
            gw = dLdx * b2[..., k] * (-0.5 * W * clip_x / w2 ** 2) + dLdy * b2[..., k] * (-0.5 * H * clip_y / w2 ** 2)
            np.add.at(gv[:, 0], t2[cov2][:, k], gx[cov2]); np.add.at(gv[:, 1], t2[cov2][:, k], gy[cov2])
            np.add.at(gv[:, 3], t2[cov2][:, k], gw[cov2])
    return gb, gv, gvc, done


def _numpy_backward_multichannel(oracle, b, px, q1_intended):
    """`_rasterise_grad_multichannel` (dirt/rasterise_ops.py:132-177) on top of the numpy restatement: groups of 3 while
    >= 3 channels remain, then singles (:148-152); grad_vertices summed over the groups (:163), the others concatenated."""
    B, H, W, C = px.shape
    V = b['vertices'].shape[1]
    gb = np.zeros_like(px); gv = np.zeros((B, V, 4)); gvc = np.zeros((B, V, C))
    vis = [oracle.visibility(b['vertices'][i], b['faces'][i], H, W) for i in range(B)]
    c0 = 0
    while c0 < C:
        G = 3 if c0 + 3 <= C else 1
        flat = np.ascontiguousarray(px[..., c0:c0 + G])          # what TF hands the op: the contiguous [B,H,W,G] slice
        for i in range(B):
            fid, bary, cw = vis[i]
            a, v_, c_, _ = _numpy_assemble_grads(b['vertices'][i], b['faces'][i], fid, bary, cw, px[i, ..., c0:c0 + G],
                                                 b['grad_pixels'][i, ..., c0:c0 + G], flat, i, q1_intended)
            gb[i, ..., c0:c0 + G] = a; gv[i] += v_; gvc[i, :, c0:c0 + G] = c_
        c0 += G
    return gb, gv, gvc


@pytest.mark.parametrize('C,seeds', [(1, [5]), (1, [6, 7]), (4, [8]), (5, [9, 10]), (7, [11])])
@pytest.mark.parametrize('q1_intended', [False, True])
def test_backward_groups_and_q1_match_numpy_restatement(oracle, C, seeds, q1_intended):
    """The 1-channel group in both Q1 modes (aliased "channels" crossing pixel, row and SCENE boundaries in a batch) and
    the multi-group sums of C = 4, 5, 7 against the independent numpy restatement -- the oracle is not compared with
    itself here."""
    b = scenes.batch_scene(120, 30, 44, C, seeds, r_lo=0.05, r_hi=0.3)
    px = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    out = oracle.backward(b['vertices'], b['faces'], px, b['grad_pixels'], flags=1 if q1_intended else 0)
    gb, gv, gvc = _numpy_backward_multichannel(oracle, b, px, q1_intended)
    assert np.array_equal(out['grad_background'], gb)
    assert np.allclose(out['grad_vertex_colors'], gvc, rtol=1e-5, atol=1e-5)
    assert np.allclose(out['grad_vertices'], gv, rtol=1e-4, atol=1e-5 * np.abs(gv).max())


def test_cylinder_translation_gradient_sanity_band(oracle):
    """What the reference's tests/rasterise_tests.py:108-116 shows as images: the gradient of an image functional with
    respect to the cylinder's translation.  dL/dvertices is a filter-based approximation (README.md:197), so central
    finite differences over a couple of pixels are a sanity band: same sign, magnitude within a factor of two."""
    h, w = 36, 48
    ramp = np.linspace(-1, 1, w, dtype=np.float32)[None, :, None] * np.ones((h, 1, 3), np.float32)
    bump = np.linspace(-1, 1, h, dtype=np.float32)[:, None, None] * np.ones((1, w, 3), np.float32)
    P = scenes._perspective(0.1, 20., 0.2, float(h) / w)
    for axis, g in ((0, ramp), (1, bump)):
        def loss(t):
            s = scenes.cylinder_scene(translation=t, bgcolor=(0.1, 0.1, 0.1), vertex_color=(0.9, 0.8, 0.7))
            s['background'][:] = 0.1
            s['vertex_colors'][:] = 0.9
            px = oracle.forward(s['background'][None], s['vertices'][None], s['vertex_colors'][None], s['faces'][None])
            return s, px, float((px[0].astype(np.float64) * g).sum())
        t0 = [0., 0., -0.25]
        s, px, _ = loss(t0)
        out = oracle.backward(s['vertices'][None], s['faces'][None], px, g[None])
        analytic = float((out['grad_vertices'][0].astype(np.float64) @ P[axis, :]).sum())   # d clip / d t_axis = row `axis` of P
        step = 0.02                                                                      # ~2 pixels at this depth
        tp, tm = list(t0), list(t0)
        tp[axis] += step; tm[axis] -= step
        fd = (loss(tp)[2] - loss(tm)[2]) / (2 * step)
        assert fd * analytic > 0, (axis, fd, analytic)
        assert 0.5 <= abs(analytic / fd) <= 2.0, (axis, fd, analytic)


@pytest.mark.parametrize('seed,shared', [(3, False), (4, True)])
def test_backward_matches_numpy_restatement(oracle, seed, shared):
    s = scenes.rand_scene(150, 48, 36, 3, seed, 0.05, 0.3, shared)   # 48x36x3 as tests/rasterise_tests.py:55-56
    px = _fwd(oracle, s)
    out = oracle.backward(s['vertices'][None], s['faces'][None], px[None], s['grad_pixels'][None], want_debug=True)
    fid, bary, cw = oracle.visibility(s['vertices'], s['faces'], 36, 48) if False else oracle.visibility(s['vertices'], s['faces'], s['height'], s['width'])
    gb, gv, gvc, dil = _numpy_assemble_grads(s['vertices'], s['faces'], fid, bary, cw, px, s['grad_pixels'])
    assert np.array_equal(out['grad_background'][0], gb)
    assert np.allclose(out['grad_vertex_colors'][0], gvc, rtol=1e-5, atol=1e-5)
    scale = np.abs(gv).max()
    assert np.allclose(out['grad_vertices'][0], gv, rtol=1e-4, atol=1e-5 * scale)
    assert np.array_equal(out['debug_thingy'][0][..., 0] > 0, dil)
    assert dil.sum() > 20, 'the scene must exercise dilation'


@pytest.mark.parametrize('C', [1, 3, 4, 5])
def test_backward_invariants(oracle, C):
    """Facts that follow from the CUDA source."""
    s = scenes.rand_scene(120, 40, 52, C, 10 + C, 0.05, 0.3)
    px = _fwd(oracle, s)
    out = oracle.backward(s['vertices'][None], s['faces'][None], px[None], s['grad_pixels'][None])
    fid, _, _ = oracle.visibility(s['vertices'], s['faces'], 40, 52)
    g = s['grad_pixels']
    assert np.all(out['grad_vertices'][0][:, 2] == 0)                                   # .z never written, :228-230
    assert np.array_equal(out['grad_background'][0], np.where((fid < 0)[..., None], g, 0))  # :143-147 + memset :247
    want = g[fid >= 0].astype(np.float64).sum(0)                                        # sum_k b_k = 1
    assert np.allclose(out['grad_vertex_colors'][0].astype(np.float64).sum(0), want, rtol=1e-4, atol=1e-3)


def test_colour_and_background_gradients_are_exact_derivatives(oracle):
    """pixels is linear in vertex_colors and background, so directional finite differences of
    L = sum(g * pixels) must equal the analytic gradients (no filter approximation involved)."""
    s = scenes.rand_scene(100, 40, 40, 3, 21, 0.05, 0.3)
    px = _fwd(oracle, s)
    out = oracle.backward(s['vertices'][None], s['faces'][None], px[None], s['grad_pixels'][None])
    rng = np.random.default_rng(0)
    g = s['grad_pixels'].astype(np.float64)
    for name, key in (('vertex_colors', 'grad_vertex_colors'), ('background', 'grad_background')):
        d = rng.standard_normal(s[name].shape).astype(np.float32)
        plus = _fwd(oracle, dict(s, **{name: s[name] + d}))
        fd = ((plus.astype(np.float64) - px) * g).sum()
        an = (out[key][0].astype(np.float64) * d).sum()
        assert abs(fd - an) <= 1e-3 * max(1.0, abs(an)), (name, fd, an)


def test_vertex_gradient_points_the_right_way(oracle):
    """dL/dvertices is a filter-based approximation (README.md:197), so finite differences are a sanity
    band, not an equality: moving a bright square right over a dark background must increase
    L = sum(pixels * x-ramp), and grad_vertices.x must be positive with the right magnitude."""
    H = W = 64
    s = 0.4
    v = np.array([[-s, -s, 0, 1], [s, -s, 0, 1], [s, s, 0, 1], [-s, s, 0, 1]], np.float32)
    f = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    cols = np.ones((4, 1), np.float32)
    bg = np.zeros((H, W, 1), np.float32)
    ramp = (np.arange(W, dtype=np.float32)[None, :, None] * np.ones((H, 1, 1), np.float32)) / W
    px = oracle.forward(bg[None], v[None], cols[None], f[None])
    out = oracle.backward(v[None], f[None], px, ramp[None])
    gx = out['grad_vertices'][0][:, 0].sum()
    # analytic: d/dt sum(ramp*pixels) for a translation t in clip units = (W/2 px per unit) * height_px * (ramp jump)
    side_px = 2 * s * W / 2
    expect = (W / 2) * side_px * (side_px / W)
    assert gx > 0 and 0.5 * expect < gx < 1.5 * expect, (gx, expect)
    assert abs(out['grad_vertices'][0][:, 1].sum()) < 0.05 * gx


def test_multichannel_grouping(oracle):
    """_rasterise_grad_multichannel (dirt/rasterise_ops.py:145-165): groups [0:3],[3:4]; grad_vertices
    summed, the rest concatenated.  Batch of 2 so the 1-channel group's flat aliasing (Q1) crosses scenes."""
    b = scenes.batch_scene(90, 24, 40, 4, seeds=[5, 6], r_lo=0.05, r_hi=0.3)
    px = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    for flags in (0, oracle.FLAG_Q1_INTENDED):
        full = oracle.backward(b['vertices'], b['faces'], px, b['grad_pixels'], flags=flags)
        g3 = oracle.backward(b['vertices'], b['faces'], px[..., :3], b['grad_pixels'][..., :3], flags=flags)
        g1 = oracle.backward(b['vertices'], b['faces'], px[..., 3:], b['grad_pixels'][..., 3:], flags=flags)
        assert np.allclose(full['grad_vertices'], g3['grad_vertices'].astype(np.float64) + g1['grad_vertices'], rtol=1e-5, atol=1e-4)
        assert np.array_equal(full['grad_vertex_colors'], np.concatenate([g3['grad_vertex_colors'], g1['grad_vertex_colors']], -1))
        assert np.array_equal(full['grad_background'], np.concatenate([g3['grad_background'], g1['grad_background']], -1))


def test_q1_quirk_changes_only_single_channel_groups(oracle):
    s = scenes.rand_scene(200, 48, 48, 3, 8, 0.05, 0.3)
    px = _fwd(oracle, s)
    a = oracle.backward(s['vertices'][None], s['faces'][None], px[None], s['grad_pixels'][None], flags=0)
    b = oracle.backward(s['vertices'][None], s['faces'][None], px[None], s['grad_pixels'][None], flags=oracle.FLAG_Q1_INTENDED)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    s1 = scenes.rand_scene(200, 48, 48, 1, 8, 0.05, 0.3)
    px1 = _fwd(oracle, s1)
    a = oracle.backward(s1['vertices'][None], s1['faces'][None], px1[None], s1['grad_pixels'][None], flags=0)
    b = oracle.backward(s1['vertices'][None], s1['faces'][None], px1[None], s1['grad_pixels'][None], flags=oracle.FLAG_Q1_INTENDED)
    assert not np.array_equal(a['grad_vertices'], b['grad_vertices'])       # the L1 of aliased channels picks other directions
    assert np.array_equal(a['grad_vertex_colors'], b['grad_vertex_colors'])


def test_too_many_vertices_is_rejected(oracle):
    """csrc/rasterise_grad_egl.cpp:399-405."""
    import ctypes
    lib = oracle.oracle._load()
    rc = lib.dirt_oracle_backward(None, None, None, None, None, None, None, None, 1, (1 << 24) + 1, 1, 4, 4, 1, 0)
    assert rc == -2


# ----------------------------------------------------------------------------------------------- golden

def test_golden_fixtures(oracle):
    """tests/golden/*.npz were written by tests/golden/make_golden.py from this oracle; the inputs are
    regenerated from their seeds.  Any drift of the oracle's arithmetic shows up here."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(GOLDEN, '*.npz')) if os.path.basename(f) not in ('ref_grads.npz', 'helpers_ref.npz'))
    assert files, 'no golden fixtures committed'
    from tests.golden.make_golden import CASES, make_inputs
    for path in files:
        name = os.path.splitext(os.path.basename(path))[0]
        z = np.load(path)
        s = make_inputs(CASES[name])
        px = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
        assert np.array_equal(px.view(np.uint32), z['pixels'].view(np.uint32)), name
        out = oracle.backward(s['vertices'], s['faces'], px, s['grad_pixels'])
        assert np.array_equal(out['grad_background'], z['grad_background']), name
        assert np.allclose(out['grad_vertices'], z['grad_vertices'], rtol=1e-6, atol=1e-6), name
        assert np.allclose(out['grad_vertex_colors'], z['grad_vertex_colors'], rtol=1e-6, atol=1e-6), name
