"""dirt_amd.matrices / lighting / projection against numpy restatements of the reference formulas
(dirt/matrices.py, dirt/lighting.py, dirt/projection.py) and against the properties that define them.
Pure host-side dense math: runs on the CPU."""
import math

import numpy as np
import pytest
import torch

from dirt_amd import lighting, matrices, projection
from tests import scenes


def _np_rodrigues(v):
    v = np.asarray(v, np.float64) + 1e-12
    n = np.linalg.norm(v)
    k = v / n
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return math.cos(n) * np.eye(3) + (1 - math.cos(n)) * np.outer(k, k) + math.sin(n) * K


def test_rodrigues_matches_formula_and_is_a_rotation():
    rng = np.random.default_rng(0)
    vs = rng.normal(size=(5, 7, 3))
    r = matrices.rodrigues(torch.from_numpy(vs), three_by_three=True).numpy()
    assert r.shape == (5, 7, 3, 3)
    for i in range(5):
        for j in range(7):
            np.testing.assert_allclose(r[i, j], _np_rodrigues(vs[i, j]), atol=1e-12)
            np.testing.assert_allclose(r[i, j] @ r[i, j].T, np.eye(3), atol=1e-12)
            assert abs(np.linalg.det(r[i, j]) - 1) < 1e-12
    r4 = matrices.rodrigues([0., 0., math.pi / 2]).numpy()
    assert r4.shape == (4, 4) and r4[3, 3] == 1 and np.all(r4[3, :3] == 0) and np.all(r4[:3, 3] == 0)
    # row vectors: x axis -> -y under a +90 degree rotation about z in this (OpenCV-docs) convention
    np.testing.assert_allclose(np.array([1., 0, 0, 1]) @ r4, [0, -1, 0, 1], atol=1e-6)
    # derivative exists at exactly zero (the reason for the reference's 1e-12, dirt/matrices.py:38)
    z = torch.zeros(3, requires_grad=True)
    matrices.rodrigues(z).sum().backward()
    assert torch.isfinite(z.grad).all()


def test_translation_scale_pad_compose():
    t = matrices.translation([[1., 2., 3.], [4., 5., 6.]]).numpy()
    assert t.shape == (2, 4, 4)
    np.testing.assert_array_equal(np.array([0., 0, 0, 1]) @ t[1], [4, 5, 6, 1])
    s = matrices.scale([2., 3., 4.]).numpy()
    np.testing.assert_array_equal(s, np.diag([2., 3, 4, 1]))
    p = matrices.pad_3x3_to_4x4(torch.arange(9.).reshape(3, 3)).numpy()
    np.testing.assert_array_equal(p[:3, :3], np.arange(9.).reshape(3, 3))
    np.testing.assert_array_equal(p[3], [0, 0, 0, 1])
    a, b = matrices.translation([1., 0., 0.]), matrices.scale([2., 2., 2.])
    # compose(A, B) is "A then B" for row vectors (dirt/matrices.py:196-207)
    np.testing.assert_array_equal((torch.tensor([0., 0, 0, 1]) @ matrices.compose(a, b)).numpy(), [2, 0, 0, 1])
    np.testing.assert_array_equal((torch.tensor([0., 0, 0, 1]) @ matrices.compose(b, a)).numpy(), [1, 0, 0, 1])
    np.testing.assert_array_equal(matrices.compose().numpy(), np.eye(4))
    assert matrices.compose(a) is a


def test_perspective_projection_maps_the_frustum_to_the_clip_cube():
    near, far, right, aspect = 0.1, 20., 0.2, 0.75
    m = matrices.perspective_projection(near, far, right, aspect).double().numpy()
    top = right * aspect
    for (x, y, z), want in [((right, top, -near), (1, 1, -1)), ((-right, -top, -near), (-1, -1, -1)),
                            ((0, 0, -far), (0, 0, 1)), ((right * far / near, 0, -far), (1, 0, 1))]:
        c = np.array([x, y, z, 1.]) @ m
        np.testing.assert_allclose(c[:3] / c[3], want, atol=1e-5)
        assert c[3] == pytest.approx(-z)
    b = matrices.perspective_projection(torch.tensor([0.1, 0.2]), 20., torch.tensor([[0.1], [0.3]]), 1.0)
    assert b.shape == (2, 2, 4, 4)
    assert b[1, 0, 0, 0] == pytest.approx(0.1 / 0.3)


def _sphere(n=12):
    th, ph = np.meshgrid(np.linspace(0.2, math.pi - 0.2, n), np.linspace(0, 2 * math.pi, 2 * n, endpoint=False), indexing='ij')
    v = np.stack([np.sin(th) * np.cos(ph), np.sin(th) * np.sin(ph), np.cos(th)], -1).reshape(-1, 3)
    idx = np.arange(n * 2 * n).reshape(n, 2 * n)
    f = []
    for i in range(n - 1):
        for j in range(2 * n):
            a, b, c, d = idx[i, j], idx[i, (j + 1) % (2 * n)], idx[i + 1, (j + 1) % (2 * n)], idx[i + 1, j]
            f += [[a, d, c], [a, c, b]]
    return v.astype(np.float32), np.array(f, np.int32)


def test_vertex_normals_match_a_dense_restatement_and_point_outwards():
    v, f = _sphere()
    n = lighting.vertex_normals(torch.from_numpy(v), torch.from_numpy(f)).numpy()
    # dense restatement of dirt/lighting.py:21-28,79-81
    tri = v[f].astype(np.float64)
    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    fn /= (np.linalg.norm(fn, axis=-1, keepdims=True) + 1e-12)
    want = np.zeros_like(v, dtype=np.float64)
    for k in range(3):
        np.add.at(want, f[:, k], fn)
    want /= (np.linalg.norm(want, axis=-1, keepdims=True) + 1e-12)
    np.testing.assert_allclose(n, want, atol=1e-5)
    assert np.all(np.sum(n * v, -1) > 0.9)  # a sphere's normals are its positions
    # batched, homogeneous input: w is dropped, batch entries are independent
    vb = torch.from_numpy(np.stack([np.concatenate([v, np.ones([len(v), 1], np.float32)], 1), np.concatenate([2 * v, np.ones([len(v), 1], np.float32)], 1)]))
    nb = lighting.vertex_normals(vb, torch.from_numpy(f).long()).numpy()
    assert nb.shape == (2, len(v), 3)
    np.testing.assert_allclose(nb[0], n, atol=1e-6)
    np.testing.assert_allclose(nb[1], n, atol=1e-5)


def test_split_vertices_and_pre_split_normals():
    v, f = _sphere(6)
    vs, fs = lighting.split_vertices_by_face(torch.from_numpy(v), torch.from_numpy(f))
    assert vs.shape == (3 * len(f), 3) and fs.dtype == torch.int32
    np.testing.assert_array_equal(fs.numpy(), np.arange(3 * len(f)).reshape(-1, 3))
    np.testing.assert_array_equal(vs.numpy().reshape(-1, 3, 3), v[f])
    n = lighting.vertex_normals_pre_split(vs, fs).numpy().reshape(-1, 3, 3)
    tri = v[f].astype(np.float64)
    fn = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    fn /= (np.linalg.norm(fn, axis=-1, keepdims=True) + 1e-12)
    for k in range(3):
        np.testing.assert_allclose(n[:, k], fn, atol=1e-5)
    vb, _ = lighting.split_vertices_by_face(torch.from_numpy(np.stack([v, 2 * v])), f)
    assert vb.shape == (2, 3 * len(f), 3)


def test_reflectance_models():
    rng = np.random.default_rng(3)
    n = rng.normal(size=(2, 9, 3)); n /= np.linalg.norm(n, axis=-1, keepdims=True)
    col = rng.uniform(size=(2, 9, 4))
    ld = rng.normal(size=(2, 3)); ld /= np.linalg.norm(ld, axis=-1, keepdims=True)
    lc = rng.uniform(size=(2, 4))
    pos = rng.normal(size=(2, 9, 3))
    T = torch.from_numpy
    cos = np.einsum('bvi,bi->bv', n, -ld)
    got = lighting.diffuse_directional(T(n), T(col), T(ld), T(lc)).numpy()
    np.testing.assert_allclose(got, lc[:, None, :] * col * np.abs(cos)[..., None], atol=1e-12)
    got = lighting.diffuse_directional(T(n), T(col), T(ld), T(lc), double_sided=False).numpy()
    np.testing.assert_allclose(got, lc[:, None, :] * col * np.maximum(cos, 0)[..., None], atol=1e-12)
    lp = rng.normal(size=(2, 3))
    rel = pos - lp[:, None, :]
    inc = rel / (np.linalg.norm(rel, axis=-1, keepdims=True) + 1e-12)
    cosp = np.sum(n * inc, -1)
    got = lighting.diffuse_point(T(pos), T(n), T(col), T(lp), T(lc), double_sided=False).numpy()
    np.testing.assert_allclose(got, lc[:, None, :] * col * np.maximum(cosp, 0)[..., None], atol=1e-12)
    cam, sh = rng.normal(size=(2, 3)), rng.uniform(1, 8, size=(2,))
    refl = ld[:, None, :] + 2 * np.einsum('bvi,bi->bv', n, -ld)[..., None] * n
    tc = cam[:, None, :] - pos
    coss = np.sum((tc / np.linalg.norm(tc, axis=-1, keepdims=True) + 1e-12) * refl, -1, keepdims=True)
    got = lighting.specular_directional(T(pos), T(n), T(col), T(ld), T(lc), T(cam), T(sh)).numpy()
    np.testing.assert_allclose(got, lc[:, None, :] * col * np.abs(coss) ** sh[:, None, None], atol=1e-10)
    # python lists are accepted like tensors (samples/simple.py:62-65)
    got = lighting.diffuse_directional(T(n[0]).float(), torch.ones(9, 3), [1., 0., 0.], [1., 1., 1.]).numpy()
    np.testing.assert_allclose(got, np.abs(n[0][:, :1]) * np.ones((1, 3)), atol=1e-6)


def test_unproject_pixels_to_rays_inverts_the_projection():
    w, h = 64, 48
    view = matrices.compose(matrices.translation([0.3, -0.2, -3.0]), matrices.rodrigues([0.1, 0.2, 0.0]))
    proj = matrices.perspective_projection(0.5, 10., 0.25, h / w)
    world_to_clip = torch.matmul(view, proj).double()
    clip_to_world = torch.linalg.inv(world_to_clip).float()
    xs, ys = np.meshgrid(np.arange(0, w, 7) + 0.5, np.arange(0, h, 5) + 0.5)
    pix = torch.from_numpy(np.stack([xs, ys], -1).astype(np.float32))
    starts, deltas = projection.unproject_pixels_to_rays(pix, clip_to_world, torch.tensor([w, h]))
    assert starts.shape == pix.shape[:-1] + (3,) and deltas.shape == starts.shape
    for t in (0.0, 0.7, 3.0):  # every point of a ray projects back onto its pixel
        p = (starts + t * deltas).double()
        c = torch.cat([p, torch.ones_like(p[..., :1])], -1) @ world_to_clip
        ndc = c[..., :3] / c[..., 3:]
        px = (ndc[..., 0] + 1) * w / 2
        py = (1 - ndc[..., 1]) * h / 2
        np.testing.assert_allclose(px.numpy(), xs, atol=2e-3)
        np.testing.assert_allclose(py.numpy(), ys, atol=2e-3)
    c = torch.cat([starts.double(), torch.ones_like(starts[..., :1]).double()], -1) @ world_to_clip
    np.testing.assert_allclose((c[..., 2] / c[..., 3]).numpy(), -1, atol=1e-3)  # starts lie on the near plane
    # batched: A = [2], B = [5]
    pb = torch.rand(2, 5, 2) * 40
    sb, db = projection.unproject_pixels_to_rays(pb, clip_to_world.expand(2, 4, 4), torch.tensor([[w, h], [w, h]]))
    s1, d1 = projection.unproject_pixels_to_rays(pb[1], clip_to_world, torch.tensor([w, h]))
    np.testing.assert_allclose(sb[1].numpy(), s1.numpy(), atol=1e-6)
    np.testing.assert_allclose(db[1].numpy(), d1.numpy(), atol=1e-6)


def test_simple_sample_pipeline_reproduces_the_baked_cube_scene():
    """samples/simple.py:34-66 through dirt_amd.matrices / lighting gives the clip-space cube that
    scenes.cube_scene (K2) bakes with numpy."""
    H, W = 256, 256
    verts = torch.tensor([[x, y, z] for z in [-1, 1] for y in [-1, 1] for x in [-1, 1]], dtype=torch.float32)
    quads = [[0, 1, 3, 2], [4, 5, 7, 6], [1, 5, 4, 0], [2, 6, 7, 3], [4, 6, 2, 0], [3, 7, 5, 1]]
    tris = sum([[[a, b, c], [c, d, a]] for a, b, c, d in quads], [])
    v_obj, faces = lighting.split_vertices_by_face(verts, torch.tensor(tris, dtype=torch.int32))
    colors = torch.ones_like(v_obj)
    v_obj = torch.cat([v_obj, torch.ones_like(v_obj[:, -1:])], dim=1)
    v_world = v_obj @ matrices.rodrigues([0., 0.5, 0.])
    normals = lighting.vertex_normals_pre_split(v_world, faces)
    view = matrices.compose(matrices.translation([0., -1.5, -3.5]), matrices.rodrigues([-0.3, 0., 0.]))
    v_clip = (v_world @ view) @ matrices.perspective_projection(near=0.1, far=20., right=0.1, aspect=float(H) / W)
    lit = lighting.diffuse_directional(normals, colors, light_direction=[1., 0., 0.], light_color=[1., 1., 1.]) * 0.8 + colors * 0.2
    s = scenes.cube_scene(H, W)
    np.testing.assert_array_equal(faces.numpy(), s['faces'])
    np.testing.assert_allclose(v_clip.numpy(), s['vertices'], atol=2e-5)
    np.testing.assert_allclose(lit.numpy(), s['vertex_colors'], atol=2e-6)


def test_texture_lookup_matches_a_restatement():
    """samples/textured.py:16-60: uv -> fractional (row, column) indices, nearest and bilinear look-ups."""
    from dirt_amd import texture as tex
    rng = np.random.default_rng(9)
    t = rng.uniform(size=(7, 5, 3))
    uv = rng.uniform(-1.5, 2.5, size=(4, 6, 2))
    idx = tex.uvs_to_pixel_indices(torch.from_numpy(uv), (7, 5)).numpy()
    np.testing.assert_allclose(idx, (uv[..., ::-1] % 1.0) * [7, 5], atol=1e-12)
    idc = tex.uvs_to_pixel_indices(torch.from_numpy(uv), (7, 5), mode='clamp').numpy()
    np.testing.assert_allclose(idc, np.clip(uv[..., ::-1], 0, 1) * [7, 5], atol=1e-12)
    near = tex.sample_texture(torch.from_numpy(t), torch.from_numpy(idx), mode='nearest').numpy()
    ii = np.minimum(idx.astype(np.int64), [6, 4])
    np.testing.assert_array_equal(near, t[ii[..., 0], ii[..., 1]])
    bil = tex.sample_texture(torch.from_numpy(t), torch.from_numpy(idx)).numpy()
    fl = np.floor(idx).astype(np.int64)
    fr = idx - fl
    g = lambda r, c: t[np.clip(r, 0, 6), np.clip(c, 0, 4)]
    want = (g(fl[..., 0], fl[..., 1]) * (1 - fr[..., 1:]) * (1 - fr[..., :1]) + g(fl[..., 0], fl[..., 1] + 1) * fr[..., 1:] * (1 - fr[..., :1])
            + g(fl[..., 0] + 1, fl[..., 1]) * (1 - fr[..., 1:]) * fr[..., :1] + g(fl[..., 0] + 1, fl[..., 1] + 1) * fr[..., 1:] * fr[..., :1])
    np.testing.assert_allclose(bil, want, atol=1e-12)
    # exact texel centres reproduce the texel; gradients reach the texture and the coordinates
    np.testing.assert_allclose(tex.sample_texture(torch.from_numpy(t), torch.tensor([[2., 3.]], dtype=torch.float64)).numpy()[0], t[2, 3])
    tt = torch.from_numpy(t).requires_grad_(True)
    uu = torch.from_numpy(uv).requires_grad_(True)
    tex.sample_texture(tt, tex.uvs_to_pixel_indices(uu, (7, 5))).sum().backward()
    assert float(tt.grad.sum()) == pytest.approx(uv.shape[0] * uv.shape[1] * 3) and torch.isfinite(uu.grad).all()


def test_unlisted_shader_parameters_are_detected():
    """A shader that closes over a parameter it does not list gets a warning, not a silently missing gradient (the
    reference's TensorFlow custom_gradient receives such `variables` automatically, dirt/rasterise_ops.py:239-246)."""
    import torch
    from dirt_amd.rasterise_ops import _unlisted_leaves
    g = torch.zeros(4, 4, 3, requires_grad=True)
    listed_w = torch.ones(3, requires_grad=True)
    stray_w = torch.full((3,), 2.0, requires_grad=True)
    constant = torch.ones(3)
    out = (g * listed_w).sum(-1) + (g * stray_w * constant).sum(-1)
    found = _unlisted_leaves(out, [g, listed_w])
    assert len(found) == 1 and found[0] is stray_w
    assert _unlisted_leaves(out, [g, listed_w, stray_w]) == []
    assert _unlisted_leaves(torch.ones(3), [g]) == []          # no graph at all


def test_shader_walk_tells_closures_of_one_factory_apart():
    """The one-time graph walk of rasterise_deferred is keyed on the shader's code AND on what it closes over: two
    closures of one factory (same code object, different captured parameters) must each be walked; the same closure
    again must not.  The captured objects are held by weak reference -- the id of a freed tensor can be reused by another
    object, a dead reference cannot match -- and a lambda re-created per step over fresh tensors is walked only the
    first few times (the walk is a debugging aid with a host-side cost)."""
    import gc
    from dirt_amd import rasterise_ops as ops
    ops._checked_shaders.clear()

    def factory(w):
        return lambda g: g * w

    wa, wb = torch.ones(3, requires_grad=True), torch.ones(3, requires_grad=True)
    fa, fb = factory(wa), factory(wb)
    assert fa.__code__ is fb.__code__
    assert ops._shader_needs_walk(fa) and ops._shader_needs_walk(fb)
    assert not ops._shader_needs_walk(fa) and not ops._shader_needs_walk(factory(wa)) and not ops._shader_needs_walk(fb)

    class Shader:
        def __init__(self, w):
            self.w = w

        def __call__(self, g):
            return g * self.w

    sa, sb = Shader(wa), Shader(wb)
    assert ops._shader_needs_walk(sa) and ops._shader_needs_walk(sb) and not ops._shader_needs_walk(sa)

    # a captured tensor that died: whatever object is created next (possibly at the same address) is a NEW closure
    ops._checked_shaders.clear()
    w = torch.ones(3, requires_grad=True)
    assert ops._shader_needs_walk(factory(w))
    del w
    gc.collect()
    w2 = torch.ones(3, requires_grad=True)
    assert ops._shader_needs_walk(factory(w2))
    # the training-loop pattern: a fresh lambda over fresh tensors every step is walked at most _WALKS_PER_CODE times
    ops._checked_shaders.clear()
    keep = [torch.ones(2, requires_grad=True) for _ in range(10)]
    import warnings
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter('always')
        ops._walk_limit_warned.clear()
        assert sum(ops._shader_needs_walk(factory(t)) for t in keep) == ops._WALKS_PER_CODE
    assert sum('further closures' in str(w.message) for w in caught) == 1   # said once that the check is off for this code object
    ops._checked_shaders.clear()


def test_shader_walk_compares_closures_by_identity_only():
    """A shader that closes over a LIST (tuple, dict) of tensors rebuilt every step: the cache must tell the closures apart
    by the identity of what they hold -- never by `==`, which on lists of multi-element tensors raises 'Boolean value of
    Tensor with more than one value is ambiguous' and on one-element tensors synchronises the device -- and must not keep the
    containers (hence the tensors) alive."""
    import gc
    import weakref
    from dirt_amd import rasterise_ops as ops
    ops._checked_shaders.clear()

    def factory(ws):
        return lambda g: g * ws[0]

    a, b = torch.ones(5, requires_grad=True), torch.ones(5, requires_grad=True)
    assert ops._shader_needs_walk(factory([a, b]))
    assert not ops._shader_needs_walk(factory([a, b]))          # another list object over the same tensors: the same closure
    assert ops._shader_needs_walk(factory([a, torch.ones(5)]))   # ... over another tensor: a new one (and no exception)
    assert ops._shader_needs_walk(factory((a, b)))               # a tuple is not a list
    # nothing a shader closes over is kept alive by the cache
    t = torch.ones(7, requires_grad=True)
    r = weakref.ref(t)
    ops._checked_shaders.clear()
    assert ops._shader_needs_walk(factory([t]))
    del t
    gc.collect()
    assert r() is None
    ops._checked_shaders.clear()
