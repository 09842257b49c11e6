"""Ports of the reference's gradient scripts as checked tests (GPU).

tests/rasterise_tests.py:50-132 builds the bevelled cylinder, pushes it through `dirt.lighting.split_vertices_by_face`,
a rotation / translation / `dirt.matrices.perspective_projection` chain and `dirt.rasterise(_batch)`, then asks
`tf.gradients` for the derivative of the image with respect to [translation, rotation_xy, bgcolor, vertex_color], one
backward pass per one-hot `d_loss_by_pixels`, and only LOOKS at the result (cv2.imshow).  Here the same graph is built
with `import dirt` over torch tensors, and every one of those four gradients is compared numerically with

    oracle.backward(clip, faces, pixels, d_loss_by_pixels)  contracted with  d(clip)/d(parameter)

the second factor computed independently in float64 on the CPU (torch.autograd.functional.jacobian of the same
matrix chain).  Tolerance: the oracle's per-element bound (tests/parity.py) pushed through the same contraction.
"""
import math

import numpy as np
import pytest
import torch

import dirt
import dirt.lighting
import dirt.matrices
import dirt.rasterise_ops
from tests import scenes
from tests import parity

pytestmark = pytest.mark.gpu

W, H = 48, 36


def _geometry(device, dtype=torch.float32):
    vertices, faces = scenes.make_cylinder(0.2, 0.75, 0.1, 0., 10)                      # tests/rasterise_tests.py:52
    vertices = np.concatenate([vertices, np.ones([len(vertices), 1])], axis=1)
    vertices, faces = dirt.lighting.split_vertices_by_face(torch.tensor(vertices, dtype=dtype, device=device),
                                                           torch.tensor(faces, dtype=torch.int32, device=device))  # :77-79
    return vertices, faces


def _project(vertices, translation, rotation_xy):
    """vertices @ view_matrix_1 @ view_matrix_2 @ projection (tests/rasterise_tests.py:58-84).  view_matrix_1 is a
    rotation about z scaled by 0.5 = rodrigues([0, 0, rotation_xy]) then scale(0.5); view_matrix_2 is
    `dirt.matrices.translation`."""
    zero = torch.zeros_like(rotation_xy)
    view_matrix_1 = dirt.matrices.compose(dirt.matrices.rodrigues(torch.stack([zero, zero, rotation_xy])),
                                          dirt.matrices.scale(torch.full([3], 0.5, dtype=vertices.dtype, device=vertices.device)))
    view_matrix_2 = dirt.matrices.translation(translation)
    projection_matrix = dirt.matrices.perspective_projection(0.1, 20., 0.2, float(H) / W).to(vertices)
    return vertices @ view_matrix_1 @ view_matrix_2 @ projection_matrix


def _clip_jacobians(translation, rotation_xy):
    """d(clip)/d(translation) [V,4,3] and d(clip)/d(rotation_xy) [V,4], float64 on the CPU."""
    vertices, _ = _geometry('cpu', torch.float64)
    t = torch.tensor(translation, dtype=torch.float64)
    r = torch.tensor(rotation_xy, dtype=torch.float64)
    jt, jr = torch.autograd.functional.jacobian(lambda a, b: _project(vertices, a, b), (t, r))
    return jt.numpy(), jr.numpy()


def _contract(ow, index, jac):
    """sum_v,k grad_vertices[v,k] * jac[v,k,...] with the per-element tolerance carried along."""
    g = ow['grad_vertices'][index].astype(np.float64)
    m = parity.GRAD_TOL * ow['mass_vertices'][index].astype(np.float64) + parity.COND_ULPS * ow['cond_vertices'][index].astype(np.float64)
    extra = jac.ndim - 2
    gg, mm = g.reshape(g.shape + (1,) * extra), m.reshape(m.shape + (1,) * extra)
    return (gg * jac).sum((0, 1)), (mm * np.abs(jac)).sum((0, 1))


def _assert_within(got, want, bound, what):
    got = np.asarray(got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got, np.float64)
    # float32 rounding of the chain itself (matmuls of the backward pass in torch) on top of the op's bound
    slack = 1e-5 * np.maximum(np.abs(want), bound / parity.GRAD_TOL) + 1e-12
    assert np.all(np.abs(got - want) <= bound + slack), '%s: got %s want %s (bound %s)' % (what, got, want, bound)


def _indicators(rng, covered, n):
    """One-hot d_loss_by_pixels as the reference's loops feed (tests/rasterise_tests.py:108-116), on a sample of pixels:
    silhouette, interior, background, frame border."""
    edge = covered & ~(np.roll(covered, 1, 0) & np.roll(covered, -1, 0) & np.roll(covered, 1, 1) & np.roll(covered, -1, 1))
    picks = [tuple(p) for kind in (np.argwhere(edge), np.argwhere(covered & ~edge), np.argwhere(~covered))
             for p in kind[rng.permutation(len(kind))[:n]]]
    return picks + [(0, 0), (H - 1, W - 1), (H // 2 - 1, 3), (H // 2, 3)]


def test_single_image_gradients_wrt_translation_rotation_bgcolor_vertex_color(gpu, oracle):
    """[gt, gr, gb, gc] = tf.gradients(im, [translation, rotation_xy, bgcolor, vertex_color], d_loss_by_pixels) with the
    values the reference feeds (:115): translation (0, 0, -0.25), rotation 0, bgcolor (.4, .2, .2), vertex_color (.7, .3, .6)."""
    rng = np.random.default_rng(0)
    vertices, faces = _geometry(gpu)
    vertex_count = vertices.shape[0]
    random_colors = torch.tensor(rng.uniform(size=[vertex_count - 75, 3]), dtype=torch.float32, device=gpu)
    translation = torch.tensor([0., 0., -0.25], device=gpu, requires_grad=True)
    rotation_xy = torch.tensor(0., device=gpu, requires_grad=True)
    bgcolor = torch.tensor([0.4, 0.2, 0.2], device=gpu, requires_grad=True)
    vertex_color = torch.tensor([0.7, 0.3, 0.6], device=gpu, requires_grad=True)

    projected_vertices = _project(vertices, translation, rotation_xy)
    vertex_colors = torch.cat([vertex_color[None, :].expand(75, 3), random_colors], dim=0)               # :88
    background = torch.cat([bgcolor[None, None, :].expand(H // 2, W, 3), torch.ones([H // 2, W, 3], device=gpu)], dim=0)
    im = dirt.rasterise_ops.rasterise(background, projected_vertices, vertex_colors, faces, height=H, width=W, channels=3)  # :90

    clip = projected_vertices.detach().cpu().numpy()[None]
    faces_np = faces.cpu().numpy()[None]
    want_im = oracle.forward(background.detach().cpu().numpy()[None], clip, vertex_colors.detach().cpu().numpy()[None], faces_np)
    assert np.array_equal(im.detach().cpu().numpy().view(np.uint32), want_im[0].view(np.uint32))
    jt, jr = _clip_jacobians([0., 0., -0.25], 0.)
    covered = (want_im[0] != background.detach().cpu().numpy()).any(-1)
    assert 200 < covered.sum() < 400

    losses = [rng.standard_normal([H, W, 3]).astype(np.float32)]
    for (y, x) in _indicators(rng, covered, 6):
        for c in range(3):
            pixel_indicator = np.zeros([H, W, 3], dtype=np.float32)
            pixel_indicator[y, x, c] = 1
            losses.append(pixel_indicator)
    nonzero = 0
    for d_loss_by_pixels in losses:
        gt, gr, gb, gc = torch.autograd.grad(im, [translation, rotation_xy, bgcolor, vertex_color],
                                             torch.from_numpy(d_loss_by_pixels).to(gpu), retain_graph=True)
        ow = oracle.backward(clip, faces_np, want_im, d_loss_by_pixels[None])
        want, bound = _contract(ow, 0, jt)
        _assert_within(gt, want, bound, 'd/d translation')
        want, bound = _contract(ow, 0, jr)
        _assert_within(gr, want, bound, 'd/d rotation_xy')
        nonzero += int(np.any(want != 0))
        # bgcolor fills the top half of the background (:90); grad_background is an exact copy of d_loss on uncovered pixels
        want_gb = ow['grad_background'][0][:H // 2].astype(np.float64).sum((0, 1))
        _assert_within(gb, want_gb, 1e-6 * np.abs(ow['grad_background'][0][:H // 2]).sum((0, 1)), 'd/d bgcolor')
        want_gc = ow['grad_vertex_colors'][0][:75].astype(np.float64).sum(0)
        _assert_within(gc, want_gc, parity.GRAD_TOL * ow['mass_vertex_colors'][0][:75].astype(np.float64).sum(0), 'd/d vertex_color')
    assert nonzero > len(losses) // 3, 'the sampled losses hardly touch the geometry'


def test_batch_gradients_wrt_translation_and_rotation(gpu, oracle):
    """[gst, gsr] = tf.gradients(ims, [translation, rotation_xy], ds_loss_by_pixels) over the batch of two
    (tests/rasterise_tests.py:89,96-97,123-132): translation (0, 0, -1), rotation 0.5, the geometry tiled."""
    rng = np.random.default_rng(1)
    vertices, faces = _geometry(gpu)
    vertex_count = vertices.shape[0]
    translation = torch.tensor([0., 0., -1.], device=gpu, requires_grad=True)
    rotation_xy = torch.tensor(0.5, device=gpu, requires_grad=True)
    projected_vertices = _project(vertices, translation, rotation_xy)
    backgrounds = torch.tensor([[0., 0., 0.], [0., 0., 1.]], device=gpu)[:, None, None, :].expand(2, H, W, 3).contiguous()
    colors = torch.tensor(rng.uniform(size=[2, vertex_count, 3]), dtype=torch.float32, device=gpu)
    ims = dirt.rasterise_ops.rasterise_batch(backgrounds, projected_vertices[None].expand(2, -1, -1), colors,
                                             faces[None].expand(2, -1, -1), height=H, width=W, channels=3)
    clip = np.tile(projected_vertices.detach().cpu().numpy()[None], [2, 1, 1])
    faces_np = np.tile(faces.cpu().numpy()[None], [2, 1, 1])
    want_ims = oracle.forward(backgrounds.cpu().numpy(), clip, colors.cpu().numpy(), faces_np)
    assert np.array_equal(ims.detach().cpu().numpy().view(np.uint32), want_ims.view(np.uint32))
    jt, jr = _clip_jacobians([0., 0., -1.], 0.5)
    covered = (want_ims[1] != backgrounds[1].cpu().numpy()).any(-1)

    losses = [rng.standard_normal([2, H, W, 3]).astype(np.float32)]
    for iib in range(2):
        for (y, x) in _indicators(rng, covered, 3):
            pixel_indicator = np.zeros([2, H, W, 3], dtype=np.float32)
            pixel_indicator[iib, y, x, int(rng.integers(3))] = 1
            losses.append(pixel_indicator)
    for ds_loss_by_pixels in losses:
        gst, gsr = torch.autograd.grad(ims, [translation, rotation_xy], torch.from_numpy(ds_loss_by_pixels).to(gpu), retain_graph=True)
        ow = oracle.backward(clip, faces_np, want_ims, ds_loss_by_pixels)
        want_t = want_r = bound_t = bound_r = 0.
        for iib in range(2):  # both scenes share the parameters: the tiled geometry sums
            w_, b_ = _contract(ow, iib, jt)
            want_t, bound_t = want_t + w_, bound_t + b_
            w_, b_ = _contract(ow, iib, jr)
            want_r, bound_r = want_r + w_, bound_r + b_
        _assert_within(gst, want_t, bound_t, 'batch d/d translation')
        _assert_within(gsr, want_r, bound_r, 'batch d/d rotation_xy')


def test_dirt_package_is_the_reference_surface():
    """dirt/__init__.py:1 and the modules the reference's samples and tests import."""
    for name in ('rasterise', 'rasterise_batch', 'rasterise_deferred', 'rasterise_batch_deferred'):
        assert callable(getattr(dirt, name))
    assert dirt.rasterise is dirt.rasterise_ops.rasterise
    for name in ('split_vertices_by_face', 'vertex_normals', 'vertex_normals_pre_split', 'diffuse_directional'):
        assert callable(getattr(dirt.lighting, name))
    for name in ('rodrigues', 'translation', 'scale', 'perspective_projection', 'compose', 'pad_3x3_to_4x4'):
        assert callable(getattr(dirt.matrices, name))
    assert math.isclose(float(dirt.matrices.perspective_projection(0.1, 20., 0.2, 0.75)[0, 0]), 0.5)
