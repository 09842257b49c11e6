"""Full-size runs (BASELINE.json configs K3 and K5) checked through size-independent properties -- the
oracle needs seconds per scene at these sizes, the properties need none of it:

  * the visibility-only render and the forward render agree (pixels == background exactly where no face
    is visible; covered pixels differ from the background);
  * forward is idempotent and batch-consistent (a scene rendered alone == the same scene inside a batch);
  * translating the scene by whole pixels translates the visibility buffer;
  * sum_v grad_vertex_colors[v, c] == sum over covered pixels of grad_pixels[., c]   (sum_k b_k = 1);
  * grad_background == grad_pixels on uncovered pixels and 0 on covered ones (csrc/rasterise_grad_egl.cu:143-147);
  * grad_vertices[..., 2] == 0 (:228-230); gradients are linear in grad_pixels;
  * the stateless backward and the state-reusing backward agree.
"""
import numpy as np
import pytest
import torch

from tests import scenes
from dirt_amd import rasterise_ops as ops
from tests import parity

pytestmark = pytest.mark.gpu


def _dev(s, dev, keys=('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')):
    return {k: torch.from_numpy(np.ascontiguousarray(s[k]))[None].to(dev) for k in keys}


@pytest.mark.parametrize('config', ['K3', 'K5'])
def test_fullsize_properties(gpu, config):
    s = scenes.config_scene(config)
    H, W, C = s['height'], s['width'], s['channels']
    t = _dev(s, gpu)
    px, state = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C, keep_state=True)
    vis = ops._op_visibility(t['vertices'], t['faces'], H, W)
    covered = vis >= 0
    frac = float(covered.float().mean())
    assert 0.5 < frac < 0.95, frac                                  # ~85 % coverage, overdraw ~2
    # forward vs visibility
    same = (px == t['background']).all(-1)
    assert bool(same[~covered].all()), 'uncovered pixels must equal the background exactly'
    assert float(same[covered].float().mean()) < 1e-3
    # idempotent
    assert torch.equal(px, ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C))

    # backward: both variants
    a = ops._op_rasterise_grad(t['vertices'], t['faces'], px, t['grad_pixels'], H, W, C, state=state)
    a = [x.clone() if x is not None else None for x in a]
    b = ops._op_rasterise_grad(t['vertices'], t['faces'], px, t['grad_pixels'], H, W, C)
    gb, gv, gvc = a[0], a[1], a[2]
    assert torch.equal(gb, b[0])
    scale_v = float(b[1].abs().max()); scale_c = float(b[2].abs().max())
    assert float((gv - b[1]).abs().max()) <= 1e-4 * scale_v
    assert float((gvc - b[2]).abs().max()) <= 1e-4 * scale_c
    g = t['grad_pixels']
    assert torch.equal(gb, torch.where(covered[..., None], torch.zeros_like(g), g))
    assert bool((gv[..., 2] == 0).all())
    want = (g.double() * covered[..., None]).sum(dim=(0, 1, 2))
    got = gvc.double().sum(dim=(0, 1))
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-2 * float(want.abs().max())), (got, want)
    # linearity in grad_pixels
    c2 = ops._op_rasterise_grad(t['vertices'], t['faces'], px, 2.0 * g, H, W, C)
    assert float((c2[1] - 2.0 * b[1]).abs().max()) <= 2e-4 * scale_v
    assert float((c2[2] - 2.0 * b[2]).abs().max()) <= 2e-4 * scale_c


def test_whole_pixel_translation_translates_visibility(gpu):
    s = scenes.config_scene('K3')
    H, W = s['height'], s['width']
    v = torch.from_numpy(s['vertices'])[None].to(gpu)
    f = torch.from_numpy(s['faces'])[None].to(gpu)
    vis0 = ops._op_visibility(v, f, H, W)
    k = 64                                              # pixels; 2k/W is exactly representable
    v2 = v.clone()
    v2[..., 0] += (2.0 * k / W) * v[..., 3]            # x_ndc += 2k/W
    vis1 = ops._op_visibility(v2, f, H, W)
    a, b = vis0[:, :, :-k], vis1[:, :, k:]
    # faces partly pushed off screen keep their index; rounding in (x + w) * W/2 may move a handful of edge samples
    assert float((a != b).float().mean()) < 1e-4


def test_batch_of_k3_scenes_matches_single(gpu):
    b = scenes.batch_scene(10000, 1024, 1024, 4, seeds=[0, 1], r_lo=0.005, r_hi=0.04)
    t = {k: torch.from_numpy(b[k]).to(gpu) for k in ('background', 'vertices', 'vertex_colors', 'faces')}
    both = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], 1024, 1024, 4)
    for i in range(2):
        one = ops._op_rasterise(t['background'][i:i + 1], t['vertices'][i:i + 1], t['vertex_colors'][i:i + 1],
                                t['faces'][i:i + 1], 1024, 1024, 4)
        assert torch.equal(both[i:i + 1], one)


def test_deferred_matches_manual_composition(gpu, oracle):
    """rasterise_deferred (dirt/rasterise_ops.py:180-310): pixels = shader(gbuffer); vertex gradient from filtering
    the SHADED image (:204-210), attribute/background gradients from the G-buffer with dL/dgbuffer (:231-237)."""
    s = scenes.rand_scene(200, 48, 64, 5, 17, 0.05, 0.3)
    bg = torch.from_numpy(s['background']).to(gpu).requires_grad_(True)
    v = torch.from_numpy(s['vertices']).to(gpu).requires_grad_(True)
    attrs = torch.from_numpy(s['vertex_colors']).to(gpu).requires_grad_(True)
    f = torch.from_numpy(s['faces']).to(gpu)
    light = torch.tensor([0.3, 0.5, 0.8], device=gpu, requires_grad=True)

    def shader(gbuffer, light_):
        return gbuffer[..., :3] * (gbuffer[..., 3:4] + gbuffer[..., 4:5]) * light_

    px = ops.rasterise_deferred(bg, v, attrs, f, shader, [light])
    d = torch.from_numpy(np.random.default_rng(0).standard_normal((48, 64, 3)).astype(np.float32)).to(gpu)
    px.backward(d)

    gbuf = oracle.forward(s['background'][None], s['vertices'][None], s['vertex_colors'][None], s['faces'][None])
    gt = torch.from_numpy(gbuf[0]).to(gpu).requires_grad_(True)
    l2 = light.detach().clone().requires_grad_(True)
    shaded = shader(gt, l2)
    assert torch.allclose(px, shaded.detach(), atol=1e-6)
    shaded.backward(d)
    want_v = oracle.backward(s['vertices'][None], s['faces'][None], shaded.detach().cpu().numpy()[None], d.cpu().numpy()[None])
    want_a = oracle.backward(s['vertices'][None], s['faces'][None], gbuf, gt.grad.cpu().numpy()[None])
    parity.grad_close(v.grad, want_v, 'grad_vertices', 'vertices', 0)
    parity.grad_close(attrs.grad, want_a, 'grad_vertex_colors', 'attributes', 0)
    assert np.array_equal(bg.grad.cpu().numpy(), want_a['grad_background'][0]), 'background'
    assert torch.allclose(light.grad, l2.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('shaded_channels', [3, 6])
def test_batch_deferred_shares_one_visibility_pass(gpu, oracle, shaded_channels):
    """rasterise_batch_deferred with the 10-channel G-buffer of samples/deferred.py:107-112: the forward's set-up
    records + visibility serve both gradient calls (shaded image and G-buffer).  6 shaded channels exceed what
    the state was sized for, so that call renders again -- same numbers either way."""
    B, H, W, C = 2, 40, 56, 10
    batch = scenes.batch_scene(300, H, W, C, [3, 4], r_lo=0.05, r_hi=0.3)
    bg = torch.from_numpy(batch['background']).to(gpu).requires_grad_(True)
    v = torch.from_numpy(batch['vertices']).to(gpu).requires_grad_(True)
    attrs = torch.from_numpy(batch['vertex_colors']).to(gpu).requires_grad_(True)
    f = torch.from_numpy(batch['faces']).to(gpu)
    mix = torch.from_numpy(np.random.default_rng(5).uniform(0.1, 1.0, (9, shaded_channels)).astype(np.float32)).to(gpu)

    def shader(gbuffer):
        return gbuffer[..., :1] * torch.tanh(gbuffer[..., 1:] @ mix)

    px = ops.rasterise_batch_deferred(bg, v, attrs, f, shader)
    d = torch.from_numpy(np.random.default_rng(1).standard_normal((B, H, W, shaded_channels)).astype(np.float32)).to(gpu)
    px.backward(d)

    gbuf = oracle.forward(batch['background'], batch['vertices'], batch['vertex_colors'], batch['faces'])
    gt = torch.from_numpy(gbuf).to(gpu).requires_grad_(True)
    shaded = shader(gt)
    assert torch.allclose(px, shaded.detach(), atol=1e-6)
    shaded.backward(d)
    want_v = oracle.backward(batch['vertices'], batch['faces'], shaded.detach().cpu().numpy(), d.cpu().numpy())
    want_a = oracle.backward(batch['vertices'], batch['faces'], gbuf, gt.grad.cpu().numpy())

    parity.grad_close(v.grad, want_v, 'grad_vertices', 'vertices')
    parity.grad_close(attrs.grad, want_a, 'grad_vertex_colors', 'attributes')
    assert np.array_equal(bg.grad.cpu().numpy(), want_a['grad_background']), 'background'


def test_step_is_capturable_in_a_hip_graph(gpu):
    """The library only enqueues kernels on the caller's stream (no allocation, no synchronisation, no
    library-owned device state), so a forward + backward step can be captured once in a hipGraph and replayed:
    pixels identical, gradients equal up to the order of the float atomics."""
    F, H, W, C, seed, r_lo, r_hi = scenes.CONFIGS['K3']
    s = scenes.rand_scene(F, H, W, C, seed, r_lo, r_hi)
    bg, v, vc, f, g = (torch.from_numpy(s[k][None]).to(gpu) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels'))

    def step():
        px, state = ops._op_rasterise(bg, v, vc, f, H, W, C, keep_state=True)
        return (px,) + ops._op_rasterise_grad(v, f, px, g, H, W, C, state=state)[:3]

    ref = [t.clone() for t in step()]
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(out[0], ref[0]) and torch.equal(out[1], ref[1])
    for a, b in zip(out[2:], ref[2:]):
        scale = max(1.0, float(b.abs().max()))
        assert float((a - b).abs().max()) <= 1e-4 * scale


def test_simple_sample_end_to_end(gpu, oracle):
    """samples/simple.py:34-78 through dirt_amd.matrices / lighting / rasterise on the GPU: the image equals the
    oracle's on the very clip-space vertices the helpers produced, and a loss on it reaches the rotation
    parameter through the rasteriser's gradient and the matrix helpers."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('example_simple', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', 'simple.py'))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    from dirt_amd import lighting, matrices
    rotation = torch.tensor([0., 0.5, 0.], device=gpu, requires_grad=True)
    px = ex.render(rotation, gpu)
    assert px.shape == (480, 640, 3)
    # rebuild the same inputs to hand them to the oracle
    vertices, faces = ex.build_cube()
    v, f = lighting.split_vertices_by_face(torch.tensor(vertices, dtype=torch.float32, device=gpu), torch.tensor(faces, dtype=torch.int32, device=gpu))
    colors = torch.ones_like(v)
    v = torch.cat([v, torch.ones_like(v[:, -1:])], dim=1)
    world = v @ matrices.rodrigues(rotation.detach())
    normals = lighting.vertex_normals_pre_split(world, f)
    view = matrices.compose(matrices.translation(torch.tensor([0., -1.5, -3.5], device=gpu)), matrices.rodrigues(torch.tensor([-0.3, 0., 0.], device=gpu)))
    clip = (world @ view) @ matrices.perspective_projection(near=0.1, far=20., right=0.1, aspect=480. / 640.).to(gpu)
    lit = lighting.diffuse_directional(normals, colors, light_direction=torch.tensor([1., 0., 0.], device=gpu), light_color=torch.tensor([1., 1., 1.], device=gpu)) * 0.8 + colors * 0.2
    want = oracle.forward(np.zeros((1, 480, 640, 3), np.float32), clip.cpu().numpy()[None], lit.cpu().numpy()[None], f.cpu().numpy()[None])
    assert np.array_equal(px.detach().cpu().numpy().view(np.uint32), want[0].view(np.uint32))
    assert 0.1 < float((px.detach().sum(-1) > 0).float().mean()) < 0.6  # the cube covers a good part of the frame
    target = ex.render(torch.tensor([0., 0.35, 0.], device=gpu), gpu).detach()
    ((px - target) ** 2).sum().backward()
    assert torch.isfinite(rotation.grad).all() and float(rotation.grad.abs().max()) > 0
    assert float(rotation.grad[1]) > 0  # turning back towards the target (smaller angle) lowers the loss


def _load_example(name):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('example_' + name, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', name + '.py'))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    return ex


def test_lighting_views_script(gpu, oracle):
    """examples/lighting_views.py = the reference's tests/lighting_tests.py:13-66 on `import dirt` (it shows four images and
    asserts nothing): the four views render, every one equals the oracle's image of the very inputs the helpers produced
    where that is cheap to rebuild (the normals view), the point light on the shared and on the split mesh agree except
    where smooth and faceted normals differ, and the lights do what their arguments say."""
    ex = _load_example('lighting_views')
    views = ex.main(write_images=False, device=gpu)
    assert set(views) == {'normals', 'directional', 'point', 'point_split'}
    cover = [(v.amax(-1) > 0) for v in views.values()]
    for v, c in zip(views.values(), cover):
        assert v.shape == (ex.HEIGHT, ex.WIDTH, 3) and torch.isfinite(v).all()
        assert torch.equal(c, cover[0]) and 2000 < int(c.sum()) < 12000   # same silhouette in all four, a cylinder's worth of pixels
    # directional light: colour (1, 1, 0) + 0.4 blue -> red == green everywhere, blue exactly the offset on the mesh
    d = views['directional']
    assert torch.equal(d[..., 0], d[..., 1]) and float((d[..., 2][cover[0]] - 0.4).abs().max()) < 1e-6
    # split vs shared normals: same light, same geometry; interiors of the side faces agree to the faceting
    assert float((views['point'] - views['point_split']).abs().mean()) < 0.05
    # the normals view against the oracle on the inputs the script built
    from dirt_amd import lighting, matrices
    pts, tris = ex.cylinder(0.2, 0.75, 0.1, 0.2, 32)
    f = torch.from_numpy(tris).to(gpu)
    v = torch.cat([torch.from_numpy(pts), torch.ones(len(pts), 1)], dim=1).to(gpu)
    spin = torch.diag(torch.tensor([0.5, 0.5, 0.5, 1.], device=gpu))
    placed = v @ spin @ matrices.translation(torch.tensor([0., 0., -0.25], device=gpu))
    clip = placed @ matrices.perspective_projection(0.1, 20., 0.2, float(ex.HEIGHT) / ex.WIDTH).to(gpu)
    cols = lighting.vertex_normals(placed[:, :3], f).abs()
    want = oracle.forward(np.zeros((1, ex.HEIGHT, ex.WIDTH, 3), np.float32), clip.cpu().numpy()[None], cols.cpu().numpy()[None], tris[None])
    assert np.array_equal(views['normals'].cpu().numpy().view(np.uint32), want[0].view(np.uint32))


def test_deferred_jacobians_script(gpu):
    """examples/deferred_jacobians.py = the reference's tests/deferred_grad_test.py:168-259 on `import dirt` (it writes four
    PNGs and asserts nothing): per-pixel Jacobians of the directly lit and of the deferred-shaded bent square with respect to
    its five variables, 2 x 3072 backward passes.  Pixels of the two routes agree (lighting per vertex of a flat face ==
    lighting per pixel); shader-only variables -- light intensity, background colour -- have the
    same Jacobian on both routes; geometry variables reach the pixels on both routes with the same overall sensitivity."""
    ex = _load_example('deferred_jacobians')
    rep = ex.main(write_images=False, device=gpu)
    assert 150 < rep['covered_pixels'] < 900
    assert rep['pixels_max_abs_difference'] < 1e-5
    for name in ('light_intensity', 'background'):
        assert rep[name]['direct_l1'] > 1.0 and rep[name]['max_abs_difference'] < 1e-5, (name, rep[name])
    for name in ('translation', 'rotation'):
        a, b = rep[name]['direct_l1'], rep[name]['deferred_l1']
        assert a > 1.0 and b > 1.0 and abs(a - b) <= 0.25 * max(a, b), (name, rep[name])
    # (the reference's scene scales the HOMOGENEOUS world vertices, w included, with no translation: a uniform scale of clip
    # space, which no pixel can see -- both routes must say so)
    assert rep['scale']['direct_l1'] < 1e-3 and rep['scale']['deferred_l1'] < 1e-3, rep['scale']


def _deferred_reference(oracle, clip, faces, attributes, n_channels, H, W, shade, d):
    """Manual composition on the oracle (dirt/rasterise_ops.py:189-248): G-buffer by the oracle, shading and its autograd
    by torch, vertex gradients from the SHADED image, attribute gradients from the G-buffer."""
    gpu = clip.device
    clip_np, faces_np, attr_np = clip.detach().cpu().numpy(), faces.cpu().numpy(), attributes.detach().cpu().numpy()
    gbuf = oracle.forward(np.zeros((1, H, W, n_channels), np.float32), clip_np[None], attr_np[None], faces_np[None])
    gt = torch.from_numpy(gbuf[0]).to(gpu).requires_grad_(True)
    shaded = shade(gt)
    shaded.backward(d)
    want_v = oracle.backward(clip_np[None], faces_np[None], shaded.detach().cpu().numpy()[None], d.cpu().numpy()[None])
    want_a = oracle.backward(clip_np[None], faces_np[None], gbuf, gt.grad.cpu().numpy()[None])
    return shaded.detach(), want_v, want_a


def _close(got, want, what, tol=1e-4):
    want = want if isinstance(want, np.ndarray) else want.detach().cpu().numpy()
    scale = max(1.0, float(np.abs(want).max()))
    assert float(np.abs(got.detach().cpu().numpy() - want).max()) <= tol * scale, what


def test_textured_sample_end_to_end(gpu, oracle):
    """samples/textured.py on dirt_amd: deferred shading with the fused texture look-up.  The image, and the gradients
    that reach the clip-space vertices, the vertex attributes, the texture and the light direction, against the manual
    composition on the oracle with the UNFUSED torch helpers."""
    from dirt_amd import texture as tex
    ex = _load_example('textured')
    vertices, uvs, faces = (torch.from_numpy(a).to(gpu) for a in ex.build_cube())
    texture = torch.from_numpy(ex.checker_texture()).to(gpu).requires_grad_(True)
    light = torch.nn.functional.normalize(torch.tensor([1., -0.3, -0.5], device=gpu), dim=0).requires_grad_(True)
    clip, attributes = ex.geometry(vertices, uvs, faces)
    clip = clip.detach().requires_grad_(True)
    attributes = attributes.detach().requires_grad_(True)
    H, W = ex.frame_height, ex.frame_width
    px = ops.rasterise_deferred(torch.zeros([H, W, 6], device=gpu), clip, attributes, faces, ex.shader_fn, [texture, light])
    covered = (px - torch.tensor([0., 0., 0.3], device=gpu)).abs().sum(-1) > 1e-6
    assert 0.1 < float(covered.float().mean()) < 0.7
    d = (2.0 / px.numel()) * px.detach()            # d/dpixels of mean(pixels ** 2)
    px.backward(d)

    tex2 = texture.detach().clone().requires_grad_(True)
    light2 = light.detach().clone().requires_grad_(True)

    def shade(gbuffer):   # the example's shader with the look-up spelled out as the reference does (samples/textured.py:120-141)
        mask, uv, normals = gbuffer[..., :1], gbuffer[..., 1:3], gbuffer[..., 3:]
        unlit = tex.sample_texture(tex2, tex.uvs_to_pixel_indices(uv, tex2.shape[:2]))
        from dirt_amd import lighting
        diffuse = lighting.diffuse_directional(normals.reshape(-1, 3), unlit.reshape(-1, 3), light2,
                                               light_color=torch.full((3,), 0.6, device=gpu), double_sided=True)
        return (diffuse.reshape(unlit.shape) + unlit * 0.4) * mask + torch.tensor([0., 0., 0.3], device=gpu) * (1. - mask)

    shaded, want_v, want_a = _deferred_reference(oracle, clip, faces, attributes, 6, H, W, shade, d)
    assert torch.allclose(px, shaded, atol=1e-6)
    parity.grad_close(clip.grad, want_v, 'grad_vertices', 'clip-space vertices', 0)
    parity.grad_close(attributes.grad, want_a, 'grad_vertex_colors', 'vertex attributes', 0)
    _close(texture.grad, tex2.grad, 'texture')
    _close(light.grad, light2.grad, 'light direction', tol=1e-5)


def test_deferred_sample_end_to_end(gpu, oracle):
    """samples/deferred.py:58-117 on dirt_amd: the 10-channel G-buffer (mask, position, colour, normal) with per-pixel
    ambient + diffuse + Phong specular lighting, the view matrix and the light direction as shader inputs."""
    ex = _load_example('deferred')
    from dirt_amd import matrices
    vertices, faces = (torch.from_numpy(a).to(gpu) for a in ex.build_cube())
    view = matrices.compose(matrices.translation(torch.tensor([0., -1.5, -3.5], device=gpu)),
                            matrices.rodrigues(torch.tensor([-0.3, 0., 0.], device=gpu)))
    light = torch.nn.functional.normalize(torch.tensor([1., -0.3, -0.5], device=gpu), dim=0)
    clip, faces, attributes = ex.geometry(vertices, faces, view)
    clip = clip.detach().requires_grad_(True)
    attributes = attributes.detach().requires_grad_(True)
    view_in = view.detach().clone().requires_grad_(True)
    light_in = light.detach().clone().requires_grad_(True)
    H, W = ex.frame_height, ex.frame_width
    px = ops.rasterise_deferred(torch.zeros([H, W, 10], device=gpu), clip, attributes, faces, ex.shader_fn, [view_in, light_in])
    assert 0.1 < float((px[..., 2] != 0.3).float().mean()) < 0.7
    d = torch.from_numpy(np.random.default_rng(8).standard_normal((H, W, 3)).astype(np.float32)).to(gpu) / (H * W)
    px.backward(d)
    view2 = view.detach().clone().requires_grad_(True)
    light2 = light.detach().clone().requires_grad_(True)
    shaded, want_v, want_a = _deferred_reference(oracle, clip, faces, attributes, 10, H, W, lambda g: ex.shader_fn(g, view2, light2), d)
    assert torch.allclose(px, shaded, atol=1e-6)
    parity.grad_close(clip.grad, want_v, 'grad_vertices', 'clip-space vertices', 0)
    parity.grad_close(attributes.grad, want_a, 'grad_vertex_colors', 'vertex attributes', 0)
    _close(view_in.grad, view2.grad, 'view matrix', tol=1e-5)
    _close(light_in.grad, light2.grad, 'light direction', tol=1e-5)


def _k5_shader(g, light):
    """A smooth 16 -> 3 channel per-pixel shader in the manner of samples/deferred.py:58-103: a mask channel, an albedo
    triple, a normal triple lit by a direction, and an emissive triple."""
    mask, albedo, normal, emissive = g[..., 0:1], g[..., 4:7], g[..., 7:10], g[..., 10:13]
    diffuse = torch.relu((normal * light).sum(-1, keepdim=True))
    return mask * albedo * (0.3 + diffuse) + 0.1 * emissive * g[..., 13:14]


def test_k5_through_rasterise_deferred_at_full_size(gpu, oracle):
    """BASELINE.json config 5 as stated: `rasterise_deferred` (dirt/rasterise_ops.py:180-257, samples/deferred.py:105-117)
    on the K5 scene -- 2048 x 2048 x 16 G-buffer, 50 000 triangles, 3-channel shader.  Both gradient passes (vertex
    gradients from filtering the SHADED image, attribute / background gradients from the G-buffer) share the forward's
    state, and both are compared per element with the oracle composition of _deferred_reference."""
    s = scenes.config_scene('K5')
    H, W, C = s['height'], s['width'], s['channels']
    assert (H, W, C, s['faces'].shape[0]) == (2048, 2048, 16, 50000)
    bg = torch.from_numpy(s['background']).to(gpu).requires_grad_(True)
    v = torch.from_numpy(s['vertices']).to(gpu).requires_grad_(True)
    attrs = torch.from_numpy(s['vertex_colors']).to(gpu).requires_grad_(True)
    f = torch.from_numpy(s['faces']).to(gpu)
    light = torch.nn.functional.normalize(torch.tensor([0.4, 0.5, 0.7], device=gpu), dim=0).requires_grad_(True)
    px = ops.rasterise_deferred(bg, v, attrs, f, _k5_shader, [light])
    d = torch.from_numpy(np.random.default_rng(11).standard_normal((H, W, 3)).astype(np.float32)).to(gpu)
    px.backward(d)

    gbuf = oracle.forward(s['background'][None], s['vertices'][None], s['vertex_colors'][None], s['faces'][None])
    gt = torch.from_numpy(gbuf[0]).to(gpu).requires_grad_(True)
    l2 = light.detach().clone().requires_grad_(True)
    shaded = _k5_shader(gt, l2)
    assert torch.allclose(px, shaded.detach(), atol=1e-6)
    shaded.backward(d)
    want_v = oracle.backward(s['vertices'][None], s['faces'][None], shaded.detach().cpu().numpy()[None], d.cpu().numpy()[None])
    want_a = oracle.backward(s['vertices'][None], s['faces'][None], gbuf, gt.grad.cpu().numpy()[None])
    parity.grad_close(v.grad, want_v, 'grad_vertices', 'K5 deferred: vertices (from the shaded image)', 0)
    parity.grad_close(attrs.grad, want_a, 'grad_vertex_colors', 'K5 deferred: attributes (from the G-buffer)', 0)
    assert np.array_equal(bg.grad.cpu().numpy(), want_a['grad_background'][0]), 'K5 deferred: background attributes'
    assert torch.allclose(light.grad, l2.grad, rtol=1e-4, atol=1e-3 * float(l2.grad.abs().max()))


def test_graphed_step_equals_the_eager_autograd_path(gpu, oracle):
    """dirt_amd.GraphedStep: rasterise_batch -> loss -> gradients captured once as a HIP graph (the remedy for eager
    autograd's host cost; the reference registers its gradient into a TensorFlow graph that session.run replays,
    dirt/rasterise_ops.py:111-129).  Replays must give the eager path's pixels bit for bit and gradients within the tight
    per-element tolerance of the oracle; inputs updated IN PLACE between replays must be seen; the vjp form
    (grad_pixels) must equal backward(grad_pixels)."""
    import dirt_amd
    TIGHT_TOL = parity.TIGHT_TOL
    F, H, W, C = 600, 160, 192, 4
    s = scenes.rand_scene(F, H, W, C, 31, 0.02, 0.2)
    bg, v, vc, f, g = (torch.from_numpy(s[k][None].copy()).to(gpu) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels'))
    # vjp form
    step = dirt_amd.GraphedStep(bg, v, vc, f, grad_pixels=g)
    for _ in range(3):
        px, (gb, gv, gvc) = step()
    torch.cuda.synchronize()
    want = oracle.forward(s['background'][None], s['vertices'][None], s['vertex_colors'][None], s['faces'][None])
    assert np.array_equal(px.cpu().numpy().view(np.uint32), want.view(np.uint32))
    ow = oracle.backward(s['vertices'][None], s['faces'][None], want, s['grad_pixels'][None], want_mass=True)
    assert np.array_equal(gb.cpu().numpy(), ow['grad_background'])
    parity.grads_close(gv, gvc, ow, 'graphed vjp', tol=TIGHT_TOL)
    assert gv.is_contiguous() and gvc.is_contiguous() and tuple(gv.shape) == (1, v.shape[1], 4) and tuple(gvc.shape) == (1, v.shape[1], C)
    # inputs updated in place are what the next replay renders
    with torch.no_grad():
        v[..., 0] += 0.03 * v[..., 3]
        g.mul_(0.5)
    px, (gb, gv, gvc) = step()
    torch.cuda.synchronize()
    v2 = v.cpu().numpy()
    want2 = oracle.forward(s['background'][None], v2, s['vertex_colors'][None], s['faces'][None])
    assert np.array_equal(px.cpu().numpy().view(np.uint32), want2.view(np.uint32))
    ow2 = oracle.backward(v2, s['faces'][None], want2, g.cpu().numpy(), want_mass=True)
    parity.grads_close(gv, gvc, ow2, 'graphed vjp after an in-place update', tol=TIGHT_TOL)
    # loss form against eager autograd on the same tensors
    target = torch.rand_like(bg)
    loss_fn = lambda p: ((p - target) ** 2).mean()
    step2 = dirt_amd.GraphedStep(bg, v, vc, f, loss_fn=loss_fn)
    loss, (gb, gv, gvc) = step2()
    leaves = [t.detach().clone().requires_grad_(True) for t in (bg, v, vc)]
    eager_loss = loss_fn(dirt_amd.rasterise_batch(leaves[0], leaves[1], leaves[2], f))
    eager_loss.backward()
    torch.cuda.synchronize()
    assert float((loss - eager_loss).abs()) <= 1e-6 * max(1.0, float(eager_loss.abs()))
    assert torch.equal(gb, leaves[0].grad)
    gp = (2.0 / target.numel()) * (torch.from_numpy(want2).to(gpu) - target)
    ow3 = oracle.backward(v2, s['faces'][None], want2, gp.cpu().numpy(), want_mass=True)
    parity.grads_close(gv, gvc, ow3, 'graphed loss', tol=TIGHT_TOL)
    parity.grads_close(leaves[1].grad, leaves[2].grad, ow3, 'eager loss', tol=TIGHT_TOL)


def test_second_backward_over_one_forward_returns_its_own_gradients(gpu, oracle):
    """The RasteriseGrad op is pure (csrc/rasterise_grad_egl.cu:244-250 clears its outputs on every call).  Here a keep-state
    forward pre-clears what ONE backward call adds into; the library remembers (host side, per workspace) whether that is
    still so and clears otherwise -- at the C-ABI level, not only in the autograd wrapper: a second call with
    DIRT_FLAG_DENSE_FROM_STATE, with the state's own accumulators as outputs, or with DIRT_FLAG_OUTPUTS_CLEARED must not
    return the sum of two passes."""
    F, H, W, C = 300, 96, 128, 4
    s = scenes.rand_scene(F, H, W, C, 17, 0.03, 0.25)
    b = {k: s[k][None] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    d = {k: torch.from_numpy(np.ascontiguousarray(b[k])).to(gpu) for k in b}
    want = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    ow = oracle.backward(b['vertices'], b['faces'], want, b['grad_pixels'], want_mass=True)
    for mode in ('dense', True):
        for dense_grads in (False, True):
            px, state = ops._op_rasterise(d['background'], d['vertices'], d['vertex_colors'], d['faces'], H, W, C, keep_state=True, dense_grads=dense_grads)
            for call in range(3):
                _, gv, gvc, _ = ops._op_rasterise_grad(d['vertices'], d['faces'], px, d['grad_pixels'], H, W, C, state=state, state_outputs=mode)
                parity.grads_close(gv.clone(), gvc.clone(), ow, 'call %d, state_outputs=%r, dense_grads=%r' % (call, mode, dense_grads))
    # the raw C ABI: OUTPUTS_CLEARED with tensors the forward never saw must clear them, not trust the flag
    from dirt_amd import _lib
    px, state = ops._op_rasterise(d['background'], d['vertices'], d['vertex_colors'], d['faces'], H, W, C, keep_state=True, dense_grads=True)
    gv = torch.full_like(d['vertices'], 7.0)
    gvc = torch.full((1, d['vertices'].shape[1], C), 7.0, device=gpu)
    gb = torch.empty_like(px)
    lib = _lib.load()
    _lib.check(lib.dirt_rasterise_backward(d['vertices'].data_ptr(), d['faces'].data_ptr(), px.data_ptr(), d['grad_pixels'].data_ptr(),
                                           gb.data_ptr(), gv.data_ptr(), gvc.data_ptr(), None, 1, d['vertices'].shape[1], F, H, W, C,
                                           state.data_ptr(), state.numel(), _lib.FLAG_REUSE_STATE | _lib.FLAG_OUTPUTS_CLEARED,
                                           torch.cuda.current_stream().cuda_stream))
    parity.grads_close(gv, gvc, ow, 'OUTPUTS_CLEARED on foreign tensors')


def test_graphed_optimisation_example_reduces_the_loss(gpu):
    """examples/fit_colors_graphed.py: a static-shape descent loop, one HIP-graph launch per iteration, parameters updated in
    place between launches: the loss must fall substantially."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', 'fit_colors_graphed.py')
    spec = importlib.util.spec_from_file_location('example_fit_colors_graphed', path)
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    history = ex.fit(gpu, steps=40, verbose=False)
    assert np.isfinite(history).all()
    assert history[-1] < 0.5 * history[0], (history[0], history[-1])


# ---- the kernel-selection table: the library's own choice must be (close to) the fastest eligible shape ---------------

def _time_shapes(config, gpu, flag_sets, steps=60, reps=3):
    """{name: (raster us, gradient us)}: the library's per-kernel HIP-event averages of forward + backward steps of `config`
    under each flag set (min over `reps` repetitions of `steps` steps)."""
    from dirt_amd import _lib
    s = scenes.config_scene(config)
    H, W, C = s['background'].shape
    t = {k: torch.from_numpy(np.ascontiguousarray(s[k]))[None].to(gpu) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    out = {}
    for name, flags in flag_sets:
        def step(fl):
            px, st = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C, flags=fl, keep_state=True, dense_grads=True)
            ops._op_rasterise_grad(t['vertices'], t['faces'], px, t['grad_pixels'], H, W, C, flags=fl, state=st, state_outputs='dense')
        for _ in range(10):
            step(flags)
        best = None
        for _ in range(reps):
            _lib.profile_reset()
            torch.cuda.synchronize()
            for _ in range(steps):
                step(flags | _lib.FLAG_PROFILE)
            torch.cuda.synchronize()
            prof = {k: ms / n * 1e3 for k, (ms, n) in _lib.profile_read().items() if n}
            cur = (prof.get('raster_kernel<shade>', 0.0), prof.get('grad_kernel', 0.0))
            best = cur if best is None else (min(best[0], cur[0]), min(best[1], cur[1]))
        out[name] = best
    return out


@pytest.mark.parametrize('config', ['K3', 'K3-768', 'K3-3ch', 'K3-256', 'K3-2048'])
def test_library_picks_the_fastest_kernel_shape(gpu, config):
    """launch_grad / launch_raster choose among kernel shapes by thresholds that were measured once (dirt_grad.hip, dirt_raster.hip,
    dirt_forward.hip).  Here every eligible shape is timed on the BASELINE configurations and the library's own choice
    (flags = 0) must be within 5 % of the fastest (+ 0.7 us: the event pairs' own jitter) -- so a kernel change that moves a
    crossover fails a test instead of leaving a stale table behind."""
    from dirt_amd import _lib
    C = scenes.CONFIGS[config][3]
    grad_sets = [('auto', 0), ('pairs', _lib.FLAG_GRAD_PAIRS), ('rows', _lib.FLAG_GRAD_ROWS)]
    if C in (1, 3, 4):
        grad_sets += [('px2', _lib.FLAG_GRAD_PX2), ('small', _lib.FLAG_GRAD_SMALL)]
    if C == 4:
        grad_sets += [('stream', _lib.FLAG_GRAD_STREAM)]
    tile_sets = [('auto', 0), ('large', _lib.FLAG_TILES_LARGE), ('small', _lib.FLAG_TILES_SMALL),
                 ('large8', _lib.FLAG_TILES_LARGE | _lib.FLAG_TILES_SMALL)]   # (both bits: 32 x 32 tiles, eight half-size waves each)
    for attempt in range(2):
        g = _time_shapes(config, gpu, grad_sets)
        r = _time_shapes(config, gpu, tile_sets)
        best_g = min(v[1] for v in g.values())
        best_r = min(v[0] for v in r.values())
        ok = g['auto'][1] <= 1.05 * best_g + 0.7 and r['auto'][0] <= 1.05 * best_r + 0.7
        if ok:
            break
    report = '%s: gradient %s | raster %s' % (config, {k: round(v[1], 1) for k, v in g.items()}, {k: round(v[0], 1) for k, v in r.items()})
    print(report)
    assert ok, 'the library\'s choice is more than 5 %% slower than another eligible shape -- ' + report
