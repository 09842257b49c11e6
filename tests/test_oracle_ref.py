"""Pins the oracle's backward restatement to the REFERENCE'S OWN kernel source -- runs without a GPU.

oracle/_ref/libdirt_ref.so is /root/reference/csrc/rasterise_grad_egl.cu (Vec3, assemble_grads,
launch_grad_assembly, upload_vertices) compiled for the host behind oracle/ref_shim/ (oracle/make_ref.py);
oracle/ref.py feeds it the oracle's visibility surfaces in the reference's atlas layout and wraps it in
the channel-group loop of dirt/rasterise_ops.py:132-177.  With DIRT_ORACLE_FLAG_F32_SEQUENTIAL the oracle
adds in float32 in the order one thread of that kernel does, so the two must agree BIT FOR BIT; in its
default mode (double accumulators, any order) the oracle must sit within float32 summation error of it.

Where /root/reference is absent and no prebuilt library travelled, the live comparisons skip and the
committed vectors tests/golden/ref_grads.npz (written here by `python -m tests.golden.make_golden --ref`)
stand in for the reference.
"""
import os

import numpy as np
import pytest

from tests import scenes
from tests.golden import make_golden

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
OUTPUTS = ('grad_background', 'grad_vertices', 'grad_vertex_colors', 'debug_thingy')


@pytest.fixture(scope='module')
def ref(oracle):
    from oracle import ref as _ref
    if not _ref.available():
        pytest.skip('oracle/_ref is not built and /root/reference is absent')
    return _ref


def _batch(s):
    keys = ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')
    return {k: (s[k] if s['background'].ndim == 4 else s[k][None]) for k in keys}


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _check_bitwise(oracle, ref, b, what):
    px = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    want = ref.backward(b['vertices'], b['faces'], px, b['grad_pixels'])
    got = oracle.backward(b['vertices'], b['faces'], px, b['grad_pixels'], flags=oracle.FLAG_F32_SEQUENTIAL, want_debug=True)
    for k in OUTPUTS:
        assert np.array_equal(_bits(got[k]), _bits(want[k])), '%s: %s differs from the reference kernel in %d elements' % (
            what, k, int(np.sum(_bits(got[k]) != _bits(want[k]))))
    # default mode: double accumulation in any order -> within float32 summation error of the reference, per element
    dbl = oracle.backward(b['vertices'], b['faces'], px, b['grad_pixels'], want_mass=True)
    assert np.array_equal(_bits(dbl['grad_background']), _bits(want['grad_background']))
    for k, m in (('grad_vertices', 'mass_vertices'), ('grad_vertex_colors', 'mass_vertex_colors')):
        finite = np.isfinite(want[k]) & np.isfinite(dbl[m])
        err = np.abs(dbl[k].astype(np.float64) - want[k])[finite]
        assert np.all(err <= 2e-6 * dbl[m][finite]), (what, k, float((err / np.maximum(dbl[m][finite], 1e-30)).max()))
        assert np.all(want[k][dbl[m] == 0] == 0), (what, k, 'the reference added something where the oracle saw no term')
    return want


CASES = {
    'cylinder': lambda: scenes.cylinder_scene(),                      # tests/rasterise_tests.py:50-99,115
    'cylinder_batch': lambda: scenes.cylinder_batch_scene(),          # tests/rasterise_tests.py:89,123-132
    'rand_c3': lambda: scenes.rand_scene(300, 96, 128, 3, seed=1, r_lo=0.03, r_hi=0.2),
    'rand_c4_groups_3_1': lambda: scenes.rand_scene(300, 96, 128, 4, seed=2, r_lo=0.03, r_hi=0.2),
    'q1_c1_single': lambda: scenes.rand_scene(300, 96, 128, 1, seed=3, r_lo=0.03, r_hi=0.2),
    'q1_c1_batch2': lambda: scenes.batch_scene(200, 48, 36, 1, seeds=[1, 2], r_lo=0.03, r_hi=0.2),
    'q1_c1_batch3_atlas_2x2': lambda: scenes.batch_scene(200, 48, 36, 1, seeds=[1, 2, 3], r_lo=0.03, r_hi=0.2),
    'c5_batch5_atlas_2x3': lambda: scenes.batch_scene(200, 48, 36, 5, seeds=[1, 2, 3, 4, 5], r_lo=0.03, r_hi=0.2),
    'c2_two_singles': lambda: scenes.rand_scene(120, 40, 56, 2, seed=9, r_lo=0.05, r_hi=0.3),
    'shared_vertices_c3': lambda: scenes.rand_scene(400, 72, 56, 3, seed=13, shared=True),
    'hostile_c4': lambda: scenes.hostile_scene(64, 80, 4, seed=5),
    'hostile_c1': lambda: scenes.hostile_scene(50, 70, 1, seed=6),
    'one_pixel_frame': lambda: scenes.rand_scene(5, 1, 1, 3, seed=4, r_lo=0.5, r_hi=1.0),
    'thin_frame': lambda: scenes.rand_scene(50, 2, 97, 3, seed=4, r_lo=0.1, r_hi=0.5),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_oracle_backward_equals_reference_kernel_bit_for_bit(oracle, ref, name):
    _check_bitwise(oracle, ref, _batch(CASES[name]()), name)


def test_bent_square_gbuffer_equals_reference_kernel(oracle, ref):
    """The 7-channel G-buffer of tests/deferred_grad_test.py:121-142: groups [0:3],[3:6],[6:7]."""
    b = make_golden.make_inputs(make_golden.CASES['bent_square_gbuffer'])
    _check_bitwise(oracle, ref, b, 'bent_square_gbuffer')


def test_k3_scene_equals_reference_kernel(oracle, ref):
    """BASELINE configs[2] at full size: 1024x1024x4, 10 000 triangles."""
    s = scenes.config_scene('K3')
    want = _check_bitwise(oracle, ref, _batch(s), 'K3')
    assert np.all(want['grad_vertices'][..., 2] == 0)  # .z is never written, csrc/rasterise_grad_egl.cu:228-230


def test_q1_intended_mode_differs_only_in_dilation_choice(oracle, ref):
    """DIRT_FLAG_Q1_INTENDED is NOT the reference's behaviour; the reference's kernel reads channels 1 and 2
    of a 1-channel tensor (csrc/rasterise_grad_egl.cu:119-123).  The default mode is the one pinned above;
    here: the two modes give different gradients on a scene where the aliasing matters, and the reference
    sides with the default."""
    b = _batch(scenes.rand_scene(300, 96, 128, 1, seed=3, r_lo=0.03, r_hi=0.2))
    px = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    want = ref.backward(b['vertices'], b['faces'], px, b['grad_pixels'])
    compat = oracle.backward(b['vertices'], b['faces'], px, b['grad_pixels'], flags=oracle.FLAG_F32_SEQUENTIAL)
    intended = oracle.backward(b['vertices'], b['faces'], px, b['grad_pixels'],
                               flags=oracle.FLAG_F32_SEQUENTIAL | oracle.FLAG_Q1_INTENDED)
    assert np.array_equal(_bits(compat['grad_vertices']), _bits(want['grad_vertices']))
    assert not np.array_equal(_bits(intended['grad_vertices']), _bits(want['grad_vertices']))
    # the colour / background gradients do not depend on the dilation direction
    assert np.array_equal(_bits(intended['grad_vertex_colors']), _bits(want['grad_vertex_colors']))


def test_reference_vertex_expansion(oracle, ref):
    """upload_vertices (csrc/rasterise_grad_egl.cu:11-33): the backward render draws vertex_in_face k of
    face f with position vertices[faces[f,k]], barycentric one-hot in (k==0, k==1) and the face's index triple
    as a flat attribute -- the (b, indices) convention the oracle's surfaces follow."""
    b = scenes.batch_scene(60, 16, 16, 3, seeds=[1, 2], r_lo=0.05, r_hi=0.3, shared=True)
    ex = ref.upload_vertices(b['vertices'], b['faces'])
    B, F = b['faces'].shape[:2]
    assert ex.shape == (B, 3 * F)
    for ib in range(B):
        f = b['faces'][ib]
        assert np.array_equal(ex['position'][ib], b['vertices'][ib][f.reshape(-1)])
        assert np.array_equal(ex['indices'][ib], np.repeat(f, 3, axis=0))
        k = np.tile(np.arange(3), F)
        assert np.array_equal(ex['barycentric'][ib], np.stack([k == 0, k == 1], -1).astype(np.float32))


def test_atlas_layout_matches_reference_formula(ref):
    """csrc/rasterise_grad_egl.cpp:408-414: horizontal_count = int(sqrt(B) + .1), rows as needed."""
    assert ref.atlas_shape(1, 10, 20) == (10, 20)
    assert ref.atlas_shape(2, 10, 20) == (20, 20)
    assert ref.atlas_shape(3, 10, 20) == (30, 20)
    assert ref.atlas_shape(4, 10, 20) == (20, 40)
    assert ref.atlas_shape(5, 10, 20) == (30, 40)
    assert ref.atlas_shape(9, 10, 20) == (30, 60)


@pytest.mark.parametrize('name', sorted(make_golden.CASES))
def test_oracle_reproduces_committed_reference_vectors(oracle, name):
    """Runs everywhere (no /root/reference needed): tests/golden/ref_grads.npz holds the reference kernel's
    outputs for the golden cases; the oracle in its sequential float32 mode must reproduce them bit for bit."""
    z = np.load(os.path.join(GOLDEN, 'ref_grads.npz'))
    b = make_golden.make_inputs(make_golden.CASES[name])
    px = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    assert np.array_equal(_bits(px), _bits(z[name + '/pixels'])), (name, 'pixels')   # reference movers around the flip-free draw
    got = oracle.backward(b['vertices'], b['faces'], px, b['grad_pixels'], flags=oracle.FLAG_F32_SEQUENTIAL, want_debug=True)
    for k in OUTPUTS:
        assert np.array_equal(_bits(got[k]), _bits(z[name + '/' + k])), (name, k)


# ------------------------------------------------------------------------------------ forward data movers

FORWARD_CASES = {
    'orientation_triangle_c3': lambda: dict(
        background=np.random.default_rng(0).uniform(0, 1, (1, 40, 60, 3)).astype(np.float32),
        vertices=np.array([[[0.5, 0.5, 0, 1], [0.9, 0.5, 0, 1], [0.7, 0.9, 0, 1]]], np.float32),   # clip y > 0: the TOP of the image
        vertex_colors=np.array([[[1, 0, 0], [0, 1, 0], [0, 0, 1]]], np.float32), faces=np.array([[[0, 1, 2]]], np.int32)),
    'cylinder': lambda: _batch(scenes.cylinder_scene()),
    'c1_batch3_atlas_2x2': lambda: scenes.batch_scene(200, 48, 36, 1, seeds=[1, 2, 3], r_lo=0.03, r_hi=0.2),
    'c4_groups_3_1_batch2': lambda: scenes.batch_scene(150, 33, 47, 4, seeds=[5, 6], r_lo=0.05, r_hi=0.3),
    'hostile_c3': lambda: _batch(scenes.hostile_scene(40, 56, 3, seed=8)),
}


@pytest.mark.parametrize('name', sorted(FORWARD_CASES))
def test_oracle_forward_equals_reference_movers_around_a_flip_free_draw(oracle, ref, name):
    """The forward op is upload_background -> GL draw -> download_pixels (csrc/rasterise_egl.cpp:348-392).  The two movers are
    the reference's own code (csrc/rasterise_egl.cu compiled for the host): the vertical flip, the atlas tiling, the
    replication of a single channel.  The draw between them is `oracle.draw_gl`, which addresses samples by GL window
    coordinates only and knows nothing about tensor rows.  Their composition must equal `oracle.forward`, which works in
    tensor orientation throughout: the forward restatement's y orientation and scene placement are then the reference's."""
    b = FORWARD_CASES[name]()
    want = ref.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    got = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    assert np.array_equal(_bits(got), _bits(want)), '%s: %d pixel values differ' % (name, int(np.sum(_bits(got) != _bits(want))))
    if name == 'orientation_triangle_c3':
        rows = np.nonzero((got[0] != b['background'][0]).any(-1))[0]
        assert rows.size and rows.max() <= 10 and rows.min() >= 2      # rows (1 - y) / 2 * 40 for y in [0.5, 0.9]


# ------------------------------------------------------------------------------- the reference's Python op layer

@pytest.fixture(scope='module')
def layer(ref):
    """dirt/rasterise_ops.py itself (imported from /root/reference over the numpy TensorFlow stand-in), its op library
    bound to the host-compiled reference kernels (oracle/ref.py::python_layer)."""
    import os
    if not os.path.exists('/root/reference/dirt/rasterise_ops.py'):
        pytest.skip('/root/reference is not present')
    return ref.python_layer()


@pytest.mark.parametrize('channels', [1, 2, 3, 4, 5, 7])
def test_reference_python_layer_equals_oracle(oracle, layer, channels):
    """`rasterise_batch` (dirt/rasterise_ops.py:51-108: dtype coercion, H/W/C inference, channel grouping, concatenation) and
    `_rasterise_grad_multichannel` (:132-177: per-group grad ops, float32 sum of grad_vertices) are the reference's own
    functions here, running on the reference's own kernels; the oracle's single-pass forward and its grouped backward must
    equal them bit for bit.  So the whole stack above the GL draw is the reference's code."""
    t = layer.tf_shim.convert_to_tensor
    b = scenes.batch_scene(120, 30, 44, channels, seeds=[5, 6, 7], r_lo=0.05, r_hi=0.3)
    px = np.asarray(layer.rasterise_batch(b['background'], b['vertices'], b['vertex_colors'], b['faces']))
    want = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    assert np.array_equal(_bits(px), _bits(want))
    g = layer._rasterise_grad_multichannel(t(b['vertices']), t(b['faces']), t(want), t(b['grad_pixels']), 'batch')
    o = oracle.backward(b['vertices'], b['faces'], want, b['grad_pixels'], flags=oracle.FLAG_F32_SEQUENTIAL)
    for k in ('grad_vertices', 'grad_vertex_colors', 'grad_background'):
        assert np.array_equal(_bits(np.asarray(g[k])), _bits(o[k])), (channels, k)
    # the single-scene entry points (:13-48, 'single' of :138-143)
    one = {k: b[k][1] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    px1 = np.asarray(layer.rasterise(one['background'], one['vertices'], one['vertex_colors'], one['faces']))
    want1 = oracle.forward(one['background'][None], one['vertices'][None], one['vertex_colors'][None], one['faces'][None])[0]
    assert np.array_equal(_bits(px1), _bits(want1))
    g1 = layer._rasterise_grad_multichannel(t(one['vertices']), t(one['faces']), t(want1), t(one['grad_pixels']), 'single')
    o1 = oracle.backward(one['vertices'][None], one['faces'][None], want1[None], one['grad_pixels'][None], flags=oracle.FLAG_F32_SEQUENTIAL)
    for k in ('grad_vertices', 'grad_vertex_colors', 'grad_background'):
        assert np.array_equal(_bits(np.asarray(g1[k])), _bits(o1[k][0])), (channels, k, 'single')


def test_reference_deferred_composition(oracle, layer):
    """`rasterise_deferred` (dirt/rasterise_ops.py:180-310) end to end in the reference's own code: G-buffer by its
    `rasterise`, the shader, and its `grad` closure -- vertex gradients from filtering the SHADED image, attribute and
    background gradients from the G-buffer with dL/dgbuffer (:204-237).  The one thing the stand-in cannot do, differentiate
    the shader (`tf.gradients`), is handed to torch.  The GPU tests compare dirt_amd's deferred path with exactly this
    composition written out on the oracle (tests/test_gpu_fullsize.py::_deferred_reference): here that composition is
    checked against the reference's."""
    import torch
    tf_shim = layer.tf_shim
    t = tf_shim.convert_to_tensor
    clip, faces, world, colours = scenes.bent_square_geometry()
    tri = world[:, :3].reshape(-1, 3, 3)
    n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    attrs = np.concatenate([np.ones([6, 1]), colours, np.repeat(n, 3, axis=0)], axis=1).astype(np.float32)   # mask, colour, normal
    H = W = 32
    bg_attrs = np.zeros([H, W, 7], np.float32)
    light = np.float32([0.3, 0.8, 0.5])
    graphs = {}

    def shader_fn(gbuffer, light_):
        g = torch.tensor(np.asarray(gbuffer), requires_grad=True)
        l = torch.tensor(np.asarray(light_), requires_grad=True)
        mask, cols, nrm = g[..., :1], g[..., 1:4], g[..., 4:7]
        px = mask * cols * (0.3 + (nrm * l).sum(-1, keepdim=True).abs()) + (1. - mask) * 0.1
        out = t(px.detach().numpy())
        graphs[id(out)] = (px, {id(gbuffer): g, id(light_): l})
        return out

    def gradients(ys, xs, grad_ys):
        px, inputs = graphs[id(ys)]
        grads = torch.autograd.grad(px, [inputs[id(x)] for x in xs], torch.tensor(np.asarray(grad_ys)))
        return [t(g.numpy()) for g in grads]

    tf_shim._gradients_hook = gradients
    try:
        pixels = layer.rasterise_deferred(bg_attrs, clip, attrs, faces, shader_fn, [light])
        d = np.random.default_rng(3).standard_normal((H, W, 3)).astype(np.float32)
        d_vertices, d_faces, d_attributes, d_background, d_light = pixels.dirt_grad_fn(t(d))
    finally:
        tf_shim._gradients_hook = None
    assert d_faces is None
    # the same composition written out on the oracle
    gbuf = oracle.forward(bg_attrs[None], clip[None], attrs[None], faces[None])
    gt = torch.tensor(gbuf[0], requires_grad=True)
    lt = torch.tensor(light, requires_grad=True)
    mask, cols, nrm = gt[..., :1], gt[..., 1:4], gt[..., 4:7]
    shaded = mask * cols * (0.3 + (nrm * lt).sum(-1, keepdim=True).abs()) + (1. - mask) * 0.1
    assert np.array_equal(_bits(np.asarray(pixels)), _bits(shaded.detach().numpy()))
    shaded.backward(torch.tensor(d))
    seq = oracle.FLAG_F32_SEQUENTIAL
    want_v = oracle.backward(clip[None], faces[None], shaded.detach().numpy()[None], d[None], flags=seq)
    want_a = oracle.backward(clip[None], faces[None], gbuf, gt.grad.numpy()[None], flags=seq)
    assert np.array_equal(_bits(np.asarray(d_vertices)), _bits(want_v['grad_vertices'][0]))
    assert np.array_equal(_bits(np.asarray(d_attributes)), _bits(want_a['grad_vertex_colors'][0]))
    assert np.array_equal(_bits(np.asarray(d_background)), _bits(want_a['grad_background'][0]))
    assert np.allclose(np.asarray(d_light), lt.grad.numpy())


def test_reference_square_test_script_runs_verbatim(ref):
    """/root/reference/tests/square_test.py -- the reference's only known-answer test -- executed AS IT IS: its `import
    tensorflow` finds the numpy stand-in, its `import dirt` the reference's own package, whose op library is the
    host-compiled reference movers around the oracle's GL draw.  It must print its success line (:54-57)."""
    import os
    path = '/root/reference/tests/square_test.py'
    if not os.path.exists(path):
        pytest.skip('/root/reference is not present')
    assert ref.run_reference_script(path) == 'successful: all pixels agree\n'


def test_reference_simple_sample_runs_verbatim_and_matches_the_port(oracle, ref):
    """/root/reference/samples/simple.py executed AS IT IS (numpy TensorFlow stand-in, the reference's own dirt package, a
    recording cv2 stub): the image it shows equals, to float rounding of the matrix helpers, what this repository's port of
    the sample (examples/simple.py's pipeline over dirt_amd.matrices / lighting) hands to the rasteriser, rendered by the
    oracle -- and the GPU tests check examples/simple.py against the oracle bit for bit (tests/test_gpu_fullsize.py)."""
    import os
    import torch
    path = '/root/reference/samples/simple.py'
    if not os.path.exists(path):
        pytest.skip('/root/reference is not present')
    ref.run_reference_script(path)
    (name, shown), = ref.run_reference_script.images
    reference_image = shown[:, :, ::-1]   # the sample shows BGR (samples/simple.py:78)
    assert reference_image.shape == (480, 640, 3)

    from dirt_amd import lighting, matrices
    import importlib.util
    spec = importlib.util.spec_from_file_location('example_simple', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', 'simple.py'))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    vertices, faces = ex.build_cube()
    v, f = lighting.split_vertices_by_face(torch.tensor(vertices, dtype=torch.float32), torch.tensor(faces, dtype=torch.int32))
    colors = torch.ones_like(v)
    v = torch.cat([v, torch.ones_like(v[:, -1:])], dim=1)
    world = v @ matrices.rodrigues(torch.tensor([0., 0.5, 0.]))
    normals = lighting.vertex_normals_pre_split(world, f)
    view = matrices.compose(matrices.translation(torch.tensor([0., -1.5, -3.5])), matrices.rodrigues(torch.tensor([-0.3, 0., 0.])))
    clip = (world @ view) @ matrices.perspective_projection(near=0.1, far=20., right=0.1, aspect=480. / 640.)
    lit = lighting.diffuse_directional(normals, colors, light_direction=torch.tensor([1., 0., 0.]), light_color=torch.tensor([1., 1., 1.])) * 0.8 + colors * 0.2
    ours = oracle.forward(np.zeros((1, 480, 640, 3), np.float32), clip.numpy()[None], lit.numpy()[None], f.numpy()[None])[0]
    differing = np.abs(ours - reference_image).max(-1) > 1e-5
    assert differing.mean() < 2e-4, 'the port of samples/simple.py and the sample itself differ in %d pixels' % int(differing.sum())
    assert 0.1 < float((reference_image.sum(-1) > 0).mean()) < 0.6


def _load_example(name):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('example_' + name, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'examples', name + '.py'))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    return ex


def _close_images(ours_u8, reference_u8, what):
    diff = np.abs(ours_u8.astype(np.int32) - reference_u8.astype(np.int32)).max(-1)
    assert ours_u8.shape == reference_u8.shape
    # the helpers' float rounding moves a value across an integer level here and there, and an edge pixel now and then
    assert (diff > 1).mean() < 5e-4, '%s: %d pixels differ by more than one level' % (what, int((diff > 1).sum()))


def test_reference_deferred_sample_runs_verbatim_and_matches_the_port(oracle, ref):
    """/root/reference/samples/deferred.py as it is -- a 10-channel G-buffer through the reference's `rasterise_deferred`,
    per-pixel ambient + diffuse + Phong lighting -- against examples/deferred.py's geometry and shader (this repository's port
    of the sample) over the oracle's G-buffer."""
    import os
    import torch
    path = '/root/reference/samples/deferred.py'
    if not os.path.exists(path):
        pytest.skip('/root/reference is not present')
    ref.run_reference_script(path)
    (name, reference_image), = ref.run_reference_script.images
    assert name == 'deferred.jpg' and reference_image.dtype == np.uint8
    ex = _load_example('deferred')
    from dirt_amd import matrices
    vertices, faces = (torch.from_numpy(a) for a in ex.build_cube())
    view = matrices.compose(matrices.translation(torch.tensor([0., -1.5, -3.5])), matrices.rodrigues(torch.tensor([-0.3, 0., 0.])))
    light = torch.nn.functional.normalize(torch.tensor([1., -0.3, -0.5]), dim=0)
    clip, faces, attributes = ex.geometry(vertices, faces, view)
    H, W = ex.frame_height, ex.frame_width
    gbuf = oracle.forward(np.zeros((1, H, W, 10), np.float32), clip.numpy()[None], attributes.numpy()[None], faces.numpy()[None])[0]
    ours = (ex.shader_fn(torch.from_numpy(gbuf), view, light) * 255).to(torch.uint8).numpy()
    _close_images(ours, reference_image, 'deferred')
    assert 0.1 < float((reference_image[..., 2] != 76).mean()) < 0.7   # background (0, 0, 0.3) * 255 = 76


def test_reference_textured_sample_runs_verbatim_and_matches_the_port(oracle, ref):
    """/root/reference/samples/textured.py as it is (its cat.jpg decoded by Pillow behind tf.image.decode_jpeg) against
    examples/textured.py's geometry with the same texture, the look-up by dirt_amd.texture's two reference-named helpers."""
    import os
    import torch
    path = '/root/reference/samples/textured.py'
    if not os.path.exists(path):
        pytest.skip('/root/reference is not present')
    ref.run_reference_script(path)
    (name, reference_image), = ref.run_reference_script.images
    assert name == 'textured.jpg' and reference_image.dtype == np.uint8
    ex = _load_example('textured')
    from PIL import Image
    from dirt_amd import lighting, texture as tex
    texture = torch.from_numpy(np.array(Image.open('/root/reference/samples/cat.jpg').convert('RGB'), dtype=np.uint8)).to(torch.float32) / 255.
    vertices, uvs, faces = (torch.from_numpy(a) for a in ex.build_cube())
    light = torch.nn.functional.normalize(torch.tensor([1., -0.3, -0.5]), dim=0)
    clip, attributes = ex.geometry(vertices, uvs, faces)
    H, W = ex.frame_height, ex.frame_width
    gbuf = torch.from_numpy(oracle.forward(np.zeros((1, H, W, 6), np.float32), clip.numpy()[None], attributes.numpy()[None], faces.numpy()[None])[0])
    mask, uv, normals = gbuf[..., :1], gbuf[..., 1:3], gbuf[..., 3:]
    unlit = tex.sample_texture(texture, tex.uvs_to_pixel_indices(uv, texture.shape[:2]))   # the example's fused look-up, unfused (CPU)
    diffuse = lighting.diffuse_directional(normals.reshape(-1, 3), unlit.reshape(-1, 3), light, light_color=torch.full((3,), 0.6), double_sided=True)
    shaded = (diffuse.reshape(unlit.shape) + unlit * 0.4) * mask + torch.tensor([0., 0., 0.3]) * (1. - mask)
    ours = (shaded * 255).to(torch.uint8).numpy()
    _close_images(ours, reference_image, 'textured')


def test_reference_multi_gpu_script_runs_verbatim(ref):
    """/root/reference/tests/multi_gpu_test.py as it is: the op under two `tf.device` scopes, evaluated in one session.  The
    stand-in has no devices, so all this checks is that the script's graph runs on the stack and draws its square twice; the
    two-device property itself is tests/test_gpu_configs.py::test_two_gpus_over_rccl (RCCL, one rank per GPU)."""
    import os
    path = '/root/reference/tests/multi_gpu_test.py'
    if not os.path.exists(path):
        pytest.skip('/root/reference is not present')
    assert ref.run_reference_script(path) == ''
