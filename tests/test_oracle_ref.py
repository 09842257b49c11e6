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

from dirt_amd import scenes
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
