"""The torch counterparts of dirt.matrices / dirt.lighting / dirt.projection (dirt_amd/) against the REFERENCE'S OWN
Python source: tests/golden/helpers_ref.npz holds the outputs of /root/reference/dirt/{matrices,lighting,projection}.py
executed over a numpy stand-in for TensorFlow (oracle/tf_shim, tests/golden/make_helpers_golden.py).  Runs on the CPU.
Where /root/reference is present the modules are also run live and must reproduce the committed vectors."""
import os

import numpy as np
import pytest
import torch

from dirt_amd import lighting, matrices, projection
from tests.golden import make_helpers_golden as gold

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'helpers_ref.npz')


class _Torch:
    """dirt_amd modules behind numpy inputs (float arrays -> float32 tensors, integer arrays -> int32)."""

    def __init__(self, module):
        self._m = module

    def __getattr__(self, name):
        fn = getattr(self._m, name)

        def call(*args, **kw):
            conv = lambda a: torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
            return fn(*[conv(a) for a in args], **{k: conv(v) for k, v in kw.items()})
        return call


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def test_torch_helpers_match_the_reference_modules():
    want = np.load(GOLDEN)
    got = gold.evaluate(_Torch(matrices), _Torch(lighting), _Torch(projection), gold.inputs(), _np)
    assert sorted(got) == sorted(k for k in want.files if not k.startswith('texture/'))
    for k in sorted(got):
        assert got[k].shape == want[k].shape, (k, got[k].shape, want[k].shape)
        if got[k].dtype.kind in 'iu':
            assert np.array_equal(got[k], want[k]), k
        else:
            scale = max(1.0, float(np.abs(want[k]).max()))
            assert np.allclose(got[k], want[k], rtol=2e-6, atol=2e-6 * scale), (k, float(np.abs(got[k] - want[k]).max()))


def test_reference_modules_reproduce_the_committed_vectors():
    if not gold.available():
        pytest.skip('/root/reference is not present (the vectors are committed)')
    m, l, p = gold.load_reference_helpers()
    live = gold.evaluate(m, l, p, gold.inputs(), lambda t: np.asarray(t))
    want = np.load(GOLDEN)
    for k in want.files:
        if not k.startswith('texture/'):
            assert np.array_equal(live[k], want[k]), k


def test_rodrigues_gradient_at_zero_matches_the_reference_construction():
    """The reference adds 1e-12 to the vector so that the derivative exists at zero (dirt/matrices.py:39-40): the torch port
    is differentiable there too and gives the generators of rotation."""
    v = torch.zeros(3, requires_grad=True)
    r = matrices.rodrigues(v, three_by_three=True)
    g = torch.autograd.grad(r[0, 1], v)[0]   # d R[0,1] / d v = -e_z in the reference's index convention (R[in, out])
    assert torch.isfinite(g).all() and abs(abs(float(g[2])) - 1.0) < 1e-3


# ------------------------------------------------------------------------------------------------ texture look-up

def _texture_reference():
    z = np.load(GOLDEN)
    return {k[len('texture/'):]: z[k] for k in z.files if k.startswith('texture/')}


def test_texture_oracle_matches_the_reference_sample_functions():
    """oracle/texture_oracle.py (what the fused HIP look-up is tested against, tests/test_texture.py) against
    `uvs_to_pixel_indices` + `sample_texture` of the reference's samples/textured.py:16-61, executed over the numpy
    TensorFlow stand-in.  Everywhere the reference's gather_nd stays inside the texture the two agree to float32 rounding;
    for samples inside the last texel row / column the reference reads row Ht / column Wt -- zeros on TensorFlow's GPU
    kernel, an error on its CPU kernel -- where this build uses the last texel (documented in the oracle): those samples,
    and only those, may differ."""
    from oracle import texture_oracle as tex
    x = gold.texture_inputs()
    want = _texture_reference()
    ht, wt = x['texture'].shape[:2]
    for mode in ('repeat', 'clamp'):
        idx = want['indices_' + mode]
        inside = (np.floor(idx[..., 0]) + 1 <= ht - 1) & (np.floor(idx[..., 1]) + 1 <= wt - 1)
        assert 0.2 < inside.mean() < 1.0   # (clamp mode sends every uv outside [0, 1) to the last texel: many samples sit on it)
        got = tex.sample_texture_uv(x['texture'], x['uvs'], mode=mode, filter='bilinear')
        assert np.allclose(got[inside], want['sample_%s_bilinear' % mode][inside], rtol=1e-6, atol=1e-6), mode
        assert not np.allclose(got[~inside], want['sample_%s_bilinear' % mode][~inside], atol=1e-3) or not (~inside).any()
        near_inside = (idx[..., 0] < ht) & (idx[..., 1] < wt)     # nearest truncates: only index == size is outside
        gotn = tex.sample_texture_uv(x['texture'], x['uvs'], mode=mode, filter='nearest')
        assert np.array_equal(gotn[near_inside], want['sample_%s_nearest' % mode][near_inside]), mode


def test_torch_texture_helpers_match_the_reference_sample_functions():
    """dirt_amd.texture.uvs_to_pixel_indices / sample_texture (same names and arguments as the sample's) likewise."""
    from dirt_amd import texture as tex
    x = gold.texture_inputs()
    want = _texture_reference()
    ht, wt = x['texture'].shape[:2]
    t, uv = torch.from_numpy(x['texture']), torch.from_numpy(x['uvs'])
    for mode in ('repeat', 'clamp'):
        idx = tex.uvs_to_pixel_indices(uv, (ht, wt), mode)
        assert np.allclose(idx.numpy(), want['indices_' + mode], rtol=1e-6, atol=1e-5), mode
        inside = (np.floor(want['indices_' + mode][..., 0]) + 1 <= ht - 1) & (np.floor(want['indices_' + mode][..., 1]) + 1 <= wt - 1)
        got = tex.sample_texture(t, idx, 'bilinear').numpy()
        assert np.allclose(got[inside], want['sample_%s_bilinear' % mode][inside], rtol=1e-5, atol=1e-5), mode


def test_reference_sample_functions_reproduce_the_committed_vectors():
    if not os.path.exists('/root/reference/samples/textured.py'):
        pytest.skip('/root/reference is not present (the vectors are committed)')
    f1, f2, to_tensor = gold.load_reference_texture_functions()
    live = gold.evaluate_texture(f1, f2, gold.texture_inputs(), lambda t: np.asarray(t), to_tensor)
    want = _texture_reference()
    for k in want:
        assert np.array_equal(live[k], want[k]), k
