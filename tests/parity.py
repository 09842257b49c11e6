"""The gradient tolerance of the parity tests (shared by tests/, tests/fuzz_parity.py and smoke()).

BASELINE.json north_star: "fp32 gradients within 1e-4".  The reference accumulates `grad_vertices` and
`grad_vertex_colors` with float atomics in unspecified order (csrc/rasterise_grad_egl.cu:140,228-230), so each
output element is only defined up to the rounding of a sum; the scale of that rounding is the L1 mass of
the terms added into THAT element, which the oracle returns (`mass_vertices`, `mass_vertex_colors`:
oracle/oracle.py::backward(want_mass=True)).  The check is per element:

    |gpu - oracle| <= tol * mass[element] + 2^-23 * cond[element]        tol: 1e-4 (the specification); 5e-6 in every GPU test (TIGHT_TOL)

`cond` (position gradients only; `cond_vertices`) is the CANCELLATION scale of the element's terms -- the same products
with the Scharr filter's differences and sum_k b_k * vertex_k.x taken over magnitudes.  It matters where a term is the
rounding residue of an exactly cancelling difference: in a frame one pixel wide every x tap is the same pixel and the
reference's ((a + b) - a) - b is +-1 ulp of the taps, not 0; such an element's "mass" is 1e-10 of its neighbours' and
its value is defined by the reference only up to an ulp of the taps (nvcc's own fma contraction would change it).
2^-23 is TWO float32 ulps of that scale.  Rounds 3-4 needed 16 (2^-20): the kernels formed a fragment's NDC position as
(i + 0.5) * (2 / n) - 1, which has an absolute error of an ulp of 1 and therefore no correct digit at the centre of the frame,
where the reference's sum_k b_k * vertex_k.x is small and accurate; round 5's fuzz sweep found two elements at 17 and 27 ulps,
which led to the exact-numerator form (dirt_grad_common.h::ndc_of).  Since then: 0 mismatches in 2 229 cases at 2^-20 and in
~520 cases each at 2^-20, 2^-22 and 2^-24 (profiles/r05_cond_study.txt); with the term removed altogether two residue
elements of 1e-9 of their neighbours' mass differ by 0.06 ulp of the scale.  Where both scales are 0 the GPU value must be
exactly 0.

Non-finite values (hostile geometry: clip_w underflow) must be non-finite on both sides in the same places.
"""
import numpy as np

GRAD_TOL = 1e-4
TIGHT_TOL = 5e-6  # 10 x the worst measured error / mass (5.4e-7, hostile geometry: profiles/r04_tolerance_probe.txt); what the GPU tests assert
import os as _os
COND_ULPS = float(_os.environ.get('DIRT_COND_ULPS', 2.0 ** -23))   # 2 float32 ulps of the cancellation scale (the override: tolerance studies)
KEYS = {'grad_vertices': 'mass_vertices', 'grad_vertex_colors': 'mass_vertex_colors'}


def _np(a):
    if isinstance(a, np.ndarray):
        return a
    return a.detach().cpu().numpy()


def grad_close(got, ow, key, what='', index=None, tol=GRAD_TOL):
    """`ow` is the oracle's backward dict (with masses), `key` one of KEYS; `index` selects a scene."""
    want, mass = ow[key], ow[KEYS[key]]
    cond = ow['cond_vertices'] if key == 'grad_vertices' else np.zeros_like(mass)
    if index is not None:
        want, mass, cond = want[index], mass[index], cond[index]
    got = _np(got).astype(np.float64)
    assert got.shape == want.shape, '%s %s: shape %s vs %s' % (what, key, got.shape, want.shape)
    bad_w = ~(np.isfinite(want) & np.isfinite(mass) & np.isfinite(cond))
    bad_g = ~np.isfinite(got)
    assert np.array_equal(bad_w, bad_g), '%s %s: non-finite values in different places (%d oracle, %d gpu)' % (
        what, key, int(bad_w.sum()), int(bad_g.sum()))
    ok = ~bad_w
    err = np.abs(got - want.astype(np.float64))[ok]
    lim = tol * mass.astype(np.float64)[ok] + COND_ULPS * cond.astype(np.float64)[ok]
    if err.size and not np.all(err <= lim):
        worst = int(np.argmax(err - lim))
        raise AssertionError('%s %s: %d of %d elements outside %g * mass + 2^-23 * cond; worst err %g at mass %g, cond %g (value %g)' % (
            what, key, int(np.sum(err > lim)), err.size, tol, err[worst], mass[ok][worst], cond[ok][worst], want[ok][worst]))


def grads_close(got_vertices, got_vertex_colors, ow, what='', index=None, tol=GRAD_TOL):
    grad_close(got_vertices, ow, 'grad_vertices', what, index, tol)
    grad_close(got_vertex_colors, ow, 'grad_vertex_colors', what, index, tol)
