"""The gradient tolerance of the parity tests (shared by tests/, tests/fuzz_parity.py and smoke()).

BASELINE.json north_star: "fp32 gradients within 1e-4".  The reference accumulates `grad_vertices` and
`grad_vertex_colors` with float atomics in unspecified order (csrc/rasterise_grad_egl.cu:140,228-230), so each
output element is only defined up to the rounding of a sum; the scale of that rounding is the L1 mass of
the terms added into THAT element, which the oracle returns (`mass_vertices`, `mass_vertex_colors`:
oracle/oracle.py::backward(want_mass=True)).  The check is per element:

    |gpu - oracle| <= 1e-4 * mass[element]          (mass == 0  =>  gpu must be exactly 0)

Non-finite values (hostile geometry: clip_w underflow) must be non-finite on both sides in the same places.
"""
import numpy as np

GRAD_TOL = 1e-4
KEYS = {'grad_vertices': 'mass_vertices', 'grad_vertex_colors': 'mass_vertex_colors'}


def _np(a):
    if isinstance(a, np.ndarray):
        return a
    return a.detach().cpu().numpy()


def grad_close(got, ow, key, what='', index=None, tol=GRAD_TOL):
    """`ow` is the oracle's backward dict (with masses), `key` one of KEYS; `index` selects a scene."""
    want, mass = ow[key], ow[KEYS[key]]
    if index is not None:
        want, mass = want[index], mass[index]
    got = _np(got).astype(np.float64)
    assert got.shape == want.shape, '%s %s: shape %s vs %s' % (what, key, got.shape, want.shape)
    bad_w = ~(np.isfinite(want) & np.isfinite(mass))
    bad_g = ~np.isfinite(got)
    assert np.array_equal(bad_w, bad_g), '%s %s: non-finite values in different places (%d oracle, %d gpu)' % (
        what, key, int(bad_w.sum()), int(bad_g.sum()))
    ok = ~bad_w
    err = np.abs(got - want.astype(np.float64))[ok]
    lim = tol * mass.astype(np.float64)[ok]
    if err.size and not np.all(err <= lim):
        worst = int(np.argmax(err - lim))
        raise AssertionError('%s %s: %d of %d elements outside %g * mass; worst err %g at mass %g (value %g)' % (
            what, key, int(np.sum(err > lim)), err.size, tol, err[worst], lim[worst] / tol, want[ok][worst]))


def grads_close(got_vertices, got_vertex_colors, ow, what='', index=None, tol=GRAD_TOL):
    grad_close(got_vertices, ow, 'grad_vertices', what, index, tol)
    grad_close(got_vertex_colors, ow, 'grad_vertex_colors', what, index, tol)
