"""Randomised parity sweep on an MI355X box: random frame sizes, channel counts, meshes (split / shared / hostile),
kernel tile shapes and flags; every case compares the HIP path with the CPU oracle (forward
and visibility bit for bit, gradients per element within 5e-6 (parity.TIGHT_TOL; the specification says 1e-4) of the L1 mass of their terms, non-finite values in the same
places: tests/parity.py).  A fixed-seed slice of it runs under pytest (tests/test_gpu_configs.py); as a script it is open-ended (a time budget);
usage: python tests/fuzz_parity.py [seconds] [seed] [hostile]   (`hostile`: mostly hostile geometry, larger frames)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
from dirt_amd import rasterise_ops as ops
from tests import scenes  # noqa: E402
from tests import parity  # noqa: E402


def run(budget=None, max_cases=None, seed=0, hard=False, max_dim=None, failures=None):
    """Random cases until `budget` seconds have passed or `max_cases` are done; returns the number of cases.
    `failures`: a list to collect gradient mismatches in instead of raising at the first (the open-ended sweep)."""
    rng = np.random.default_rng(seed)
    dev = torch.device('cuda', 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    t0, n = time.time(), 0
    top = max_dim or (700 if hard else 400)
    while (budget is None or time.time() - t0 < budget) and (max_cases is None or n < max_cases):
        H, W = int(rng.integers(1, top)), int(rng.integers(1, top))
        C = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 8, 10, 16]))
        kind = rng.choice(['split', 'shared', 'hostile', 'tiny'], p=[0.1, 0.1, 0.7, 0.1] if hard else None)
        seed_ = int(rng.integers(0, 1 << 30))
        if kind == 'hostile':
            s = scenes.hostile_scene(H, W, C, seed_, int(rng.integers(10, 1500)))
        elif kind == 'tiny':
            s = scenes.rand_scene(int(rng.integers(1, 4000)), H, W, C, seed_, 0.001, 0.02)
        else:
            s = scenes.rand_scene(int(rng.integers(1, 3000)), H, W, C, seed_, float(rng.uniform(0.005, 0.1)), float(rng.uniform(0.1, 0.8)), kind == 'shared')
        flags = int(rng.choice([0, 0x200, 0x400, 0x600])) | int(rng.choice([0, 1])) | int(rng.choice([0, 0x1000, 0x2000, 0x4000, 0x8000, 0x8000, 0x10000]))
        b = {k: v[None] for k, v in s.items() if isinstance(v, np.ndarray)}
        if kind in ('split', 'shared') and rng.random() < 0.4:  # a batch of scenes of the same sizes
            B = int(rng.integers(2, 4))
            F = b['faces'].shape[1]
            bs = scenes.batch_scene(F, H, W, C, [seed_ + i for i in range(B)], r_lo=0.01, r_hi=0.3, shared=(kind == 'shared'))
            b = {k: bs[k] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
        want = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
        use_state = rng.random() < 0.5  # the autograd path: the forward keeps its state, the backward consumes it
        got = ops._op_rasterise(t(b['background']), t(b['vertices']), t(b['vertex_colors']), t(b['faces']), H, W, C, flags=flags & ~1,
                                keep_state=use_state)
        got, state = got if use_state else (got, None)
        tag = (kind, b['vertices'].shape[0], H, W, C, seed_, hex(flags), use_state)
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), ('forward', tag)
        ow = oracle.backward(b['vertices'], b['faces'], want, b['grad_pixels'], flags=flags & 1, want_mass=True)
        gb, gv, gvc, _ = ops._op_rasterise_grad(t(b['vertices']), t(b['faces']), t(want), t(b['grad_pixels']), H, W, C, flags=flags, state=state)
        assert np.array_equal(gb.cpu().numpy(), ow['grad_background']), ('grad_background', tag)
        try:
            parity.grads_close(gv, gvc, ow, str(tag), tol=parity.TIGHT_TOL)
        except AssertionError as e:
            if failures is None:
                raise
            failures.append(str(e))
        n += 1
    return n


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    t0 = time.time()
    failures = []
    n = run(budget=budget, seed=int(sys.argv[2]) if len(sys.argv) > 2 else 0, hard=len(sys.argv) > 3 and sys.argv[3] == 'hostile', failures=failures)
    for f in failures:
        print('MISMATCH', f)
    print('fuzz_parity: %d random cases in %.0f s, forward / visibility / grad_background bit-exact in all, %d gradient mismatches'
          % (n, time.time() - t0, len(failures)))
    if failures:
        sys.exit(1)


if __name__ == '__main__':
    main()
