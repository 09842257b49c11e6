"""Writes the golden fixtures tests/golden/*.npz.

The reference cannot be imported or run in this container (TensorFlow, CUDA and NVIDIA EGL are absent;
its kernels are DEVICE_GPU only) and it ships no stored expected arrays, so these vectors come from the
CPU oracle, which is itself pinned by tests/test_oracle.py (square_test known answer, exact-arithmetic
coverage, analytic interpolation, numpy restatement of assemble_grads).  They let the GPU box compare
the HIP path with committed numbers, and they detect drift of the oracle.

Run from the repository root:  python -m tests.golden.make_golden
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from dirt_amd import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name: (kind, args)
CASES = {
    'square_k1': ('square',),
    'cube_k2_64': ('cube', 64, 64),
    'rand_c3_48x36': ('rand', 150, 36, 48, 3, [3], 0.05, 0.3, False),       # the frame of tests/rasterise_tests.py:55-56
    'rand_c4_batch2': ('rand', 300, 64, 80, 4, [11, 12], 0.03, 0.2, False),  # groups [0:3],[3:4], Q1 across scenes
    'shared_c1': ('rand', 400, 72, 56, 1, [13], 0.0, 0.0, True),
}


def make_inputs(case):
    kind = case[0]
    if kind == 'square':
        s = scenes.square_scene()
        s['grad_pixels'] = np.random.default_rng(0).standard_normal(s['background'].shape).astype(np.float32)
        return {k: s[k][None] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    if kind == 'cube':
        s = scenes.cube_scene(case[1], case[2])
        s['background'] = np.random.default_rng(1).uniform(0, 1, s['background'].shape).astype(np.float32)
        s['grad_pixels'] = np.random.default_rng(2).standard_normal(s['background'].shape).astype(np.float32)
        return {k: s[k][None] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    _, F, H, W, C, seeds, rlo, rhi, shared = case
    b = scenes.batch_scene(F, H, W, C, seeds, r_lo=rlo, r_hi=rhi, shared=shared)
    return {k: b[k] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}


def main():
    import oracle
    for name, case in CASES.items():
        s = make_inputs(case)
        px = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
        out = oracle.backward(s['vertices'], s['faces'], px, s['grad_pixels'])
        fid = np.stack([oracle.visibility(s['vertices'][i], s['faces'][i], px.shape[1], px.shape[2])[0]
                        for i in range(px.shape[0])])
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, pixels=px, face_id=fid, **out)
        print('%-18s %s  %d bytes' % (name, px.shape, os.path.getsize(path)))


if __name__ == '__main__':
    main()
