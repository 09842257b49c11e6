"""Writes the golden fixtures tests/golden/*.npz.

The reference's ops cannot be imported or run in this container (TensorFlow, CUDA and NVIDIA EGL are
absent; its kernels are DEVICE_GPU only) and it ships no stored expected arrays.  Two kinds of vectors:
  * <case>.npz: the CPU oracle's forward, visibility and backward (with the per-element L1 mass the
    parity tolerance refers to), pinned by tests/test_oracle.py;
  * ref_grads.npz (`--ref`, needs /root/reference): for the same cases, the output of the REFERENCE'S OWN
    `assemble_grads` / `launch_grad_assembly` compiled for the host (oracle/make_ref.py), fed the oracle's
    visibility surfaces, and `pixels` = the reference's own upload_background / download_pixels around the oracle's
    flip-free GL draw.  tests/test_oracle_ref.py requires the oracle to reproduce these bit for bit
    wherever the tests run (the reference itself does not travel to the GPU box).

Run from the repository root:  python -m tests.golden.make_golden [--ref]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests import scenes  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

# name: (kind, args)
CASES = {
    'square_k1': ('square',),
    'cube_k2_64': ('cube', 64, 64),
    'rand_c3_48x36': ('rand', 150, 36, 48, 3, [3], 0.05, 0.3, False),       # the frame of tests/rasterise_tests.py:55-56
    'rand_c4_batch2': ('rand', 300, 64, 80, 4, [11, 12], 0.03, 0.2, False),  # groups [0:3],[3:4], Q1 across scenes
    'shared_c1': ('rand', 400, 72, 56, 1, [13], 0.0, 0.0, True),
    'cylinder_48x36': ('cylinder',),             # the scene of the reference's tests/rasterise_tests.py:50-99,115
    'cylinder_batch2': ('cylinder_batch',),      # ... and its batch of two (:89,123-132)
    'bent_square_gbuffer': ('bent_square',),     # the 7-channel G-buffer of tests/deferred_grad_test.py:121-142
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
    if kind == 'cylinder':
        s = scenes.cylinder_scene()
        return {k: s[k][None] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    if kind == 'cylinder_batch':
        s = scenes.cylinder_batch_scene()
        return {k: s[k] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    if kind == 'bent_square':
        # vertex attributes of get_pixels_deferred_v2 (tests/deferred_grad_test.py:123-124): mask 1, colours 3, normals 3
        clip, faces, world, colours = scenes.bent_square_geometry()
        tri = world[:, :3].reshape(-1, 3, 3)
        n = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        attrs = np.concatenate([np.ones([6, 1]), colours, np.repeat(n, 3, axis=0)], axis=1).astype(np.float32)
        g = np.random.default_rng(5).standard_normal([1, 32, 32, 7]).astype(np.float32)
        return {'background': np.zeros([1, 32, 32, 7], np.float32), 'vertices': clip[None], 'vertex_colors': attrs[None],
                'faces': faces[None], 'grad_pixels': g}
    _, F, H, W, C, seeds, rlo, rhi, shared = case
    b = scenes.batch_scene(F, H, W, C, seeds, r_lo=rlo, r_hi=rhi, shared=shared)
    return {k: b[k] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}


def main():
    import oracle
    for name, case in CASES.items():
        s = make_inputs(case)
        px = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
        out = oracle.backward(s['vertices'], s['faces'], px, s['grad_pixels'], want_mass=True)
        fid = np.stack([oracle.visibility(s['vertices'][i], s['faces'][i], px.shape[1], px.shape[2])[0]
                        for i in range(px.shape[0])])
        path = os.path.join(HERE, name + '.npz')
        np.savez_compressed(path, pixels=px, face_id=fid, **out)
        print('%-18s %s  %d bytes' % (name, px.shape, os.path.getsize(path)))


def main_ref():
    import oracle
    from oracle import ref
    out = {}
    for name, case in CASES.items():
        s = make_inputs(case)
        px = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
        r = ref.backward(s['vertices'], s['faces'], px, s['grad_pixels'])
        for k, v in r.items():
            out[name + '/' + k] = v
        # forward: the reference's own upload_background / download_pixels around the oracle's flip-free GL draw
        out[name + '/pixels'] = ref.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    path = os.path.join(HERE, 'ref_grads.npz')
    np.savez_compressed(path, **out)
    print('ref_grads.npz: %d arrays, %d bytes' % (len(out), os.path.getsize(path)))


if __name__ == '__main__':
    main_ref() if '--ref' in sys.argv else main()
