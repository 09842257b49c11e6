"""Writes tests/golden/helpers_ref.npz: outputs of the REFERENCE'S OWN pure-Python helpers -- dirt/matrices.py,
dirt/lighting.py, dirt/projection.py, imported from where they lie under /root/reference -- on seeded inputs.

TensorFlow is not in this image: the three files are imported with oracle/tf_shim first on sys.path, a numpy stand-in for
the ~35 TensorFlow functions they call (eager float32 arrays; see its docstring).  The functions executed are the
reference's, line for line; what is not the reference's is the arithmetic underneath tf.matmul / tf.norm / ... (numpy's).
tests/test_helpers_ref.py compares the torch counterparts in dirt_amd/ with these vectors (they travel to the GPU box,
/root/reference does not), and with the live modules where the reference is present.

Run from the repository root:  python -m tests.golden.make_helpers_golden
"""
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = '/root/reference/dirt'


def available():
    return all(os.path.exists(os.path.join(REF, n + '.py')) for n in ('matrices', 'lighting', 'projection'))


def load_reference_helpers():
    """-> (matrices, lighting, projection): the reference's modules over the numpy TensorFlow stand-in."""
    shim = os.path.join(ROOT, 'oracle', 'tf_shim')
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'tensorflow' or k.startswith('tensorflow.')}
    sys.path.insert(0, shim)
    try:
        mods = []
        for name in ('matrices', 'lighting', 'projection'):
            spec = importlib.util.spec_from_file_location('dirt_reference_' + name, os.path.join(REF, name + '.py'))
            m = importlib.util.module_from_spec(spec)
            old, sys.dont_write_bytecode = sys.dont_write_bytecode, True   # no __pycache__ under /root/reference (read-only by policy)
            try:
                spec.loader.exec_module(m)
            finally:
                sys.dont_write_bytecode = old
            mods.append(m)
        return tuple(mods)
    finally:
        sys.path.remove(shim)
        for k in [k for k in sys.modules if k == 'tensorflow' or k.startswith('tensorflow.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def load_reference_texture_functions():
    """-> (uvs_to_pixel_indices, sample_texture, to_tensor) of /root/reference/samples/textured.py:16-61 over the numpy TensorFlow stand-in.
    The sample is a script (it imports cv2 and renders at import time), so only these two function definitions are
    compiled, from the file where it lies, into a namespace whose `tf` is the stand-in."""
    import ast
    path = '/root/reference/samples/textured.py'
    tree = ast.parse(open(path).read(), path)
    wanted = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ('uvs_to_pixel_indices', 'sample_texture')]
    assert len(wanted) == 2
    shim = os.path.join(ROOT, 'oracle', 'tf_shim')
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'tensorflow' or k.startswith('tensorflow.')}
    sys.path.insert(0, shim)
    try:
        import tensorflow as tf_shim
        ns = {'tf': tf_shim}
        exec(compile(ast.Module(body=wanted, type_ignores=[]), path, 'exec'), ns)
        return ns['uvs_to_pixel_indices'], ns['sample_texture'], tf_shim.convert_to_tensor
    finally:
        sys.path.remove(shim)
        for k in [k for k in sys.modules if k == 'tensorflow' or k.startswith('tensorflow.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def texture_inputs():
    rng = np.random.default_rng(7)
    return {'texture': rng.uniform(0, 1, (9, 13, 3)).astype(np.float32),
            'uvs': rng.uniform(-1.5, 2.5, (6, 11, 2)).astype(np.float32)}


def evaluate_texture(uvs_to_pixel_indices, sample_texture, x, to_np, wrap=lambda a: a):
    out = {}
    for mode in ('repeat', 'clamp'):
        idx = uvs_to_pixel_indices(wrap(x['uvs']), list(x['texture'].shape[:2]), mode)
        out['indices_' + mode] = to_np(idx)
        for filt in ('bilinear', 'nearest'):
            out['sample_%s_%s' % (mode, filt)] = to_np(sample_texture(wrap(x['texture']), idx, filt))
    return out


def inputs():
    """Seeded inputs, shared with the tests."""
    rng = np.random.default_rng(42)
    f32 = lambda a: np.asarray(a, np.float32)
    faces = np.array([[0, 1, 2], [2, 3, 4], [4, 5, 6], [0, 2, 6], [1, 3, 5]], np.int32)
    return {
        'vectors': f32(rng.standard_normal((3, 2, 3))), 'vector_zero': f32(np.zeros(3)),
        'translation': f32(rng.standard_normal((4, 3))), 'scale': f32(rng.uniform(0.5, 2., (2, 3))),
        'near': f32([0.1, 0.5]), 'far': f32([20., 10.]), 'right': f32([0.2, 0.4]), 'aspect': f32([0.75, 1.0]),
        'mat3': f32(rng.standard_normal((2, 3, 3))),
        'verts': f32(rng.standard_normal((2, 7, 3))), 'verts4': f32(rng.standard_normal((7, 4))), 'faces': faces,
        'colors': f32(rng.uniform(0, 1, (2, 7, 3))), 'light_dir': f32([[0.6, 0., -0.8], [0., 1., 0.]]),
        'light_col': f32([[1., 0.9, 0.8], [0.2, 0.3, 0.4]]), 'camera': f32(rng.standard_normal((2, 3))), 'shininess': f32([4., 9.]),
        'light_pos': f32(rng.standard_normal((2, 3)) * 3),
        'pixel_locations': f32(rng.uniform(0, 60, (2, 5, 2))), 'image_size': np.array([[64, 48], [32, 32]], np.int32),
        'clip_to_world': f32(np.eye(4) + 0.1 * rng.standard_normal((2, 4, 4))),
    }


def evaluate(matrices, lighting, projection, x, to_np):
    """Every helper the torch side provides, on the shared inputs; `to_np` turns a module's tensor into numpy."""
    out = {}
    out['rodrigues'] = to_np(matrices.rodrigues(x['vectors']))
    out['rodrigues_3x3'] = to_np(matrices.rodrigues(x['vectors'], three_by_three=True))
    out['rodrigues_zero'] = to_np(matrices.rodrigues(x['vector_zero']))
    out['translation'] = to_np(matrices.translation(x['translation']))
    out['scale'] = to_np(matrices.scale(x['scale']))
    out['perspective'] = to_np(matrices.perspective_projection(x['near'], x['far'], x['right'], x['aspect']))
    out['perspective_scalar'] = to_np(matrices.perspective_projection(0.1, 20., 0.2, 0.75))
    out['pad'] = to_np(matrices.pad_3x3_to_4x4(x['mat3']))
    out['compose'] = to_np(matrices.compose(matrices.rodrigues(x['vectors'][0]), matrices.translation(x['translation'][:2]),
                                            matrices.scale(x['scale'])))
    out['vertex_normals'] = to_np(lighting.vertex_normals(x['verts'], x['faces']))
    out['vertex_normals_single_w'] = to_np(lighting.vertex_normals(x['verts4'], x['faces']))
    nv, nf = lighting.split_vertices_by_face(x['verts'], x['faces'])
    out['split_vertices'], out['split_faces'] = to_np(nv), to_np(nf)
    out['vertex_normals_pre_split'] = to_np(lighting.vertex_normals_pre_split(nv, nf))
    normals = lighting.vertex_normals(x['verts'], x['faces'])
    for ds in (True, False):
        tag = '_double' if ds else '_single'
        out['diffuse_directional' + tag] = to_np(lighting.diffuse_directional(normals, x['colors'], x['light_dir'], x['light_col'], double_sided=ds))
        # (the reference's specular_directional adds light_direction [*, 3] to [*, V, 3] without a new axis: it only
        # broadcasts for an unbatched direction, which is how its samples call it)
        out['specular_directional' + tag] = to_np(lighting.specular_directional(x['verts'], normals, x['colors'], x['light_dir'][0], x['light_col'],
                                                                               x['camera'], x['shininess'], double_sided=ds))
        out['diffuse_point' + tag] = to_np(lighting.diffuse_point(x['verts'], normals, x['colors'], x['light_pos'], x['light_col'], double_sided=ds))
    starts, deltas = projection.unproject_pixels_to_rays(x['pixel_locations'], x['clip_to_world'], x['image_size'])
    out['ray_starts'], out['ray_deltas'] = to_np(starts), to_np(deltas)
    return out


def main():
    m, l, p = load_reference_helpers()
    out = evaluate(m, l, p, inputs(), lambda t: np.asarray(t))
    f1, f2, to_tensor = load_reference_texture_functions()
    for k, v in evaluate_texture(f1, f2, texture_inputs(), lambda t: np.asarray(t), to_tensor).items():
        out['texture/' + k] = v
    path = os.path.join(HERE, 'helpers_ref.npz')
    np.savez_compressed(path, **out)
    print('helpers_ref.npz: %d arrays, %d bytes' % (len(out), os.path.getsize(path)))


if __name__ == '__main__':
    main()
