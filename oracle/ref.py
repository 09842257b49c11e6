"""ctypes/numpy front-end of oracle/_ref/libdirt_ref.so -- the reference's OWN `assemble_grads`
(csrc/rasterise_grad_egl.cu:93-236) and `launch_grad_assembly` (:238-278) compiled for the host
(oracle/make_ref.py, oracle/ref_shim/).  TEST INFRASTRUCTURE: pins oracle/dirt_oracle.c's backward
restatement to the reference's kernel source.

What is the reference's and what is not:
  * every gradient operation, the zeroing of the outputs, the atlas indexing (`iib`, the vertical
    flip), quirk Q1's aliasing reads: the reference's code, executed here;
  * the two surfaces the kernel reads are rendered by NVIDIA's OpenGL driver in the reference
    (csrc/rasterise_grad_egl.cpp:432-456, csrc/shaders.cpp:45-79); here they are filled from the
    oracle's specification-pinned visibility (`oracle.visibility`), over the reference's clear values
    (csrc/rasterise_grad_egl.cpp:442-445), in the reference's atlas layout (:408-428);
  * the channel-group loops of `forward` / `backward` below are dirt/rasterise_ops.py:86-108,132-177 restated over numpy
    (they also run where /root/reference is absent); `python_layer()` is the reference's rasterise_ops.py ITSELF, imported
    over a numpy stand-in for TensorFlow and bound to the kernels above -- tests/test_oracle_ref.py requires both routes,
    and the reference's deferred wrapper with its gradient closure, to equal the oracle bit for bit.
"""
import ctypes
import math
import numpy as np


import contextlib as _contextlib


@_contextlib.contextmanager
def no_bytecode():
    """Importing a module from /root/reference must not write __pycache__ there (the reference tree is read-only by policy)."""
    import sys
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        yield
    finally:
        sys.dont_write_bytecode = old

from . import make_ref
from . import oracle as _oracle

_lib = None


def available():
    return make_ref.build() is not None


def _load():
    global _lib
    if _lib is None:
        so = make_ref.build()
        if so is None:
            raise RuntimeError('oracle/_ref is not built and /root/reference is not present')
        lib = ctypes.CDLL(so)
        fp = ctypes.POINTER(ctypes.c_float)
        i = ctypes.c_int
        lib.dirt_ref_rasterise_grad.argtypes = [fp] * 9 + [i] * 7
        lib.dirt_ref_rasterise_grad.restype = i
        lib.dirt_ref_upload_vertices.argtypes = [fp, ctypes.POINTER(ctypes.c_int32), ctypes.c_void_p, i, i, i]
        lib.dirt_ref_upload_vertices.restype = i
        lib.dirt_ref_upload_background.argtypes = [fp, fp] + [i] * 6
        lib.dirt_ref_upload_background.restype = i
        lib.dirt_ref_download_pixels.argtypes = [fp, fp] + [i] * 6
        lib.dirt_ref_download_pixels.restype = i
        _lib = lib
    return _lib


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def atlas_shape(batch_size, height, width):
    """Framebuffer size RasteriseGradOpGpu::Compute chooses for a batch, csrc/rasterise_grad_egl.cpp:408-414."""
    horizontal_count = int(math.sqrt(np.float32(batch_size)) + np.float32(.1))
    vertical_count = batch_size // horizontal_count + (0 if batch_size % horizontal_count == 0 else 1)
    return height * vertical_count, width * horizontal_count


def surfaces(vertices, faces, height, width):
    """The two RGBA32F colour attachments after the backward render of a batch:
    (barycentrics_and_depth, indices), each [buffer_height, buffer_width, 4], GL orientation."""
    B = vertices.shape[0]
    bh, bw = atlas_shape(B, height, width)
    bary_w = np.empty((bh, bw, 4), np.float32)
    bary_w[..., :3] = -1.
    bary_w[..., 3] = np.inf  # clear values, csrc/rasterise_grad_egl.cpp:442-445
    index = np.full((bh, bw, 4), -1., np.float32)
    frames_per_row = bw // width
    for ib in range(B):
        fid, bary, cw = _oracle.visibility(vertices[ib], faces[ib], height, width)
        frame_x = (ib % frames_per_row) * width     # glViewport, csrc/rasterise_grad_egl.cpp:434-437
        frame_y = (ib // frames_per_row) * height
        covered = fid >= 0
        tri = np.asarray(faces[ib], np.int64)[np.where(covered, fid, 0)].astype(np.float32)  # flat ivec3 -> float, shaders.cpp:53,76
        tile_b = np.concatenate([bary, cw[..., None]], -1)
        tile_i = np.where(covered[..., None], np.concatenate([tri, np.full(fid.shape + (1,), -1., np.float32)], -1), np.float32(-1.))
        bary_w[frame_y:frame_y + height, frame_x:frame_x + width] = tile_b[::-1]  # tensor rows are top-first
        index[frame_y:frame_y + height, frame_x:frame_x + width] = tile_i[::-1]
    return bary_w, index


def rasterise_grad_op(vertices, faces, pixels, grad_pixels, surf=None):
    """One `RasteriseGrad` op call, channels in {1, 3} -> dict like the op's namedtuple (dirt/rasterise_ops.py:113-128)."""
    lib = _load()
    vertices = np.ascontiguousarray(vertices, np.float32)
    faces = np.ascontiguousarray(faces, np.int32)
    B, H, W, C = pixels.shape
    V = vertices.shape[1]
    assert C in (1, 3) and grad_pixels.shape == pixels.shape and vertices.shape == (B, V, 4)
    if V > (1 << 24):
        raise ValueError('RasteriseGrad supports a maximum of %d vertices' % (1 << 24))  # csrc/rasterise_grad_egl.cpp:399-405

    def padded(a):
        # Quirk Q1: for C == 1 the kernel reads two floats past each pixel, so past the end of the
        # tensor for the last two pixels (undefined in the reference).  The oracle defines those
        # reads as the last element; pad accordingly.
        flat = np.ascontiguousarray(a, np.float32).reshape(-1)
        return np.concatenate([flat, np.repeat(flat[-1:], 2)])

    pix, gpix = padded(pixels), padded(grad_pixels)
    bary_w, index = surf if surf is not None else surfaces(vertices, faces, H, W)
    bh, bw = bary_w.shape[:2]
    out = {'grad_background': np.full((B, H, W, C), np.nan, np.float32),
           'grad_vertices': np.full((B, V, 4), np.nan, np.float32),
           'grad_vertex_colors': np.full((B, V, C), np.nan, np.float32),
           'debug_thingy': np.full((B, H, W, 3), np.nan, np.float32)}
    rc = lib.dirt_ref_rasterise_grad(_fp(vertices), _fp(pix), _fp(gpix), _fp(bary_w), _fp(index),
                                     _fp(out['grad_background']), _fp(out['grad_vertices']), _fp(out['grad_vertex_colors']),
                                     _fp(out['debug_thingy']), B, V, H, W, C, bw, bh)
    if rc != 0:
        raise ValueError('dirt_ref_rasterise_grad failed: %d' % rc)
    return out


def backward(vertices, faces, pixels, grad_pixels):
    """`_rasterise_grad_multichannel(..., 'batch')`, dirt/rasterise_ops.py:132-177: groups of three
    channels while three remain, then singles; grad_vertices summed, the others concatenated.
    debug_thingy is the first group's (what oracle.backward exports)."""
    pixels = np.asarray(pixels, np.float32)
    grad_pixels = np.asarray(grad_pixels, np.float32)
    vertices = np.ascontiguousarray(vertices, np.float32)
    faces = np.ascontiguousarray(faces, np.int32)
    channels = pixels.shape[3]
    surf = surfaces(vertices, faces, pixels.shape[1], pixels.shape[2])
    results = []
    begin_channel = 0
    while begin_channel < channels:
        end_channel = begin_channel + 3 if begin_channel + 3 <= channels else begin_channel + 1
        results.append(rasterise_grad_op(vertices, faces, pixels[..., begin_channel:end_channel],
                                         grad_pixels[..., begin_channel:end_channel], surf))
        begin_channel = end_channel
    grad_vertices = results[0]['grad_vertices']
    for result in results[1:]:
        grad_vertices = grad_vertices + result['grad_vertices']  # fp32 sum over groups, as tf's `sum`
    return {'grad_vertices': grad_vertices,
            'grad_vertex_colors': np.concatenate([r['grad_vertex_colors'] for r in results], -1),
            'grad_background': np.concatenate([r['grad_background'] for r in results], -1),
            'debug_thingy': results[0]['debug_thingy']}


def rasterise_op(background, vertices, vertex_colors, faces):
    """One `Rasterise` op call, channels in {1, 3}, as RasteriseOpGpu::Compute runs it (csrc/rasterise_egl.cpp:276-407):
    the reference's OWN upload_background (background -> framebuffer atlas: vertical flip, scene tiling, C = 1 replicated),
    a GL draw per scene with the viewport of :362-369, the reference's OWN download_pixels.  The draw itself is the
    oracle's flip-free `draw_gl`: the GL pipeline is the one part of the op that is not the reference's source."""
    lib = _load()
    background = np.ascontiguousarray(background, np.float32)
    vertices = np.ascontiguousarray(vertices, np.float32)
    vertex_colors = np.ascontiguousarray(vertex_colors, np.float32)
    faces = np.ascontiguousarray(faces, np.int32)
    B, H, W, C = background.shape
    assert C in (1, 3)
    bh, bw = atlas_shape(B, H, W)   # the forward op sizes its atlas the same way (csrc/rasterise_egl.cpp:326-334)
    atlas = np.full((bh, bw, 4), np.nan, np.float32)
    rc = lib.dirt_ref_upload_background(_fp(background), _fp(atlas), B, H, W, C, bw, bh)
    if rc != 0:
        raise ValueError('dirt_ref_upload_background failed: %d' % rc)
    frames_per_row = bw // W
    for ib in range(B):
        _oracle.draw_gl(vertices[ib], vertex_colors[ib], faces[ib], atlas, H, W, (ib % frames_per_row) * W, (ib // frames_per_row) * H)
    pixels = np.full((B, H, W, C), np.nan, np.float32)
    rc = lib.dirt_ref_download_pixels(_fp(atlas), _fp(pixels), B, H, W, C, bw, bh)
    if rc != 0:
        raise ValueError('dirt_ref_download_pixels failed: %d' % rc)
    return pixels


def forward(background, vertices, vertex_colors, faces):
    """`rasterise_batch`'s channel grouping, dirt/rasterise_ops.py:86-108: one op for 1 or 3 channels, else groups of three
    while three remain, then singles, concatenated."""
    background = np.asarray(background, np.float32)
    vertex_colors = np.asarray(vertex_colors, np.float32)
    channels = background.shape[3]
    if channels in (1, 3):
        return rasterise_op(background, vertices, vertex_colors, faces)
    pixels, begin_channel = [], 0
    while begin_channel < channels:
        end_channel = begin_channel + 3 if begin_channel + 3 <= channels else begin_channel + 1
        pixels.append(rasterise_op(background[..., begin_channel:end_channel], vertices, vertex_colors[..., begin_channel:end_channel], faces))
        begin_channel = end_channel
    return np.concatenate(pixels, -1)


_python_layer = None


def _op_library_for(tf_shim):
    """The attributes of the module tf.load_op_library returns (dirt/rasterise_ops.py:81-85,113-118), on the host-compiled kernels."""
    import collections
    result = collections.namedtuple('RasteriseGrad', ['grad_background', 'grad_vertices', 'grad_vertex_colors', 'debug_thingy'])

    class OpLibrary:
        @staticmethod
        def rasterise(background, vertices, vertex_colors, faces, height, width, channels, name=None):
            b = np.asarray(background)
            assert b.shape[1:] == (height, width, channels)
            return tf_shim.convert_to_tensor(rasterise_op(b, np.asarray(vertices), np.asarray(vertex_colors), np.asarray(faces)))

        @staticmethod
        def rasterise_grad(vertices, faces, pixels, grad_pixels, height, width, channels, name=None):
            out = rasterise_grad_op(np.asarray(vertices), np.asarray(faces), np.asarray(pixels), np.asarray(grad_pixels))
            return result(*[tf_shim.convert_to_tensor(out[k]) for k in result._fields])

    return OpLibrary


def run_reference_script(path, entry='main'):
    """Run one of the reference's own scripts (e.g. /root/reference/tests/square_test.py) VERBATIM: `import tensorflow` gives
    the numpy stand-in, `import dirt` the reference's own package (dirt/__init__.py, dirt/rasterise_ops.py) with its op
    library bound to the host-compiled kernels and the oracle's GL draw; `import cv2` (the samples display their result)
    gives a stub that records what is shown, and tf.write_file records instead of writing.  Returns what the script printed;
    `run_reference_script.images` then holds the (window or file name, image) pairs it showed or wrote."""
    import contextlib
    import importlib.util
    import io
    import os
    import sys
    shim = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tf_shim')
    import types
    saved = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k in ('tensorflow', 'dirt', 'cv2') or k.startswith('tensorflow.') or k.startswith('dirt.')}
    sys.path[:0] = [shim, '/root/reference']
    run_reference_script.images = shown = []
    cv2 = types.ModuleType('cv2')
    cv2.imshow = lambda name, image: shown.append((name, np.array(image)))
    cv2.waitKey = lambda *a: 0
    cv2.imwrite = lambda name, image: shown.append((name, np.array(image))) or True
    try:
        sys.modules['cv2'] = cv2
        import tensorflow as tf_shim
        tf_shim._op_library = _op_library_for(tf_shim)
        tf_shim.written_files = shown   # tf.write_file(name, tf.image.encode_jpeg(uint8 image)) lands in the same list
        spec = importlib.util.spec_from_file_location('dirt_reference_script', path)
        module = importlib.util.module_from_spec(spec)
        out = io.StringIO()
        with contextlib.redirect_stdout(out), no_bytecode():
            spec.loader.exec_module(module)
            getattr(module, entry)()
        return out.getvalue()
    finally:
        del sys.path[:2]
        for k in [k for k in sys.modules if k in ('tensorflow', 'dirt', 'cv2') or k.startswith('tensorflow.') or k.startswith('dirt.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def python_layer():
    """The reference's OWN Python op layer -- /root/reference/dirt/rasterise_ops.py, imported from where it lies over the
    numpy TensorFlow stand-in (oracle/tf_shim) -- with its `_rasterise_module` bound to the host-compiled reference kernels
    above: `.rasterise` = rasterise_op (reference upload / download around the oracle's flip-free GL draw), `.rasterise_grad` =
    rasterise_grad_op (the reference's assemble_grads on the oracle's surfaces).  `rasterise`, `rasterise_batch`,
    `_rasterise_grad_multichannel` and the forward half of the deferred wrappers then run as the reference wrote them:
    dtype coercion, channel grouping, concatenation, the float32 sum of grad_vertices over groups."""
    global _python_layer
    if _python_layer is not None:
        return _python_layer
    import importlib.util
    import os
    import sys
    path = '/root/reference/dirt/rasterise_ops.py'
    if not os.path.exists(path):
        raise RuntimeError('/root/reference is not present')
    shim = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tf_shim')
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'tensorflow' or k.startswith('tensorflow.')}
    sys.path.insert(0, shim)
    try:
        import tensorflow as tf_shim
        OpLibrary = _op_library_for(tf_shim)
        tf_shim._op_library = OpLibrary
        spec = importlib.util.spec_from_file_location('dirt_reference_rasterise_ops', path)
        module = importlib.util.module_from_spec(spec)
        with no_bytecode():
            spec.loader.exec_module(module)
        assert module._rasterise_module is OpLibrary
        module.tf_shim = tf_shim
        _python_layer = module
        return module
    finally:
        sys.path.remove(shim)
        for k in [k for k in sys.modules if k == 'tensorflow' or k.startswith('tensorflow.')]:
            del sys.modules[k]
        sys.modules.update(saved)


def upload_vertices(vertices, faces):
    """The expanded vertex buffer of the backward render (csrc/rasterise_grad_egl.cu:11-33):
    -> structured array [B, 3F] of (position[4], barycentric[2], indices[3])."""
    lib = _load()
    vertices = np.ascontiguousarray(vertices, np.float32)
    faces = np.ascontiguousarray(faces, np.int32)
    B, V = vertices.shape[:2]
    F = faces.shape[1]
    dt = np.dtype([('position', np.float32, 4), ('barycentric', np.float32, 2), ('indices', np.int32, 3)])
    out = np.zeros((B, 3 * F), dt)
    rc = lib.dirt_ref_upload_vertices(_fp(vertices), faces.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                      out.ctypes.data_as(ctypes.c_void_p), B, V, F)
    if rc != 0:
        raise ValueError('dirt_ref_upload_vertices failed: %d' % rc)
    return out
