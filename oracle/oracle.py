"""ctypes/numpy front-end of oracle/dirt_oracle.c (TEST INFRASTRUCTURE -- see that file's header).

The function names mirror the reference's TF op module: `rasterise_batch` is the forward of
dirt/rasterise_ops.py:51-108 and `rasterise_grad` the multichannel gradient of
dirt/rasterise_ops.py:132-177, both over numpy arrays.
"""
import ctypes
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libdirt_oracle.so')
_lib = None

FLAG_Q1_INTENDED = 1
FLAG_F32_SEQUENTIAL = 2  # float32 adds in the order one thread of the reference's kernel makes them: equals oracle/_ref bit for bit


def build(force=False):
    """Compile oracle/dirt_oracle.c with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, 'dirt_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE, 'all'])
    return _SO


def _load():
    global _lib
    if _lib is not None:
        return _lib
    build()
    try:
        lib = ctypes.CDLL(_SO)
    except OSError:
        build(force=True)
        lib = ctypes.CDLL(_SO)
    fp = ctypes.POINTER(ctypes.c_float)
    ip = ctypes.POINTER(ctypes.c_int32)
    i = ctypes.c_int
    lib.dirt_oracle_forward.argtypes = [fp, fp, fp, ip, fp, i, i, i, i, i, i]
    lib.dirt_oracle_forward.restype = i
    lib.dirt_oracle_backward.argtypes = [fp, ip, fp, fp, fp, fp, fp, fp, i, i, i, i, i, i, ctypes.c_uint]
    lib.dirt_oracle_backward.restype = i
    lib.dirt_oracle_backward_ex.argtypes = [fp, ip, fp, fp, fp, fp, fp, fp, fp, fp, fp, i, i, i, i, i, i, ctypes.c_uint]
    lib.dirt_oracle_backward_ex.restype = i
    lib.dirt_oracle_visibility.argtypes = [fp, ip, ip, fp, fp, i, i, i, i]
    lib.dirt_oracle_visibility.restype = i
    lib.dirt_oracle_draw_gl.argtypes = [fp, fp, ip, fp, i, i, i, i, i, i, i, i, i]
    lib.dirt_oracle_draw_gl.restype = i
    lib.dirt_oracle_num_threads.restype = i
    lib.dirt_oracle_set_num_threads.argtypes = [i]
    _lib = lib
    return lib


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def num_threads():
    return int(_load().dirt_oracle_num_threads())


def set_num_threads(n):
    _load().dirt_oracle_set_num_threads(int(n))


def forward(background, vertices, vertex_colors, faces):
    """background [B,H,W,C], vertices [B,V,4], vertex_colors [B,V,C], faces [B,F,3] -> pixels [B,H,W,C]."""
    lib = _load()
    background, vertices, vertex_colors, faces = _f(background), _f(vertices), _f(vertex_colors), _i(faces)
    B, H, W, C = background.shape
    V, F = vertices.shape[1], faces.shape[1]
    assert vertices.shape == (B, V, 4) and vertex_colors.shape == (B, V, C) and faces.shape == (B, F, 3)
    pixels = np.empty_like(background)
    rc = lib.dirt_oracle_forward(_fp(background), _fp(vertices), _fp(vertex_colors), _ip(faces), _fp(pixels), B, V, F, H, W, C)
    if rc != 0:
        raise ValueError('dirt_oracle_forward failed: %d' % rc)
    return pixels


def backward(vertices, faces, pixels, grad_pixels, flags=0, want_debug=False, want_mass=False):
    """-> dict(grad_background [B,H,W,C], grad_vertices [B,V,4], grad_vertex_colors [B,V,C][, debug_thingy [B,H,W,3]]
    [, mass_vertices [B,V,4], mass_vertex_colors [B,V,C]]).

    `mass_*` is, per output element, the sum of |term| over everything the reference's atomics add into
    it (csrc/rasterise_grad_egl.cu:140,228-230; for `.w` the two products of :230 count separately): the
    scale of the per-element tolerance |gpu - oracle| <= 1e-4 * mass of the parity tests.  `cond_vertices` is the
    cancellation scale of the same terms (the Scharr differences and sum_k b_k * vertex_k taken with magnitudes): a term
    that is the rounding residue of an exactly cancelling difference is defined only up to a few ulps of it."""
    lib = _load()
    vertices, faces, pixels, grad_pixels = _f(vertices), _i(faces), _f(pixels), _f(grad_pixels)
    B, H, W, C = pixels.shape
    V, F = vertices.shape[1], faces.shape[1]
    assert grad_pixels.shape == pixels.shape and vertices.shape == (B, V, 4) and faces.shape == (B, F, 3)
    gb = np.empty_like(pixels)
    gv = np.empty((B, V, 4), np.float32)
    gvc = np.empty((B, V, C), np.float32)
    dbg = np.empty((B, H, W, 3), np.float32) if want_debug else None
    mv = np.empty((B, V, 4), np.float32) if want_mass else None
    mvc = np.empty((B, V, C), np.float32) if want_mass else None
    cv = np.empty((B, V, 4), np.float32) if want_mass else None
    rc = lib.dirt_oracle_backward_ex(_fp(vertices), _ip(faces), _fp(pixels), _fp(grad_pixels), _fp(gb), _fp(gv), _fp(gvc),
                                     _fp(dbg) if want_debug else None, _fp(mv) if want_mass else None,
                                     _fp(mvc) if want_mass else None, _fp(cv) if want_mass else None, B, V, F, H, W, C, flags)
    if rc != 0:
        raise ValueError('dirt_oracle_backward failed: %d' % rc)
    out = {'grad_background': gb, 'grad_vertices': gv, 'grad_vertex_colors': gvc}
    if want_debug:
        out['debug_thingy'] = dbg
    if want_mass:
        out['mass_vertices'] = mv
        out['mass_vertex_colors'] = mvc
        out['cond_vertices'] = cv
    return out


def visibility(vertices, faces, height, width):
    """One scene: vertices [V,4], faces [F,3] -> (face_id [H,W] int32, bary [H,W,3], clip_w [H,W])."""
    lib = _load()
    vertices, faces = _f(vertices), _i(faces)
    V, F = vertices.shape[0], faces.shape[0]
    fid = np.empty((height, width), np.int32)
    bary = np.empty((height, width, 3), np.float32)
    cw = np.empty((height, width), np.float32)
    rc = lib.dirt_oracle_visibility(_fp(vertices), _ip(faces), _ip(fid), _fp(bary), _fp(cw), V, F, height, width)
    if rc != 0:
        raise ValueError('dirt_oracle_visibility failed: %d' % rc)
    return fid, bary, cw


def draw_gl(vertices, vertex_colors, faces, atlas, height, width, frame_x, frame_y):
    """The GL draw of one scene into `atlas` [atlas_h, atlas_w, 4] (float32, GL orientation, modified in place) with the
    viewport (frame_x, frame_y, width, height); no notion of tensor rows (see dirt_oracle_draw_gl)."""
    lib = _load()
    vertices, vertex_colors, faces = _f(vertices), _f(vertex_colors), _i(faces)
    assert atlas.dtype == np.float32 and atlas.flags['C_CONTIGUOUS'] and atlas.ndim == 3 and atlas.shape[2] == 4
    rc = lib.dirt_oracle_draw_gl(_fp(vertices), _fp(vertex_colors), _ip(faces), _fp(atlas), vertices.shape[0], faces.shape[0],
                                 height, width, vertex_colors.shape[1], atlas.shape[1], atlas.shape[0], frame_x, frame_y)
    if rc != 0:
        raise ValueError('dirt_oracle_draw_gl failed: %d' % rc)


# Names of the reference's op module (dirt/rasterise_ops.py:81,113)
rasterise_batch = forward
rasterise_grad = backward
