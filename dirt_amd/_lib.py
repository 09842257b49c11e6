"""ctypes binding of libdirt_hip.so (the C ABI declared in include/dirt_hip.h).

This is the counterpart of `tf.load_op_library(.../librasterise.so)` in the reference
(dirt/rasterise_ops.py:5-10).  Unlike the reference, which swallows a load failure with a warning and
leaves `_rasterise_module = None`, a missing or unloadable library raises as soon as an op is used:
there is no CPU or eager fallback behind these entry points.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('DIRT_AMD_LIBRARY') or os.path.join(_HERE, 'libdirt_hip.so')  # override: instrumented builds (tools/)
ABI_VERSION = 4

FLAG_Q1_INTENDED = 1
FLAG_KEEP_STATE = 2
FLAG_REUSE_STATE = 4
FLAG_DENSE_FROM_STATE = 8   # backward: sum in the state's interleaved accumulators, copy out into the caller's dense tensors
FLAG_OUTPUTS_CLEARED = 0x10  # backward: the dense outputs are the ones dirt_rasterise_forward_train cleared (checked by the library)
FLAG_PROFILE = 0x100
FLAG_TILES_LARGE = 0x200
FLAG_TILES_SMALL = 0x400
FLAG_SHARED_FACES = 0x800
FLAG_GRAD_ROWS = 0x1000    # gradient kernel: every 8x8 block walks its own faces (default for small frames)
FLAG_GRAD_PAIRS = 0x2000   # ... or pairs of blocks share a face (default otherwise)
FLAG_GRAD_SMALL = 0x4000   # ... or the one-pixel-per-lane kernel on 16x16 tiles (default for small frames with 1, 3 or 4 channels)
FLAG_GRAD_PX2 = 0x8000    # ... or the two-pixels-per-lane kernel on 32x16 tiles (1, 3, 4 channels)
FLAG_GRAD_PX4 = 0x10000   # ... or the four-pixels-per-lane kernel where the library would choose px2
FLAG_GRAD_STREAM = 0x20000  # ... or the streaming kernel (4 channels, whole 32x32 tiles; LDS-DMA loads under the compute)
TEX_CLAMP = 1
TEX_NEAREST = 2

E_INVALID_ARGUMENT = -1
E_TOO_MANY_VERTICES = -2
E_WORKSPACE = -3
E_HIP = -4

_lib = None

# every symbol include/dirt_hip.h declares (tests/test_boundary.py checks header <-> library)
SYMBOLS = ('dirt_abi_version', 'dirt_last_error', 'dirt_workspace_bytes', 'dirt_rasterise_forward', 'dirt_rasterise_forward_train',
           'dirt_rasterise_backward', 'dirt_rasterise_visibility', 'dirt_state_grad_buffers', 'dirt_profile_count',
           'dirt_profile_name',
           'dirt_profile_read', 'dirt_profile_reset', 'dirt_texture_sample_forward', 'dirt_texture_sample_backward',
           'dirt_texture_sample_backward_image', 'dirt_texture_last_error')


class DirtLibraryError(RuntimeError):
    pass


def load():
    """dlopen libdirt_hip.so and declare the prototypes.  Raises DirtLibraryError if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DirtLibraryError(
            'libdirt_hip.so is missing (%s): build it with `python -m dirt_amd.build` (hipcc, gfx950). '
            'dirt_amd has no fallback path.' % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise DirtLibraryError('failed to load %s: %s' % (LIB_PATH, e))
    vp, fp, ip = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p  # raw device addresses
    i, sz, u = ctypes.c_int, ctypes.c_size_t, ctypes.c_uint
    lib.dirt_abi_version.restype = i
    lib.dirt_last_error.restype = ctypes.c_char_p
    lib.dirt_workspace_bytes.argtypes = [i] * 6
    lib.dirt_workspace_bytes.restype = sz
    lib.dirt_rasterise_forward.argtypes = [fp, fp, fp, ip, fp, i, i, i, i, i, i, vp, sz, u, vp]
    lib.dirt_rasterise_forward.restype = i
    lib.dirt_rasterise_forward_train.argtypes = [fp, fp, fp, ip, fp, fp, fp, i, i, i, i, i, i, vp, sz, u, vp]
    lib.dirt_rasterise_forward_train.restype = i
    lib.dirt_rasterise_backward.argtypes = [fp, ip, fp, fp, fp, fp, fp, fp, i, i, i, i, i, i, vp, sz, u, vp]
    lib.dirt_rasterise_backward.restype = i
    lib.dirt_rasterise_visibility.argtypes = [fp, ip, ip, i, i, i, i, i, vp, sz, u, vp]
    lib.dirt_rasterise_visibility.restype = i
    lib.dirt_state_grad_buffers.argtypes = [vp, sz, i, i, i, i, i, i, ctypes.POINTER(vp), ctypes.POINTER(vp),
                                            ctypes.POINTER(i), ctypes.POINTER(i)]
    lib.dirt_state_grad_buffers.restype = i
    lib.dirt_profile_count.restype = i
    lib.dirt_profile_name.argtypes = [i]
    lib.dirt_profile_name.restype = ctypes.c_char_p
    lib.dirt_profile_read.argtypes = [i, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong)]
    lib.dirt_profile_read.restype = i
    lib.dirt_profile_reset.restype = i
    ll = ctypes.c_longlong
    lib.dirt_texture_sample_forward.argtypes = [fp, fp, fp, ll, i, i, i, i, u, vp]
    lib.dirt_texture_sample_forward.restype = i
    lib.dirt_texture_sample_backward.argtypes = [fp, fp, fp, fp, fp, ll, i, i, i, i, i, u, vp]
    lib.dirt_texture_sample_backward.restype = i
    override = bool(os.environ.get('DIRT_AMD_LIBRARY'))   # an A/B build of another round (tools/): older ABIs are let through
    if hasattr(lib, 'dirt_texture_sample_backward_image') or not override:
        lib.dirt_texture_sample_backward_image.argtypes = [fp, fp, fp, fp, fp, ll, ll, i, i, i, i, i, u, vp]
        lib.dirt_texture_sample_backward_image.restype = i
    lib.dirt_texture_last_error.restype = ctypes.c_char_p
    if lib.dirt_abi_version() != ABI_VERSION and not override:
        raise DirtLibraryError('libdirt_hip.so ABI %d != expected %d' % (lib.dirt_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def last_error():
    return load().dirt_last_error().decode('utf-8', 'replace')


def check(rc):
    """Map a C-ABI return code to the exception the reference raises for the same condition:
    errors::InvalidArgument -> ValueError (csrc/rasterise_egl.cpp:302-316); HIP failures, which the
    reference turns into LOG(FATAL), -> RuntimeError."""
    if rc == 0:
        return
    msg = last_error()
    if rc in (E_INVALID_ARGUMENT, E_TOO_MANY_VERTICES, E_WORKSPACE):
        raise ValueError(msg)
    raise RuntimeError(msg)


def profile_reset():
    check(load().dirt_profile_reset())


def profile_read():
    """{kernel name: (total_ms, launches)} for calls made with FLAG_PROFILE on this thread."""
    lib = load()
    out = {}
    for slot in range(lib.dirt_profile_count()):
        ms, n = ctypes.c_double(0), ctypes.c_longlong(0)
        check(lib.dirt_profile_read(slot, ctypes.byref(ms), ctypes.byref(n)))
        out[lib.dirt_profile_name(slot).decode()] = (ms.value, n.value)
    return out
