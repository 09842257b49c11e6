"""`dirt.rasterise_ops` for PyTorch-ROCm tensors on MI355X.

Mirrors the reference's Python op API -- same function names, positional order, defaults, shapes
and dtype coercions (dirt/rasterise_ops.py:13-108) and the same gradient wiring
(dirt/rasterise_ops.py:111-129: gradients for [background, vertices, vertex_colors], None for
faces) -- over torch tensors.  The TF custom ops `Rasterise` / `RasteriseGrad` are replaced by the
C-ABI entry points of libdirt_hip.so (include/dirt_hip.h); tensors are handed over as raw device
pointers together with the current HIP stream.  There is no CPU implementation, as in the
reference (its kernels are DEVICE_GPU only, csrc/rasterise_egl.cpp:410): inputs must be on a GPU.

Differences from the reference, by design:
  * any `channels` >= 1 is one native call; the result equals the reference's channel-grouped
    evaluation (dirt/rasterise_ops.py:86-108,132-177), which issues one op per group;
  * the deferred wrappers rasterise visibility once per gradient call instead of once per group.
"""
import collections
import ctypes

import torch

from . import _lib

__all__ = ['rasterise', 'rasterise_batch', 'rasterise_deferred', 'rasterise_batch_deferred']

_WORKSPACE_SLOTS = 4   # scratch buffers kept per process: the most recently used (device, stream) pairs
_workspaces = collections.OrderedDict()


# Host-side cost of a call (tools/profile_host.py): torch.cuda.device(...) as a context manager and
# torch.cuda.current_stream(...).cuda_stream cost 6 of the 33 us an eager forward + backward pair took to issue (33.1 -> 27.0 us per step) --
# both resolve the device index through several Python layers.  The guard below is a no-op when the tensors' device is
# already current (the usual case), and the stream handle comes from the C binding where this torch has it.
_get_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def _stream_handle(dev):
    """The current HIP stream of `dev` as an integer handle (what the C ABI takes)."""
    if _get_raw_stream is not None and dev.index is not None:
        return _get_raw_stream(dev.index)
    return torch.cuda.current_stream(dev).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on_device(dev):
    """`with _on_device(dev):` -- torch.cuda.device(dev) unless `dev` is the current device already."""
    if dev.index is not None and torch.cuda.current_device() == dev.index:
        return _NO_GUARD
    return torch.cuda.device(dev)


def _workspace(device, nbytes):
    """Grow-only scratch per (device, stream) -- the analogue of the reference's grow-only GL buffers
    (csrc/rasterise_egl.cpp:325-333), but owned by the caller's allocator, not by the library -- for the calls that keep
    no state (inference, `_op_visibility`, the stateless backward).  At most _WORKSPACE_SLOTS of them are kept (least
    recently used first out): a long job that creates streams as it goes does not accumulate one buffer per stream
    handle it ever saw.  An evicted buffer goes back to torch's caching allocator, which keeps it alive until the work
    already queued on its stream has run (it was allocated on that stream)."""
    key = (device.index, _stream_handle(device))
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    _workspaces.move_to_end(key)
    while len(_workspaces) > _WORKSPACE_SLOTS:
        _workspaces.popitem(last=False)
    return ws


def _as_tensor(x, dtype, like=None):
    """tf.convert_to_tensor(x, dtype=...) of dirt/rasterise_ops.py:44-47,68-71."""
    if isinstance(x, torch.Tensor):
        return x if x.dtype == dtype else x.to(dtype)
    device = like.device if isinstance(like, torch.Tensor) else None
    return torch.as_tensor(x, dtype=dtype, device=device)


def _first_tensor(*xs):
    for x in xs:
        if isinstance(x, torch.Tensor) and x.is_cuda:
            return x
    for x in xs:
        if isinstance(x, torch.Tensor):
            return x
    return None


def _check_forward_shapes(background, vertices, vertex_colors, faces, height, width, channels):
    # OP_REQUIRES conditions of csrc/rasterise_egl.cpp:301-316, same messages
    if not (background.dim() == 4 and background.shape[1] == height and background.shape[2] == width
            and background.shape[3] == channels):
        raise ValueError('Rasterise expects background_tensor to be 4D, and bgcolor.shape == [None, height, width, channels]')
    if not (vertices.dim() == 3 and vertices.shape[2] == 4):
        raise ValueError('Rasterise expects vertices to be 3D, and vertices.shape[2] == 4')
    if not (vertex_colors.dim() == 3 and vertex_colors.shape[1] == vertices.shape[1] and vertex_colors.shape[2] == channels):
        raise ValueError('Rasterise expects vertex_colors to be 3D, and vertex_colors.shape == [None, vertices.shape[1], channels]')
    if not ((faces.dim() == 3 or faces.dim() == 2) and faces.shape[-1] == 3):  # 2D: one topology shared by the batch
        raise ValueError('Rasterise expects faces to be 3D, and faces.shape[2] == 3')
    batch_size = vertices.shape[0]
    if not (background.shape[0] == batch_size and vertex_colors.shape[0] == batch_size
            and (faces.dim() == 2 or faces.shape[0] == batch_size)):
        raise ValueError('Rasterise expects all arguments to have same leading (batch) dimension')


def _check_backward_shapes(vertices, faces, pixels, grad_pixels):
    # OP_REQUIRES conditions of csrc/rasterise_grad_egl.cpp:349-377
    if not (vertices.dim() == 3 and vertices.shape[2] == 4):
        raise ValueError('RasteriseGrad expects vertices to be 3D, and vertices.shape[2] == 4')
    if not ((faces.dim() == 3 or faces.dim() == 2) and faces.shape[-1] == 3):  # 2D: one topology shared by the batch
        raise ValueError('RasteriseGrad expects faces to be 3D, and faces.shape[2] == 3')
    if pixels.dim() != 4:
        raise ValueError('RasteriseGrad expects pixels to be 4D, and pixels.shape == [None, height, width, channels]')
    if grad_pixels.dim() != 4 or grad_pixels.shape[1:] != pixels.shape[1:]:
        raise ValueError('RasteriseGrad expects grad_pixels to be 4D, and grad_pixels.shape == [None, height, width, channels]')
    batch_size = vertices.shape[0]
    if not ((faces.dim() == 2 or faces.shape[0] == batch_size) and pixels.shape[0] == batch_size
            and grad_pixels.shape[0] == batch_size):
        raise ValueError('RasteriseGrad expects all arguments to have same leading (batch) dimension')


def _require_gpu(*tensors):
    dev = None
    for t in tensors:
        if not t.is_cuda:
            raise RuntimeError(
                'dirt_amd ops run on an MI355X only (the reference registers its kernels for DEVICE_GPU only, '
                'csrc/rasterise_egl.cpp:410); got a %s tensor. There is no CPU fallback.' % t.device)
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise ValueError('all tensors must be on the same device (%s vs %s)' % (dev, t.device))
    return dev


_SIZES = {}   # (B, V, F, H, W, C) -> workspace bytes: a pure function of the sizes, asked once (host-side cost per call)
_LAYOUTS = {}  # ... -> (offset of grad_vertices, offset of grad_vertex_colors, row strides) of the state's accumulators


def _workspace_bytes(lib, B, V, F, H, W, C):
    key = (B, V, F, H, W, C)
    n = _SIZES.get(key)
    if n is None:
        n = lib.dirt_workspace_bytes(B, V, F, H, W, C)
        if n == 0:
            raise ValueError(_lib.last_error())
        _SIZES[key] = n
    return n


def _dense16(t):
    """A dense tensor whose storage the kernels may access with 16-byte loads: `.contiguous()` returns contiguous
    views unchanged, and a view such as `x[1:]` of a 5x5x3 frame starts at an address that is not a multiple of 16
    (the C ABI rejects those); the reference accepts any tensor, so such views are copied."""
    t = t.contiguous()
    if t.data_ptr() % 16 != 0:
        t = t.clone(memory_format=torch.contiguous_format)
    return t


def _op_rasterise(background, vertices, vertex_colors, faces, height, width, channels, flags=0, keep_state=False,
                  state_channels=0, dense_grads=False):
    """`_rasterise_module.rasterise` (dirt/rasterise_ops.py:81-85): the raw forward op, no autograd.

    With keep_state=True returns (pixels, state): `state` is a private workspace holding the set-up
    records and the visibility buffer, to be handed to `_op_rasterise_grad(..., state=state)` so the
    backward pass does not render again (DIRT_FLAG_KEEP_STATE / DIRT_FLAG_REUSE_STATE).  `state_channels`
    sizes the state for later backward calls with up to that many channels (deferred shading: the shaded
    image need not have the G-buffer's channel count).  `dense_grads` (with keep_state): the forward launch also
    clears the DENSE grad_vertices / grad_vertex_colors tensors of the coming backward call
    (dirt_rasterise_forward_train: the RasteriseGrad op's outputs, csrc/rasterise_grad_egl.cpp:381-391); they ride on
    the state until `_op_rasterise_grad(..., state_outputs='dense')` takes them."""
    lib = _lib.load()
    _check_forward_shapes(background, vertices, vertex_colors, faces, height, width, channels)
    dev = _require_gpu(background, vertices, vertex_colors, faces)
    background, vertices, vertex_colors, faces = (_dense16(t) for t in (background, vertices, vertex_colors, faces))
    B, V, F = vertices.shape[0], vertices.shape[1], faces.shape[-2]
    if faces.dim() == 2:
        flags |= _lib.FLAG_SHARED_FACES
    pixels = torch.empty_like(background)
    grads = None
    with _on_device(dev):
        nbytes = _workspace_bytes(lib, B, V, F, height, width, channels)
        if keep_state:
            if state_channels > channels:
                nbytes = _workspace_bytes(lib, B, V, F, height, width, state_channels)
            # one state per autograd forward (it must outlive the call, until its backward).  torch's caching allocator is
            # the free list: the block of a state whose graph has died is handed out again for the next forward of the
            # same size, so a training loop cycles through one or two blocks and reserved memory stays flat
            # (tools/soak.py asserts that over 10 000 steps).
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            flags |= _lib.FLAG_KEEP_STATE
        else:
            ws = _workspace(dev, nbytes)
        if keep_state and dense_grads and B * V > 0:
            grads = (torch.empty_like(vertices), torch.empty((B, V, channels), dtype=torch.float32, device=dev))
            _lib.check(lib.dirt_rasterise_forward_train(
                background.data_ptr(), vertices.data_ptr(), vertex_colors.data_ptr(), faces.data_ptr(), pixels.data_ptr(),
                grads[0].data_ptr(), grads[1].data_ptr(),
                B, V, F, height, width, channels, ws.data_ptr(), ws.numel(), flags, _stream_handle(dev)))
        else:
            _lib.check(lib.dirt_rasterise_forward(
                background.data_ptr(), vertices.data_ptr(), vertex_colors.data_ptr(), faces.data_ptr(), pixels.data_ptr(),
                B, V, F, height, width, channels, ws.data_ptr(), ws.numel(), flags, _stream_handle(dev)))
    if keep_state:
        ws._dirt_channels = channels   # the layout of the state's gradient accumulators depends on the channel count
        ws._dirt_grads = grads         # dense gradient outputs this forward cleared (one backward call may take them)
    return (pixels, ws) if keep_state else pixels


def _op_rasterise_grad(vertices, faces, pixels, grad_pixels, height, width, channels, flags=0, want_debug=False,
                       state=None, state_outputs=True):
    """`_rasterise_module.rasterise_grad` (dirt/rasterise_ops.py:113-118): returns
    (grad_background, grad_vertices, grad_vertex_colors, debug_thingy or None).

    `state`: the workspace a keep_state forward of the same vertices / faces / frame left behind; the call then
    neither sets up nor renders again.  With state_outputs=True (one backward per forward: autograd) the vertex
    gradients accumulate in the buffers that forward pre-cleared inside the state and the returned tensors are
    views of it; with state_outputs='dense' the call returns DENSE tensors, the op's contract (csrc/rasterise_grad_egl.cpp:381-391):
    the ones a `dense_grads` forward cleared in its own launch, added into directly (DIRT_FLAG_OUTPUTS_CLEARED: what the autograd
    path and bench.py use), or else fresh ones that receive a copy of the state's accumulators (DIRT_FLAG_DENSE_FROM_STATE); with
    state_outputs=False they are fresh tensors, cleared and added into directly, so a state can serve any number of
    backward calls (deferred shading: one for the shaded image, one for the G-buffer)."""
    lib = _lib.load()
    _check_backward_shapes(vertices, faces, pixels, grad_pixels)
    if tuple(pixels.shape[1:]) != (height, width, channels):
        raise ValueError('RasteriseGrad expects pixels to be 4D, and pixels.shape == [None, height, width, channels]')
    dev = _require_gpu(vertices, faces, pixels, grad_pixels)
    vertices, faces, pixels, grad_pixels = (_dense16(t) for t in (vertices, faces, pixels, grad_pixels))
    B, V, F = vertices.shape[0], vertices.shape[1], faces.shape[-2]
    if faces.dim() == 2:
        flags |= _lib.FLAG_SHARED_FACES
    grad_background = torch.empty_like(pixels)
    debug = torch.empty((B, height, width, 3), dtype=torch.float32, device=dev) if want_debug else None
    with _on_device(dev):
        nbytes = _workspace_bytes(lib, B, V, F, height, width, channels)
        if state is not None and state.numel() < nbytes:
            state = None  # sized for fewer channels than this call has: render again
        if state is not None and getattr(state, '_dirt_channels', channels) != channels:
            state_outputs = False  # the accumulators inside the state were laid out (and cleared) for another channel count
        taken = None
        if state is not None and state_outputs == 'dense' and getattr(state, '_dirt_grads', None) is not None:
            taken, state._dirt_grads = state._dirt_grads, None   # the tensors that forward cleared: taken once
            if tuple(taken[1].shape) != (B, V, channels):
                taken = None
        if taken is not None:
            ws = state
            flags |= _lib.FLAG_REUSE_STATE | _lib.FLAG_OUTPUTS_CLEARED
            grad_vertices, grad_vertex_colors = taken
        elif state is not None and (not state_outputs or state_outputs == 'dense'):
            ws = state
            flags |= _lib.FLAG_REUSE_STATE | (_lib.FLAG_DENSE_FROM_STATE if state_outputs == 'dense' else 0)
            grad_vertices = torch.empty_like(vertices)
            grad_vertex_colors = torch.empty((B, V, channels), dtype=torch.float32, device=dev)
        elif state is not None:
            # the gradients accumulate in the buffers the forward pass already cleared inside `state`;
            # the returned tensors are views of it
            ws = state
            flags |= _lib.FLAG_REUSE_STATE
            key = (B, V, F, height, width, channels, ws.data_ptr() % 256)
            layout = _LAYOUTS.get(key)
            if layout is None:
                gv_p, gvc_p, gv_s, gvc_s = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
                _lib.check(lib.dirt_state_grad_buffers(ws.data_ptr(), ws.numel(), B, V, F, height, width, channels,
                                                       ctypes.byref(gv_p), ctypes.byref(gvc_p), ctypes.byref(gv_s), ctypes.byref(gvc_s)))
                # offsets (in floats) inside the workspace and row strides: they depend on the sizes and on the workspace's
                # alignment only (the library carves from the next 256-byte boundary)
                layout = _LAYOUTS[key] = ((gv_p.value - ws.data_ptr()) // 4, (gvc_p.value - ws.data_ptr()) // 4, gv_s.value, gvc_s.value)
            o1, o2, s1, s2 = layout
            # rows of s1 / s2 floats (the two accumulators interleaved: one row per vertex): strided views
            if ws.data_ptr() % 4 != 0 or ws.numel() % 4 != 0:
                raise RuntimeError('state workspace is not float-aligned')
            wsf = ws.view(torch.float32)
            grad_vertices = torch.as_strided(wsf, (B, V, 4), (V * s1, s1, 1), o1)
            grad_vertex_colors = torch.as_strided(wsf, (B, V, channels), (V * s2, s2, 1), o2)
        else:
            ws = _workspace(dev, nbytes)
            grad_vertices = torch.empty_like(vertices)
            grad_vertex_colors = torch.empty((B, V, channels), dtype=torch.float32, device=dev)
        _lib.check(lib.dirt_rasterise_backward(
            vertices.data_ptr(), faces.data_ptr(), pixels.data_ptr(), grad_pixels.data_ptr(),
            grad_background.data_ptr(), grad_vertices.data_ptr(), grad_vertex_colors.data_ptr(),
            debug.data_ptr() if want_debug else None,
            B, V, F, height, width, channels, ws.data_ptr(), ws.numel(), flags,
            _stream_handle(dev)))
    return grad_background, grad_vertices, grad_vertex_colors, debug


def _op_visibility(vertices, faces, height, width):
    """Front-most face per pixel, [B,H,W] int32 (-1 = background)."""
    lib = _lib.load()
    dev = _require_gpu(vertices, faces)
    vertices, faces = _dense16(vertices), _dense16(faces)
    B, V, F = vertices.shape[0], vertices.shape[1], faces.shape[-2]
    flags = _lib.FLAG_SHARED_FACES if faces.dim() == 2 else 0
    face_id = torch.empty((B, height, width), dtype=torch.int32, device=dev)
    with _on_device(dev):
        nbytes = lib.dirt_workspace_bytes(B, V, F, height, width, 1)
        ws = _workspace(dev, nbytes)
        _lib.check(lib.dirt_rasterise_visibility(
            vertices.data_ptr(), faces.data_ptr(), face_id.data_ptr(), B, V, F, height, width,
            ws.data_ptr(), ws.numel(), flags, _stream_handle(dev)))
    return face_id


class _Rasterise(torch.autograd.Function):
    """The `Rasterise` op with its registered gradient (dirt/rasterise_ops.py:111-129)."""

    @staticmethod
    def forward(ctx, background, vertices, vertex_colors, faces, height, width, channels):
        needs_grad = any(ctx.needs_input_grad[:3])
        if needs_grad:
            pixels, state = _op_rasterise(background, vertices, vertex_colors, faces, height, width, channels, keep_state=True, dense_grads=True)
        else:
            pixels, state = _op_rasterise(background, vertices, vertex_colors, faces, height, width, channels), None
        ctx.save_for_backward(vertices, faces, pixels)  # op.inputs[1], op.inputs[3], op.outputs[0]
        ctx.state = state  # set-up records + visibility of this very call (never shared with other calls)
        ctx.hwc = (height, width, channels)
        return pixels

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_pixels):
        vertices, faces, pixels = ctx.saved_tensors
        height, width, channels = ctx.hwc
        # The first backward accumulates into the buffers the forward pass pre-cleared inside the state (no clearing
        # launch) and returns views of them.  A later backward over the same forward (retain_graph=True, per-loss
        # autograd.grad, Jacobian loops) must neither add onto those nor alias them: it gets fresh, cleared outputs --
        # the reference's grad op is pure (csrc/rasterise_grad_egl.cu:244-250 clears its outputs on every call).
        first = not getattr(ctx, 'state_outputs_used', False)
        ctx.state_outputs_used = True
        # autograd gets DENSE tensors, as the reference's op returns (csrc/rasterise_grad_egl.cpp:381-391): the ones the forward
        # launch cleared (`dense_grads`), which the gradient kernel adds into directly -- a backward call is ONE launch.
        # (Round 4 summed in the state's interleaved accumulators and copied them out with a second launch.)
        if grad_pixels.dtype != torch.float32:
            grad_pixels = grad_pixels.to(torch.float32)
        grad_background, grad_vertices, grad_vertex_colors, _ = _op_rasterise_grad(
            vertices, faces, pixels, grad_pixels, height, width, channels, state=ctx.state,
            state_outputs='dense' if first else False)
        return grad_background, grad_vertices, grad_vertex_colors, None, None, None, None  # None wrt faces


def rasterise(background, vertices, vertex_colors, faces, height=None, width=None, channels=None, name=None):
    """Rasterises the given `vertices` and `faces` over `background` (dirt/rasterise_ops.py:13-48).

    Args:
        background: float32 tensor [height, width, channels], the image to render over
        vertices: float32 tensor [vertex count, 4], vertex locations in OpenGL clip space
        vertex_colors: float32 tensor [vertex count, channels]; interpolated perspective-correctly
        faces: int32 tensor [face count, 3] of indices into `vertices`
        height, width, channels: python ints; inferred from `background` when None
        name: ignored (kept for signature compatibility)

    Returns: float32 tensor [height, width, channels], top row first.
    """
    like = _first_tensor(background, vertices, vertex_colors, faces)
    background = _as_tensor(background, torch.float32, like)
    vertices = _as_tensor(vertices, torch.float32, like)
    vertex_colors = _as_tensor(vertex_colors, torch.float32, like)
    faces = _as_tensor(faces, torch.int32, like)
    return rasterise_batch(background[None], vertices[None], vertex_colors[None], faces[None], height, width, channels, name)[0]


def rasterise_batch(background, vertices, vertex_colors, faces, height=None, width=None, channels=None, name=None):
    """Rasterises a batch of meshes with the same numbers of vertices and faces
    (dirt/rasterise_ops.py:51-108); every argument of `rasterise` gains a leading batch dimension.

    Extension: `faces` may also be [face count, 3] -- one topology shared by the whole batch, which the reference
    can only express by tiling it (the TODO at csrc/rasterise_egl.cpp:314, tests/rasterise_tests.py:89)."""
    like = _first_tensor(background, vertices, vertex_colors, faces)
    background = _as_tensor(background, torch.float32, like)
    vertices = _as_tensor(vertices, torch.float32, like)
    vertex_colors = _as_tensor(vertex_colors, torch.float32, like)
    faces = _as_tensor(faces, torch.int32, like)
    if background.dim() != 4:
        raise ValueError('Rasterise expects background_tensor to be 4D, and bgcolor.shape == [None, height, width, channels]')
    if height is None:
        height = int(background.shape[1])
    if width is None:
        width = int(background.shape[2])
    if channels is None:
        channels = int(background.shape[3])
    assert channels > 0  # dirt/rasterise_ops.py:87
    return _Rasterise.apply(background, vertices, vertex_colors, faces, int(height), int(width), int(channels))


def _rasterise_grad_multichannel(vertices, faces, pixels, d_loss_by_pixels, single_or_batch, state=None):
    """dirt/rasterise_ops.py:132-177; one native call evaluates every channel group.  `state`: see
    `_op_rasterise_grad` (used with state_outputs=False, so it may be shared between calls)."""
    assert single_or_batch in ['single', 'batch']
    if single_or_batch == 'single':
        vertices, faces, pixels, d_loss_by_pixels = vertices[None], faces[None], pixels[None], d_loss_by_pixels[None]
    assert pixels.dim() == 4
    height, width, channels = (int(s) for s in pixels.shape[1:])
    gb, gv, gvc, _ = _op_rasterise_grad(vertices, faces, pixels, d_loss_by_pixels, height, width, channels,
                                        state=state, state_outputs=False)
    if single_or_batch == 'single':
        return {'grad_vertices': gv[0], 'grad_vertex_colors': gvc[0], 'grad_background': gb[0]}
    return {'grad_vertices': gv, 'grad_vertex_colors': gvc, 'grad_background': gb}


_checked_shaders = collections.OrderedDict()   # code object (or id of a callable without one) -> closures of it whose graph was walked
_CHECKED_SHADERS_MAX = 256
_WALKS_PER_CODE = 4   # distinct closures of one code object that get the graph walk; a lambda re-created every step over
                      # fresh tensors (the common training-loop pattern) then stops paying for it after four steps


def _closure_identity(shader_fn):
    """(code object, objects the function closes over / is bound to): what identifies a shader for the one-time graph walk.
    Two closures of one factory share a code object but not their parameters."""
    fn = shader_fn if hasattr(shader_fn, '__code__') else getattr(shader_fn, '__call__', shader_fn)
    code = getattr(fn, '__code__', None)
    held = []
    for c in (getattr(fn, '__closure__', None) or ()):
        try:
            held.append(c.cell_contents)
        except ValueError:   # an empty cell
            held.append(None)
    held.append(getattr(fn, '__self__', None))
    return (code if code is not None else id(shader_fn)), held


class _Id:
    """Identity of an object that cannot be weakly referenced (ints, strings, tuples, lists, dicts): its id and type -- never
    the object itself (a closure over a list of tensors must not pin them) and never its value (`==` on user objects can
    raise or synchronise the device: a list of tensors compares element-wise and then asks for a truth value)."""
    __slots__ = ('ident', 'kind', 'parts')

    def __init__(self, obj, depth):
        self.ident, self.kind = id(obj), type(obj)
        # one level into the common containers, so that a list rebuilt every step over the SAME tensors still matches
        # and one over fresh tensors does not (an id alone can be recycled once the container is freed)
        self.parts = None
        if depth > 0 and isinstance(obj, (list, tuple)) and len(obj) <= 64:
            self.parts = [_ref(o, depth - 1) for o in obj]
        elif depth > 0 and isinstance(obj, dict) and len(obj) <= 64:
            self.parts = [_ref(o, depth - 1) for o in obj.values()]


def _ref(obj, depth=1):
    """A weak reference where the object allows one (tensors, modules, functions); else an `_Id`."""
    import weakref
    try:
        return weakref.ref(obj)
    except TypeError:
        return _Id(obj, depth)


def _same_one(r, o, depth=1):
    import weakref
    if isinstance(r, weakref.ref):
        return r() is not None and r() is o     # a dead referent never matches: ids of freed objects get reused, references do not
    if r.kind is not type(o):
        return False
    if r.parts is not None:                     # a container: same length, same elements (by identity)
        items = list(o.values()) if isinstance(o, dict) else (list(o) if isinstance(o, (list, tuple)) else None)
        return items is not None and len(items) == len(r.parts) and all(_same_one(p, q, depth - 1) for p, q in zip(r.parts, items))
    return r.ident == id(o)


def _same(entry, held):
    """Strictly by identity: no `==` on anything a shader closes over."""
    return len(entry) == len(held) and all(_same_one(r, o) for r, o in zip(entry, held))


_walk_limit_warned = set()   # code objects whose graph walk has been switched off (warned once each)


def _shader_needs_walk(shader_fn):
    """True the first time this (code, closure) is seen -- and only for the first _WALKS_PER_CODE closures of a code object:
    beyond that (a lambda re-created every step over fresh tensors) the check is switched off for that code object, with
    one warning saying so."""
    code, held = _closure_identity(shader_fn)
    entries = _checked_shaders.get(code)
    if entries is None:
        entries = _checked_shaders[code] = []
    _checked_shaders.move_to_end(code)
    while len(_checked_shaders) > _CHECKED_SHADERS_MAX:   # bounded: a job that builds shaders as it goes does not grow it
        _checked_shaders.popitem(last=False)
    if any(_same(e, held) for e in entries):
        return False
    if len(entries) >= _WALKS_PER_CODE:
        if code not in _walk_limit_warned:
            if len(_walk_limit_warned) < _CHECKED_SHADERS_MAX:
                _walk_limit_warned.add(code)
            import warnings
            warnings.warn('rasterise_deferred: %d different closures of one shader function have been checked for tensors that '
                          'require grad but are not listed in shader_additional_inputs / shader_parameters; further closures of '
                          'it are not checked (the check walks the autograd graph on the host).' % _WALKS_PER_CODE, stacklevel=4)
        return False
    entries.append([_ref(o) for o in held])
    return True


def _unlisted_leaves(output, listed, limit=2000):
    """Leaf tensors that `output` depends on (requires_grad) and that are not in `listed`: parameters a shader closes over
    without naming them in `shader_parameters`.  TensorFlow's custom_gradient hands those to the gradient function as
    `variables` (dirt/rasterise_ops.py:239-246); a torch.autograd.Function cannot, so they would silently get no gradient."""
    if not isinstance(output, torch.Tensor) or output.grad_fn is None:
        return []
    known = {id(t) for t in listed if isinstance(t, torch.Tensor)}
    found, seen, stack = [], set(), [output.grad_fn]   # (the nodes themselves are kept: ids of dead wrappers get reused)
    while stack and len(seen) < limit:
        fn = stack.pop()
        if fn is None or fn in seen:
            continue
        seen.add(fn)
        var = getattr(fn, 'variable', None)   # AccumulateGrad: a leaf
        if var is not None and id(var) not in known and all(var is not f for f in found):
            found.append(var)
        stack.extend(nf for nf, _ in fn.next_functions)
    return found


class _RasteriseDeferred(torch.autograd.Function):
    """`_rasterise_deferred_internal._impl` (dirt/rasterise_ops.py:189-248) as a custom autograd node."""

    @staticmethod
    def forward(ctx, shader_fn, single_or_batch, n_extra, vertices, faces, attributes, background, *rest):
        shader_additional_inputs = rest[:n_extra]
        shader_params = rest[n_extra:]  # parameters closed over by shader_fn (TF's `variables`)
        # ONE visibility pass serves the forward and both gradient calls of the backward (the reference draws the
        # scene once per channel group in each of the three: dirt/rasterise_ops.py:86-108,145-165)
        batched = [t if single_or_batch == 'batch' else t[None] for t in (background, vertices, attributes, faces)]
        height, width, channels = (int(n) for n in batched[0].shape[1:])
        needs_grad = any(ctx.needs_input_grad[3:7])
        if needs_grad:
            gbuffer, state = _op_rasterise(*batched, height, width, channels, keep_state=True, state_channels=4)
        else:
            gbuffer, state = _op_rasterise(*batched, height, width, channels), None
        if single_or_batch == 'single':
            gbuffer = gbuffer[0]
        ctx.state = state
        with torch.enable_grad():
            gbuffer_in = gbuffer.detach().requires_grad_(True)
            extra_in = [t.detach().requires_grad_(t.is_floating_point()) if isinstance(t, torch.Tensor) else t
                        for t in shader_additional_inputs]
            pixels = shader_fn(gbuffer_in, *extra_in)
        # (the graph walk is a debugging aid with a host-side cost: done on the first call of each shader function only)
        stray = _unlisted_leaves(pixels, [gbuffer_in] + list(extra_in) + list(shader_params)) if _shader_needs_walk(shader_fn) else []
        if stray:
            import warnings
            warnings.warn('rasterise_deferred: shader_fn uses %d tensor(s) that require grad but are neither '
                          'shader_additional_inputs nor shader_parameters (shapes %s); they will receive no gradient. '
                          'Pass them as shader_parameters=[...] (the counterpart of the `variables` TensorFlow hands to '
                          'a custom_gradient).' % (len(stray), [tuple(t.shape) for t in stray[:4]]), stacklevel=3)
        ctx.single_or_batch = single_or_batch
        ctx.n_extra = n_extra
        ctx.n_params = len(shader_params)
        ctx.graph = (gbuffer_in, extra_in, pixels, shader_params)
        ctx.save_for_backward(vertices, faces)
        return pixels.detach()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, d_loss_by_pixels):
        vertices, faces = ctx.saved_tensors
        gbuffer_in, extra_in, pixels, shader_params = ctx.graph
        sob = ctx.single_or_batch
        # vertex gradients from filtering the SHADED image (dirt/rasterise_ops.py:204-210)
        d_loss_by_vertices = _rasterise_grad_multichannel(
            vertices, faces, pixels.detach(), d_loss_by_pixels.contiguous(), sob, ctx.state)['grad_vertices']
        # backprop through shader_fn to the G-buffer (dirt/rasterise_ops.py:212-229)
        diff_extra = [t for t in extra_in if isinstance(t, torch.Tensor) and t.requires_grad]
        diff_params = [t for t in shader_params if isinstance(t, torch.Tensor) and t.requires_grad]
        if pixels.requires_grad:
            grads = torch.autograd.grad(pixels, [gbuffer_in] + diff_extra + diff_params, d_loss_by_pixels, allow_unused=True,
                                        retain_graph=True)  # the node may be differentiated again (retain_graph=True upstream)
        else:  # a shader that does not depend on any differentiable input
            grads = [None] * (1 + len(diff_extra) + len(diff_params))
        d_loss_by_gbuffer = grads[0]
        if d_loss_by_gbuffer is None:
            d_loss_by_gbuffer = torch.zeros_like(gbuffer_in)
        extra_grads = iter(grads[1:1 + len(diff_extra)])
        param_grads = iter(grads[1 + len(diff_extra):])
        # attribute / background gradients from the G-buffer (dirt/rasterise_ops.py:231-237)
        d_attr = _rasterise_grad_multichannel(vertices, faces, gbuffer_in.detach(), d_loss_by_gbuffer.contiguous(), sob,
                                              ctx.state)
        out = [None, None, None, d_loss_by_vertices, None, d_attr['grad_vertex_colors'], d_attr['grad_background']]
        for t in extra_in:
            out.append(next(extra_grads) if isinstance(t, torch.Tensor) and t.requires_grad else None)
        for t in shader_params:
            out.append(next(param_grads) if isinstance(t, torch.Tensor) and t.requires_grad else None)
        return tuple(out)


def _rasterise_deferred_internal(background, vertices, attributes, faces, shader_fn, shader_additional_inputs,
                                 single_or_batch, name, shader_parameters=()):
    like = _first_tensor(background, vertices, attributes, faces)
    background = _as_tensor(background, torch.float32, like)
    vertices = _as_tensor(vertices, torch.float32, like)
    attributes = _as_tensor(attributes, torch.float32, like)
    faces = _as_tensor(faces, torch.int32, like)
    extra = [t if isinstance(t, torch.Tensor) else torch.as_tensor(t, device=background.device)
             for t in shader_additional_inputs]
    params = list(shader_parameters)
    return _RasteriseDeferred.apply(shader_fn, single_or_batch, len(extra), vertices, faces, attributes, background,
                                    *extra, *params)


def rasterise_deferred(background_attributes, vertices, vertex_attributes, faces, shader_fn,
                       shader_additional_inputs=[], name=None, shader_parameters=()):
    """Rasterises a G-buffer of `vertex_attributes` and shades it with `shader_fn`
    (dirt/rasterise_ops.py:260-310).  Equivalent to
    `shader_fn(rasterise(background_attributes, vertices, vertex_attributes, faces), *shader_additional_inputs)`
    but the vertex gradient is obtained by filtering the shaded image.  Tensors that `shader_fn`
    depends on must be passed through `shader_additional_inputs`; `torch.nn.Parameter`s it closes over
    may be listed in `shader_parameters` (the counterpart of TF's `variables`)."""
    return _rasterise_deferred_internal(background_attributes, vertices, vertex_attributes, faces, shader_fn,
                                        shader_additional_inputs, 'single', name, shader_parameters)


def rasterise_batch_deferred(background_attributes, vertices, vertex_attributes, faces, shader_fn,
                             shader_additional_inputs=[], name=None, shader_parameters=()):
    """Batched `rasterise_deferred` (dirt/rasterise_ops.py:313-332)."""
    return _rasterise_deferred_internal(background_attributes, vertices, vertex_attributes, faces, shader_fn,
                                        shader_additional_inputs, 'batch', name, shader_parameters)
