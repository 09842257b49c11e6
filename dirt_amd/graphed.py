"""A training step of `rasterise_batch` captured ONCE as a HIP graph and replayed per iteration.

Why: `dirt.rasterise_batch(...).backward()` through torch's eager autograd costs 130-200 us of host time per step at
1024 x 1024 (the backward engine hands every GPU node to a per-device worker thread and waits for it; the Python wrappers;
the allocator), three to four times the 48 us the kernels take -- the GPU idles two thirds of such a step (bench.py:
`ms_per_step_autograd`).  The reference has the same shape of problem and the same remedy: a TensorFlow graph is built once
and `session.run` replays it (dirt/rasterise_ops.py:111-129 registers the gradient into that graph).  Here the graph is a
HIP graph: forward (set-up + raster kernels), the caller's loss and the registered gradient (one gradient kernel) are
recorded once on fixed buffers; a step is then one `hipGraphLaunch`.

    step = dirt_amd.GraphedStep(background, vertices, vertex_colors, faces, loss_fn=lambda px: ((px - target) ** 2).mean())
    for it in range(n):
        loss, (g_background, g_vertices, g_vertex_colors) = step()     # one graph launch
        with torch.no_grad():
            vertices -= lr * g_vertices                                 # in place: the graph reads these very tensors

The tensors given to the constructor ARE the graph's inputs (the CUDA-graphs idiom: update them in place, or `copy_` new
data into them, between steps); the returned tensors are the graph's static outputs, overwritten by the next step.
Without `loss_fn` the step is the vector-Jacobian product with a `grad_pixels` tensor (also bound in place): exactly
`rasterise_batch(...).backward(grad_pixels)`.
"""
import torch

from . import rasterise_ops as ops

__all__ = ['GraphedStep', 'backward']


def _cell_has(cell):
    try:
        cell.cell_contents
        return True
    except ValueError:   # an empty cell
        return False


class GraphedStep:
    """rasterise_batch -> [loss_fn] -> gradients, captured as one HIP graph for the shapes of the given tensors.

    Args:
        background [B,H,W,C], vertices [B,V,4], vertex_colors [B,V,C]: float32, contiguous, on the GPU -- bound IN PLACE;
        faces [B,F,3] or [F,3] int32 (bound in place; not differentiated, as in the reference: dirt/rasterise_ops.py:129);
        loss_fn: pixels [B,H,W,C] -> scalar tensor, traced into the graph with its backward (it must be capturable:
            no host synchronisation, fixed shapes); or None;
        grad_pixels [B,H,W,C]: with loss_fn=None, the gradient w.r.t. the pixels (bound in place);
        warmup: eager iterations on a side stream before capture (allocator and library warm-up).
    Attributes after construction: `pixels`, `loss` (or None), `grads` = (grad_background, grad_vertices,
    grad_vertex_colors): the graph's static outputs.
    """

    def __init__(self, background, vertices, vertex_colors, faces, loss_fn=None, grad_pixels=None, warmup=3):
        for t in (background, vertices, vertex_colors):
            if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
                raise ValueError('GraphedStep binds float32, contiguous GPU tensors in place')
        if not (isinstance(faces, torch.Tensor) and faces.is_cuda and faces.dtype == torch.int32 and faces.is_contiguous()):
            raise ValueError('GraphedStep: faces must be an int32, contiguous GPU tensor')
        if background.dim() != 4:
            raise ValueError('Rasterise expects background_tensor to be 4D, and bgcolor.shape == [None, height, width, channels]')
        if (loss_fn is None) == (grad_pixels is None):
            raise ValueError('GraphedStep needs exactly one of loss_fn and grad_pixels')
        if grad_pixels is not None and not (isinstance(grad_pixels, torch.Tensor) and grad_pixels.is_cuda and grad_pixels.dtype == torch.float32
                                            and grad_pixels.is_contiguous() and grad_pixels.shape == background.shape):
            raise ValueError('GraphedStep: grad_pixels must be a float32, contiguous GPU tensor of the image\'s shape')
        bound = [t for t in (background, vertices, vertex_colors, faces, grad_pixels) if t is not None]
        if any(t.device != background.device for t in bound):
            raise ValueError('GraphedStep: all bound tensors must be on one device (%s)' % ', '.join(str(t.device) for t in bound))
        # the graph reads and writes these very buffers: two of them sharing storage (grad_pixels aliasing the background, say)
        # would make a replay read what the same replay overwrites
        spans = [(t.untyped_storage().data_ptr() + t.storage_offset() * t.element_size(), t.numel() * t.element_size()) for t in bound]
        for i in range(len(spans)):
            for j in range(i + 1, len(spans)):
                (a0, an), (b0, bn) = spans[i], spans[j]
                if an and bn and a0 < b0 + bn and b0 < a0 + an:
                    raise ValueError('GraphedStep: bound tensors must not share memory')
        # tensors loss_fn closes over (a target image, say) are baked into the graph BY ADDRESS: they are kept alive here, and
        # must be updated in place like the bound inputs
        self._loss_closure = [c.cell_contents for c in (getattr(loss_fn, '__closure__', None) or ()) if _cell_has(c)]
        self.background, self.vertices, self.vertex_colors, self.faces = background, vertices, vertex_colors, faces
        self.grad_pixels, self.loss_fn = grad_pixels, loss_fn
        self._hwc = tuple(int(n) for n in background.shape[1:])
        dev = background.device
        with torch.cuda.device(dev):
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(max(1, warmup)):
                    self._eager()
            torch.cuda.current_stream(dev).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.pixels, self.loss, self.grads = self._eager()

    def _eager(self):
        """One step through the ordinary autograd path (what the graph records)."""
        h, w, c = self._hwc
        leaves = [t.detach().requires_grad_(True) for t in (self.background, self.vertices, self.vertex_colors)]   # same storage
        pixels = ops.rasterise_batch(leaves[0], leaves[1], leaves[2], self.faces, h, w, c)
        if self.loss_fn is not None:
            loss = self.loss_fn(pixels)
            grads = torch.autograd.grad(loss, leaves)
            return pixels.detach(), loss.detach(), tuple(grads)
        grads = torch.autograd.grad(pixels, leaves, grad_outputs=self.grad_pixels)
        return pixels.detach(), None, tuple(grads)

    def __call__(self):
        """Replays the step on the current stream.  Returns (loss or pixels, (grad_background, grad_vertices,
        grad_vertex_colors)): static tensors, valid until the next call."""
        self.graph.replay()
        return (self.loss if self.loss is not None else self.pixels), self.grads

    replay = __call__


def backward(tensors, grad_tensors=None, **kwargs):
    """`torch.autograd.backward(tensors, grad_tensors, ...)` with torch's backward engine kept on the CALLING thread for the
    duration of the call (a scoped `torch.autograd.set_multithreading_enabled(False)`, restored on return -- no process-wide
    switch).  By default the engine hands every GPU node to a per-device worker thread and waits for it: two thread wake-ups
    per backward(), ~100 us of a 160 us eager step at 1024 x 1024 (bench.py: `ms_per_step_autograd` against
    `..._engine_on_calling_thread`).  For loops whose shapes change from step to step (where GraphedStep does not apply):

        loss = loss_fn(dirt_amd.rasterise_batch(background, vertices, vertex_colors, faces))
        dirt_amd.backward(loss)          # instead of loss.backward()
    """
    with torch.autograd.set_multithreading_enabled(False):
        torch.autograd.backward(tensors, grad_tensors, **kwargs)
