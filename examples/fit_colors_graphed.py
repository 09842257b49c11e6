#!/usr/bin/env python3
"""A static-shape optimisation loop on a captured HIP graph (dirt_amd.GraphedStep): vertex colours and a screen-space
offset of a random mesh are fitted to a target image by gradient descent.  One graph launch per iteration runs the forward
(set-up + raster kernels), the loss, and the registered gradient (one kernel); the update happens IN PLACE on the tensors the
graph was captured with.  The reference works the same way by construction: its ops and their registered gradient
(dirt/rasterise_ops.py:111-129) are nodes of a TensorFlow graph that is built once and evaluated in a session
(samples/simple.py:76-81).

    python examples/fit_colors_graphed.py
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import dirt_amd  # noqa: E402
from tests import scenes  # noqa: E402


def fit(device, height=256, width=256, faces=400, steps=60, lr_color=200.0, lr_shift=2.0e-3, verbose=True):
    """Returns the loss history.  The target is the same mesh with other colours, shifted by a few pixels."""
    s = scenes.rand_scene(faces, height, width, 3, seed=5, r_lo=0.05, r_hi=0.25)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    background, vertices, f = t(s['background'][None]) * 0.1, t(s['vertices'][None]), t(s['faces'][None])
    target_colors = t(s['vertex_colors'][None])
    shift = torch.tensor([0.03, -0.02], device=device)
    target_vertices = vertices.clone()
    target_vertices[..., :2] += shift * target_vertices[..., 3:4]          # a translation in NDC is (x + s w, y + s w) in clip space
    with torch.no_grad():
        target = dirt_amd.rasterise_batch(background, target_vertices, target_colors, f)
    colors = torch.full_like(target_colors, 0.5)
    step = dirt_amd.GraphedStep(background, vertices, colors, f, loss_fn=lambda px: ((px - target) ** 2).mean())
    history = []
    warm = min(5, steps // 2)   # (torch loads the update's own kernels on first use: not timed)
    t0 = None
    for it in range(steps):
        if it == warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        loss, (_, g_vertices, g_colors) = step()                         # one hipGraphLaunch
        with torch.no_grad():                                             # in place: the graph reads these very tensors
            colors -= lr_color * g_colors
            colors.clamp_(0.0, 1.0)
            # the mesh is rigid: the gradient of a common NDC shift is the sum of the per-vertex clip-space gradients times w
            g_shift = (g_vertices[..., :2] * vertices[..., 3:4]).sum(dim=(0, 1))
            vertices[..., :2] -= lr_shift * g_shift.sign() * vertices[..., 3:4]
        history.append(loss.clone())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    history = [float(x) for x in history]
    if verbose:
        print('loss %.5f -> %.5f in %d graph launches, %.1f us per iteration (forward + loss + backward + update)' % (
            history[0], history[-1], steps, dt / max(1, steps - warm) * 1e6))
    return history


if __name__ == '__main__':
    if not torch.cuda.is_available():
        raise SystemExit('this example needs an MI355X (dirt_amd has no CPU path)')
    fit(torch.device('cuda', 0))
