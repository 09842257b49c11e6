"""Host-hygiene soak on an MI355X: many autograd steps, on fresh streams now and then, with flat device memory.

    python tools/soak.py [steps=10000]

Asserts that torch.cuda.memory_reserved() does not grow after the first few hundred steps of a plain training loop (every
forward allocates a state workspace; torch's caching allocator must recycle them), that the module-level scratch cache
stays bounded when streams come and go (`rasterise_ops._workspaces`, least recently used out) with allocated memory
flat, and prints the host-side rate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from dirt_amd import rasterise_ops as ops
from tests import scenes  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    dev = torch.device('cuda:0')
    s = scenes.rand_scene(2000, 256, 256, 4, 5, 0.01, 0.08)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    bg, v, vc = (t(s[k]).requires_grad_(True) for k in ('background', 'vertices', 'vertex_colors'))
    f, g = t(s['faces']), t(s['grad_pixels'])
    def one_step():
        px = ops.rasterise(bg, v, vc, f)
        px.backward(g)
        bg.grad = v.grad = vc.grad = None

    # ---- phase 1: the training loop proper, one stream: RESERVED memory must be flat ----
    reserved, t0 = None, time.time()
    for i in range(steps):
        one_step()
        if i == 300:
            torch.cuda.synchronize()
            reserved = torch.cuda.memory_reserved(dev)
    torch.cuda.synchronize()
    rate = steps / (time.time() - t0)
    end = torch.cuda.memory_reserved(dev)
    assert reserved is None or end <= reserved, 'reserved memory grew: %d -> %d bytes' % (reserved, end)

    # ---- phase 2: short-lived streams (inference + visibility on each leave a scratch buffer behind): the module's scratch
    #      cache stays bounded and ALLOCATED memory flat.  (Reserved memory is torch's business here: its caching allocator
    #      keeps a 20 MB segment per stream that ever allocated, up to its pool of 32 streams per device.) ----
    torch.cuda.synchronize()
    allocated = torch.cuda.memory_allocated(dev)
    for i in range(64):
        st = torch.cuda.Stream(device=dev)
        st.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(st), torch.no_grad():
            ops.rasterise(bg, v, vc, f)
            ops._op_visibility(v[None], f[None], 256, 256)
        torch.cuda.current_stream(dev).wait_stream(st)
        one_step()
        assert len(ops._workspaces) <= ops._WORKSPACE_SLOTS
    torch.cuda.synchronize()
    scratch = sum(w.numel() for w in ops._workspaces.values())
    assert torch.cuda.memory_allocated(dev) <= allocated + scratch + (1 << 20), (allocated, torch.cuda.memory_allocated(dev), scratch)
    print('soak ok: %d steps at %.0f steps/s, reserved %.1f MB at step 300 and %.1f MB at the end; 64 extra streams: %d cached scratch '
          'buffers (%.1f MB), allocated %.1f -> %.1f MB'
          % (steps, rate, (reserved or 0) / 1e6, end / 1e6, len(ops._workspaces), scratch / 1e6, allocated / 1e6,
             torch.cuda.memory_allocated(dev) / 1e6))


if __name__ == '__main__':
    main()
