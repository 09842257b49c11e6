"""GPU box: the fused texture look-up (dirt_texture.hip; samples/textured.py:16-61) at 2048 x 2048 x Ct = 3 over a 512 x 512
texture -- forward and backward kernel times (HIP events), algorithmic GB/s and the fraction of the 8 TB/s HBM peak.  The
(u, v) field is what a G-buffer holds: smooth, rotated, `scale` texture tiles across the frame (scale 4: ~1 texel per pixel;
1: 4 x 4 pixels per texel), read in place from channels 1:3 of a 6-channel buffer (samples/textured.py:120-122).
usage: python tools/bench_texture.py [H W Ht Wt Ct] [json path]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dirt_amd import texture  # noqa: E402

args = [a for a in sys.argv[1:] if not a.endswith('.json')]
out_path = next((a for a in sys.argv[1:] if a.endswith('.json')), None)
H, W, Ht, Wt, Ct = (int(a) for a in args[:5]) if len(args) >= 5 else (2048, 2048, 512, 512, 3)
dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
tex = torch.from_numpy(rng.uniform(0, 1, (Ht, Wt, Ct)).astype(np.float32)).to(dev)
g = torch.from_numpy(rng.standard_normal((H, W, Ct)).astype(np.float32)).to(dev)
ys, xs = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing='ij')
results = []
for scale in (4.0, 1.0):
    c, s = np.cos(0.2), np.sin(0.2)
    u = (c * xs / W + s * ys / H) * scale + 0.13
    v = (-s * xs / W + c * ys / H) * scale + 0.41
    gbuf = np.zeros((H, W, 6), np.float32)
    gbuf[..., 1], gbuf[..., 2] = u, v
    gb = torch.from_numpy(gbuf).to(dev)

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / n * 1e3   # us

    t_fwd = timed(lambda: texture.sample_texture_uv(tex, gb[..., 1:3]))
    tl = tex.clone().requires_grad_(True)
    gl = gb.clone().requires_grad_(True)

    def fwd_bwd():
        tl.grad = gl.grad = None
        texture.sample_texture_uv(tl, gl[..., 1:3]).backward(g)
    t_both = timed(fwd_bwd)
    # (the autograd round trip also zero-fills gl.grad's six channels and clears grad_texture: the C ABI call alone is timed below)
    from dirt_amd import _lib, rasterise_ops as _ops
    lib = _lib.load()
    gt = torch.empty_like(tex)
    guv = torch.empty((H, W, 2), dtype=torch.float32, device=dev)
    src = gb[..., 1:3]

    def bwd_only():   # the look-ups as an H x W image: 16 x 16-pixel tiles (what dirt_amd.texture's autograd node calls)
        rc = lib.dirt_texture_sample_backward_image(tex.data_ptr(), src.data_ptr(), g.data_ptr(), gt.data_ptr(), guv.data_ptr(), H, W, Ht, Wt, Ct, 6, 2, 0,
                                                    _ops._stream_handle(dev))
        assert rc == 0

    def bwd_flat():   # ... as a flat list of H W look-ups: runs of 256
        rc = lib.dirt_texture_sample_backward(tex.data_ptr(), src.data_ptr(), g.data_ptr(), gt.data_ptr(), guv.data_ptr(), H * W, Ht, Wt, Ct, 6, 2, 0,
                                              _ops._stream_handle(dev))
        assert rc == 0
    t_bwd = timed(bwd_only)
    t_bwd_flat = timed(bwd_flat)
    n = H * W
    fwd_bytes = n * (8 + 4 * Ct) + Ht * Wt * Ct * 4
    bwd_bytes = n * (8 + 4 * Ct + 8) + 2 * Ht * Wt * Ct * 4
    r = {'scale': scale, 'pixels_per_texel': (H / (Ht * scale)) * (W / (Wt * scale)), 'forward_us': t_fwd, 'backward_us': t_bwd, 'backward_flat_list_us': t_bwd_flat, 'autograd_fwd_bwd_us': t_both,
         'forward_GBps': fwd_bytes / t_fwd / 1e3, 'backward_GBps': bwd_bytes / t_bwd / 1e3,
         'forward_frac_of_8TBps': fwd_bytes / t_fwd / 1e3 / 8000, 'backward_frac_of_8TBps': bwd_bytes / t_bwd / 1e3 / 8000}
    results.append(r)
    print(json.dumps(r))
if out_path:
    json.dump({'workload': 'texture look-up %dx%d pixels, texture %dx%dx%d, (u, v) in place from a 6-channel G-buffer' % (H, W, Ht, Wt, Ct), 'results': results},
              open(out_path, 'w'), indent=1)
