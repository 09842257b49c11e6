"""Times the forward (raster) and backward kernels of a configuration under the tile-shape flags (HIP-event averages
of the library's own per-kernel profile).  usage: python tools/time_variants.py [config]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dirt_amd import _lib, rasterise_ops as ops
from tests import scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else 'K3'
F, H, W, C, seed, rlo, rhi = scenes.CONFIGS[cfg]
s = scenes.rand_scene(F, H, W, C, seed, rlo, rhi)
dev = torch.device('cuda:0')
t = {k: torch.from_numpy(np.ascontiguousarray(s[k]))[None].to(dev) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
for name, flags in (('auto', 0), ('large tiles', _lib.FLAG_TILES_LARGE), ('small tiles', _lib.FLAG_TILES_SMALL)):
    def step(fl):
        px, state = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C, flags=fl, keep_state=True)
        ops._op_rasterise_grad(t['vertices'], t['faces'], px, t['grad_pixels'], H, W, C, flags=fl, state=state)
    for _ in range(20):
        step(flags)
    _lib.profile_reset()
    torch.cuda.synchronize()
    for _ in range(100):
        step(flags | _lib.FLAG_PROFILE)
    torch.cuda.synchronize()
    print(cfg, name, {k: round(ms / n * 1e3, 1) for k, (ms, n) in _lib.profile_read().items() if n})
