import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from dirt_amd import _lib, rasterise_ops as ops
from tests import scenes
_lib.load()
F, H, W, C, seed0, r_lo, r_hi = scenes.CONFIGS['K3']
b = scenes.batch_scene(F, H, W, C, [seed0], r_lo=r_lo, r_hi=r_hi)
dev = torch.device('cuda:0')
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bg, v, vc, f, g = (t(b[k]) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels'))
def step():
    px, state = ops._op_rasterise(bg, v, vc, f, H, W, C, keep_state=True)
    return ops._op_rasterise_grad(v, f, px, g, H, W, C, state=state)
for _ in range(50): step()
torch.cuda.synchronize()
for n in (200, 1000):
    t0 = time.perf_counter()
    for _ in range(n): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('n=%d: enqueue %.1f us/step, total %.1f us/step' % (n, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6))
