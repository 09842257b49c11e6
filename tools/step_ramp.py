"""Per-step GPU times of the first steps after a synchronisation (why a 20-step timing costs more per step than a 200-step one)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dirt_amd import scenes, _lib, rasterise_ops as ops
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
K = 20
for rep in range(3):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(K):
        step(); ev[i + 1].record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    d = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
    print('wall %.1f us/step (enqueue done at %.0f us, sync returned at %.0f us); GPU event deltas (us):' % ((t2 - t0) / K * 1e6, (t1 - t0) * 1e6, (t2 - t0) * 1e6), ' '.join('%.0f' % x for x in d), ' sum %.0f' % sum(d))
