"""Host-side cost of one eager step (the two raw ops bench.py times), by function: cProfile over N steps.
usage (GPU box): python tools/profile_host.py [steps]"""
import cProfile, pstats, sys, os, time
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
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
for _ in range(100): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print('enqueue %.1f us/step (no profiler)' % ((t1 - t0) / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(n): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats('tottime')
st.print_stats(22)
