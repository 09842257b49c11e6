"""Host-side profile of the autograd path: dirt_amd.rasterise_batch(...).backward() on the K3 scene (cProfile).
usage (GPU box): python tools/profile_autograd.py [steps]"""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dirt_amd import rasterise_ops as ops
from tests import scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
F, H, W, C, seed, lo, hi = scenes.CONFIGS['K3']
b = scenes.batch_scene(F, H, W, C, [seed], r_lo=lo, r_hi=hi)
dev = torch.device('cuda:0')
bg, v, vc, f, g = (torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels'))
bg_l, v_l, vc_l = (x.clone().requires_grad_(True) for x in (bg, v, vc))

def step():
    bg_l.grad = v_l.grad = vc_l.grad = None
    ops.rasterise_batch(bg_l, v_l, vc_l, f, H, W, C).backward(g)

def raw():
    px, st = ops._op_rasterise(bg, v, vc, f, H, W, C, keep_state=True)
    ops._op_rasterise_grad(v, f, px, g, H, W, C, state=st)

for fn, name in ((raw, 'raw ops'), (step, 'autograd')):
    for _ in range(50): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print('%-9s host issue %.1f us/step, with GPU drain %.1f us/step' % (name, t_issue / n * 1e6, t_all / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(n): step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print(s.getvalue()[:6000])
