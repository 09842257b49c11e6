"""Where a short timing's extra microseconds go: K steps between two synchronisations, host and GPU clocks side by side.
usage: python tools/step_ramp.py [K]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dirt_amd import _lib, rasterise_ops as ops
from tests import scenes
_lib.load()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
F, H, W, C, seed0, r_lo, r_hi = scenes.CONFIGS['K3']
b = scenes.batch_scene(F, H, W, C, [seed0], r_lo=r_lo, r_hi=r_hi)
dev = torch.device('cuda:0')
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
bg, v, vc, f, g = (t(b[k]) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels'))
def step():
    px, state = ops._op_rasterise(bg, v, vc, f, H, W, C, keep_state=True)
    return ops._op_rasterise_grad(v, f, px, g, H, W, C, state=state)
for _ in range(50): step()
for rep in range(4):
    e0, e1, done = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if rep >= 2: e0.record()
    step()
    t1 = time.perf_counter()
    for i in range(K - 1): step()
    t2 = time.perf_counter()
    if rep >= 2: e1.record()
    done.record()
    while not done.query(): pass
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    gpu = e0.elapsed_time(e1) * 1e3 if rep >= 2 else float('nan')
    print('K=%d: first step issued after %.0f us, all after %.0f us, GPU done (polled) at %.0f us, synchronize returned at %.0f us = %.2f us/step; GPU clock between first and last launch: %.0f us' % (
        K, (t1 - t0) * 1e6, (t2 - t0) * 1e6, (t3 - t0) * 1e6, (t4 - t0) * 1e6, (t4 - t0) / K * 1e6, gpu))
