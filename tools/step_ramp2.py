"""Where the wall-clock of a 20-step timed region goes (GPU box): host issue time, GPU time between events, the lag of the
first kernel behind the first call and of the host behind the last kernel.  usage: python tools/step_ramp2.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dirt_amd import rasterise_ops as ops
from tests import scenes

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device('cuda:0')
F, H, W, C, seed, rlo, rhi = scenes.CONFIGS['K3']
b = scenes.batch_scene(F, H, W, C, [seed], r_lo=rlo, r_hi=rhi)
bg, v, vc, f, g = (torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels'))


def step():
    px, st = ops._op_rasterise(bg, v, vc, f, H, W, C, keep_state=True, dense_grads=True)
    return ops._op_rasterise_grad(v, f, px, g, H, W, C, state=st, state_outputs='dense')


def barrier():
    done = torch.cuda.Event()
    done.record()
    while not done.query():
        pass
    torch.cuda.synchronize()


for _ in range(30):
    step()
for rep in range(6):
    for _ in range(5):
        step()
    barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    t_issue = time.perf_counter()
    e1.record()
    t_rec = time.perf_counter()
    while not e1.query():
        pass
    t_q = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    gpu = e0.elapsed_time(e1) * 1e3
    print('steps %d: wall %.1f us (%.2f/step) | host issue %.1f us (%.2f/step) | gpu e0->e1 %.1f us (%.2f/step) | record %.1f query-spin %.1f sync %.1f us' % (
        steps, (t1 - t0) * 1e6, (t1 - t0) * 1e6 / steps, (t_issue - t0) * 1e6, (t_issue - t0) * 1e6 / steps, gpu, gpu / steps,
        (t_rec - t_issue) * 1e6, (t_q - t_rec) * 1e6, (t1 - t_q) * 1e6))
