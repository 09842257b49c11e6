"""Per-wave phase timing of setup_kernel (tracing build of the library, tools/build_tools.sh).  usage: python tools/trace_setup.py [config]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dirt_amd import _lib, rasterise_ops as ops
from tests import scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else 'K3'
lib = _lib.load()
F, H, W, C, seed, rlo, rhi = scenes.CONFIGS[cfg]
s = scenes.rand_scene(F, H, W, C, seed, rlo, rhi)
dev = torch.device('cuda:0')
t = {k: torch.from_numpy(np.ascontiguousarray(s[k]))[None].to(dev) for k in ('background', 'vertices', 'vertex_colors', 'faces')}
buf = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
for it in range(4):
    if it == 3:
        lib.dirt_debug_set_trace_setup(ctypes.c_void_p(buf.data_ptr()))
    px, state = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C, keep_state=True)
    torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 16)
a = a[a[:, 0] != 0]
d = np.diff(a[:, :6].astype(np.float64), axis=1)
names = ['zero side job + clear + barrier', 'pass 1: loads, set-up, record store, histogram', 'barrier', 'prefix, directory stores, barriers', 'pass 2: entries']
print('%s: %d waves; clocks per wave, mean / median / max' % (cfg, len(a)))
for i, n in enumerate(names):
    print('  %-50s %8.0f %8.0f %8.0f' % (n, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
tot = a[:, 5] - a[:, 0]
print('  %-50s %8.0f %8.0f %8.0f' % ('total', tot.mean(), np.median(tot), tot.max()))
w0 = a[:, 8].astype(np.float64); w1 = w0 + a[:, 9]
print('  wall clock (100 MHz): wave starts 0 .. %.2f us; ends %.2f .. %.2f us; clocks per us %.0f' % (
    (w0.max() - w0.min()) / 100, (w1.min() - w0.min()) / 100, (w1.max() - w0.min()) / 100, tot.sum() / (a[:, 9].sum() / 100.0)))
