"""Per-wave phase timing of grad_kernel (tracing build of the library: -DDIRT_TRACE, see tools/trace_grad.sh).
usage: python tools/trace_grad.py [config]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dirt_amd import _lib, scenes, rasterise_ops as ops

cfg = sys.argv[1] if len(sys.argv) > 1 else 'K3'
lib = _lib.load()
F, H, W, C, seed, rlo, rhi = scenes.CONFIGS[cfg]
s = scenes.rand_scene(F, H, W, C, seed, rlo, rhi)
dev = torch.device('cuda:0')
t = {k: torch.from_numpy(np.ascontiguousarray(s[k]))[None].to(dev) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
ntiles = ((W + 31) // 32) * ((H + 31) // 32)
buf = torch.zeros(ntiles * 4 * 16, dtype=torch.int64, device=dev)
for it in range(3):
    px, state = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C, keep_state=True)
    if it == 2:
        lib.dirt_debug_set_trace_grad(ctypes.c_void_p(buf.data_ptr()))
    ops._op_rasterise_grad(t['vertices'], t['faces'], px, t['grad_pixels'], H, W, C, state=state)
    torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 16)
tt = a[:, :8].astype(np.float64)
d = np.diff(tt, axis=1)
names = ['issue loads + state tile', 'store planes', 'barrier', 'Scharr', 'dilation', 'list + roles', 'face loop']
print('%s: %d waves; clocks per wave (s_memtime), mean / median / max' % (cfg, len(a)))
for i, n in enumerate(names):
    print('  %-26s %9.0f %9.0f %9.0f' % (n, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
tot = tt[:, 7] - tt[:, 0]
print('  %-26s %9.0f %9.0f %9.0f' % ('total', tot.mean(), np.median(tot), tot.max()))
print('  dilated pairs listed per wave: mean %.1f max %d;  face-loop iterations per wave: mean %.1f max %d' % (a[:, 12].mean(), a[:, 12].max(), a[:, 13].mean(), a[:, 13].max()))
print('  face loop clocks per iteration: %.0f' % (d[:, 6].sum() / max(1, a[:, 13].sum())))
print('  kernel span (first start .. last end, clocks; not comparable across CUs): %d' % (tt[:, 7].max() - tt[:, 0].min()))
