"""Per-wave phase timing of grad_kernel_px2 (tracing build of the library: -DDIRT_TRACE).  usage: python tools/trace_px2.py [config] [flags]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dirt_amd import _lib, rasterise_ops as ops
from tests import scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else 'K3'
flags = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0x8000
lib = _lib.load()
F, H, W, C, seed, rlo, rhi = scenes.CONFIGS[cfg]
s = scenes.rand_scene(F, H, W, C, seed, rlo, rhi)
dev = torch.device('cuda:0')
t = {k: torch.from_numpy(np.ascontiguousarray(s[k]))[None].to(dev) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
nwg = ((W + 31) // 32) * ((H + 15) // 16)
buf = torch.zeros(nwg * 4 * 16, dtype=torch.int64, device=dev)
for it in range(3):
    px, state = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C, keep_state=True, dense_grads=True)
    if it == 2:
        lib.dirt_debug_set_trace_grad_px2(ctypes.c_void_p(buf.data_ptr()))
    ops._op_rasterise_grad(t['vertices'], t['faces'], px, t['grad_pixels'], H, W, C, flags=flags, state=state, state_outputs='dense')
    torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 16)
a = a[a[:, 0] != 0]
tt = a[:, :9].astype(np.float64)
d = np.diff(tt, axis=1)
names = ['issue loads + state tile', 'store planes', 'barrier', 'Scharr', 'dilation', 'positions + ring', 'face loop', 'gbk stores']
print('%s px2: %d waves; clocks per wave (s_memtime), mean / median / max' % (cfg, len(a)))
for i, n in enumerate(names):
    print('  %-26s %9.0f %9.0f %9.0f' % (n, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
tot = tt[:, 8] - tt[:, 0]
print('  %-26s %9.0f %9.0f %9.0f' % ('total', tot.mean(), np.median(tot), tot.max()))
print('  ring cells per wave: mean %.1f max %d;  face-loop iterations per wave: mean %.2f max %d;  clocks per iteration %.0f' % (
    a[:, 12].mean(), a[:, 12].max(), a[:, 13].mean(), a[:, 13].max(), d[:, 6].sum() / max(1, a[:, 13].sum())))
w0 = a[:, 14].astype(np.float64); w1 = w0 + (a[:, 15] >> 20).astype(np.float64)
t0 = w0.min()
st = (w0 - t0) / 100.0; en = (w1 - t0) / 100.0
print('  wall clock: wave starts p50 %.2f p75 %.2f p90 %.2f max %.2f us; ends p10 %.2f p50 %.2f p90 %.2f max %.2f us; wave duration mean %.2f us' % (
    *np.percentile(st, [50, 75, 90, 100]), *np.percentile(en, [10, 50, 90, 100]), (en - st).mean()))
# by start-time cohort: the first round (resident at launch) and what follows
first = st < 1.5
print('  waves started within 1.5 us: %d (%.0f%%): duration mean %.2f us, load wait (start -> barrier passed) %.0f clocks; later waves: duration %.2f us, load wait %.0f clocks' % (
    first.sum(), 100 * first.mean(), (en - st)[first].mean(), (tt[first, 3] - tt[first, 0]).mean(), (en - st)[~first].mean() if (~first).any() else 0,
    (tt[~first, 3] - tt[~first, 0]).mean() if (~first).any() else 0))
hist, edges = np.histogram(st, bins=np.arange(0, st.max() + 2, 2.0))
print('  wave starts per 2 us bin:', list(hist))
hist, edges = np.histogram(en, bins=np.arange(0, en.max() + 2, 2.0))
print('  wave ends per 2 us bin:  ', list(hist))
