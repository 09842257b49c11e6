"""Per-wave phase timing of grad_kernel_stream (tracing build of the library: -DDIRT_TRACE, tools/build_tools.sh).
usage: python tools/trace_stream.py [config] [scenes]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dirt_amd import _lib, rasterise_ops as ops
from tests import scenes

cfg = sys.argv[1] if len(sys.argv) > 1 else 'K3'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lib = _lib.load()
F, H, W, C, seed, rlo, rhi = scenes.CONFIGS[cfg]
b = scenes.batch_scene(F, H, W, C, [seed + i for i in range(B)], r_lo=rlo, r_hi=rhi)
dev = torch.device('cuda:0')
t = {k: torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
ntiles = (W // 32) * (H // 32) * B
buf = torch.zeros(ntiles * 4 * 16, dtype=torch.int64, device=dev)
for it in range(3):
    px, state = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C, keep_state=True)
    if it == 2:
        lib.dirt_debug_set_trace_grad_stream(ctypes.c_void_p(buf.data_ptr()))
    ops._op_rasterise_grad(t['vertices'], t['faces'], px, t['grad_pixels'], H, W, C, state=state, flags=_lib.FLAG_GRAD_STREAM)
    torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 16)
a = a[a[:, 0] != 0]
NT = 9
tt = a[:, :NT].astype(np.float64)
d = np.diff(tt, axis=1)
names = ['addresses + issue slice 0', 'wait slice 0', 'issue slice 1 + slice 0 work', 'wait slice 1', 'slice 1 work', 'positions + ring', 'face loop', 'gbk stores']
print('%s x %d: %d waves; clocks per wave (s_memtime), mean / median / max' % (cfg, B, len(a)))
for i, n in enumerate(names):
    print('  %-30s %9.0f %9.0f %9.0f' % (n, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
tot = tt[:, NT - 1] - tt[:, 0]
print('  %-30s %9.0f %9.0f %9.0f' % ('total', tot.mean(), np.median(tot), tot.max()))
print('  ring cells per wave: mean %.1f max %d;  face-loop iterations per wave: mean %.2f max %d; clocks per iteration %.0f' % (
    a[:, 12].mean(), a[:, 12].max(), a[:, 13].mean(), a[:, 13].max(), d[:, 6].sum() / max(1, a[:, 13].sum())))
w0 = a[:, 14].astype(np.float64); w1 = w0 + (a[:, 15] >> 20).astype(np.float64)
t0 = w0.min()
st, en = (w0 - t0) / 100.0, (w1 - t0) / 100.0
print('  wall clock (100 MHz): wave starts p50 %.2f p90 %.2f max %.2f us; ends p10 %.2f p50 %.2f p90 %.2f max %.2f us; wave duration mean %.2f us' % (
    *np.percentile(st, [50, 90, 100]), *np.percentile(en, [10, 50, 90, 100]), (en - st).mean()))
clk = tot.sum() / ((en - st).sum())   # clocks per us
print('  clocks per us: %.0f' % clk)
# when (global clock) each phase boundary is crossed
for i in range(1, NT):
    at = st + (tt[:, i] - tt[:, 0]) / clk
    print('    marker %d (%-30s done): p10 %.2f p50 %.2f p90 %.2f max %.2f us' % (i, names[i - 1], *np.percentile(at, [10, 50, 90, 100])))
if len(a) == 4096:
    grp = (np.arange(4096) // 4) // 256
    print('  by dispatch group (blocks 0-255, 256-511, ...): start us | ' + ' | '.join(n[:14] for n in names) + ' | total | end us')
    for g in range(4):
        m = grp == g
        print('    group %d: %5.2f | ' % (g, st[m].mean()) + ' | '.join('%6.0f' % d[m, i].mean() for i in range(len(names))) + ' | %6.0f | %5.2f' % (tot[m].mean(), en[m].mean()))
