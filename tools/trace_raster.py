"""Per-wave phase timing of raster_kernel (tracing build of the library: -DDIRT_TRACE, tools/build_tools.sh builds it).
usage: python tools/trace_raster.py [config]"""
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
nwaves = 4 * 4096 * 4
buf = torch.zeros(nwaves * 16, dtype=torch.int64, device=dev)
for it in range(3):
    if it == 2:
        lib.dirt_debug_set_trace(ctypes.c_void_p(buf.data_ptr()))
    px, state = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C, keep_state=True)
    torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 16)
a = a[a[:, 0] != 0]
tt = a[:, :7].astype(np.float64)
d = np.diff(tt, axis=1)
names = ['request directory cells', 'clear + barrier', 'scan: cells -> entries -> list', 'barrier (list built)', 'candidates', 'shade + store']
print('%s: %d waves; clocks per wave (single-round tiles), mean / median / max' % (cfg, len(a)))
for i, n in enumerate(names):
    print('  %-34s %9.0f %9.0f %9.0f' % (n, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
for n, col in (('  of candidates: staging', 8), ('  of candidates: loop', 9), ('  of candidates: barrier', 10)):
    print('  %-34s %9.0f %9.0f %9.0f' % (n, a[:, col].mean(), np.median(a[:, col]), a[:, col].max()))
tot = tt[:, 6] - tt[:, 0]
print('  %-34s %9.0f %9.0f %9.0f' % ('total', tot.mean(), np.median(tot), tot.max()))
print('  candidates visited per wave: mean %.1f max %d' % (a[:, 11].mean(), a[:, 11].max()))

# global clock (100 MHz): when the waves start and end, and how that goes with the order in which a CU got its workgroups
w0 = a[:, 12].astype(np.float64); w1 = w0 + a[:, 13].astype(np.float64); t0 = w0.min()
st, en = (w0 - t0) / 100.0, (w1 - t0) / 100.0
print('  wave starts (us): p50 %.2f p99 %.2f max %.2f;  ends: p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f;  mean duration %.2f us' % (
    *np.percentile(st, [50, 99, 100]), *np.percentile(en, [10, 50, 90, 99, 100]), (en - st).mean()))
print('  s_memtime clocks per us of global clock: %.0f' % (tot.sum() / (en - st).sum()))
blk = a[:, 14].astype(np.int64)
if blk.max() == 1023:
    for r in range(4):
        sel = (blk >> 3) // 32 == r
        print('    workgroups %d of a CU (blocks with (b / 8) / 32 == %d): end mean %.2f us, candidates visited per wave %.1f' % (r + 1, r, en[sel].mean(), a[sel, 11].mean()))
for i, n in enumerate(names):
    print('  corr(total, %-32s) = %5.2f' % (n, np.corrcoef(tot, d[:, i])[0, 1]))
