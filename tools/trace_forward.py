"""Per-wave phase timing of setup_kernel_v2 and raster_kernel_v2 (tracing build of the library: -DDIRT_TRACE, tools/build_tools.sh).
usage: python tools/trace_forward.py [config]"""
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
bs = torch.zeros(4096 * 16, dtype=torch.int64, device=dev)
br = torch.zeros(4 * 4096 * 8 * 16, dtype=torch.int64, device=dev)   # (up to eight waves per workgroup)
for it in range(4):
    if it == 3:
        lib.dirt_debug_set_trace_forward(ctypes.c_void_p(bs.data_ptr()), ctypes.c_void_p(br.data_ptr()))
    px, state = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C, keep_state=True)
    torch.cuda.synchronize()


def report(buf, nt, names, title):
    a = buf.cpu().numpy().reshape(-1, 16)
    a = a[a[:, 0] != 0]
    tt = a[:, :nt].astype(np.float64)
    d = np.diff(tt, axis=1)
    print('%s %s: %d waves; clocks per wave, mean / median / max' % (cfg, title, len(a)))
    for i, n in enumerate(names):
        print('  %-46s %8.0f %8.0f %8.0f' % (n, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
    tot = tt[:, nt - 1] - tt[:, 0]
    print('  %-46s %8.0f %8.0f %8.0f' % ('total', tot.mean(), np.median(tot), tot.max()))
    w0 = a[:, 12].astype(np.float64); w1 = w0 + a[:, 13]; t0 = w0.min()
    st, en = (w0 - t0) / 100.0, (w1 - t0) / 100.0
    clk = tot.sum() / (en - st).sum()
    print('  wall clock: wave starts p50 %.2f max %.2f us; ends p10 %.2f p50 %.2f p90 %.2f max %.2f us; clocks per us %.0f' % (
        *np.percentile(st, [50, 100]), *np.percentile(en, [10, 50, 90, 100]), clk))
    for i in range(1, nt):
        at = st + (tt[:, i] - tt[:, 0]) / clk
        print('    marker %d (%-40s): p10 %.2f p50 %.2f p90 %.2f max %.2f us' % (i, names[i - 1][:40], *np.percentile(at, [10, 50, 90, 100])))
    return a


report(bs, 6, ['requests issued', 'indices / vertices there (+ second trip)', 'edge functions -> set-up record in LDS', 'barrier, records out, barrier',
               'coverage records + directory stores'], 'setup_kernel_v2 (wave 0 of each chunk)')
a = report(br, 10, ['cells requested, side job', 'barrier', 'wait cells, claim slots, list', 'barrier', 'coverage records: DMA + wait', 'barrier',
                    'shade DMA issue + coverage loop', 'wait shade data + barrier', 'shade + stores'], 'raster_kernel_v2')
print('  candidates visited per wave: mean %.1f max %d' % (a[:, 15].mean(), a[:, 15].max()))
