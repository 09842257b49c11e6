"""Per-wave phase timing of grad_kernel (tracing build of the library: -DDIRT_TRACE, see tools/build_tools.sh).
usage: python tools/trace_grad.py [config]"""
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
t = {k: torch.from_numpy(np.ascontiguousarray(s[k]))[None].to(dev) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
ntiles = max(((W + 31) // 32) * ((H + 31) // 32), ((W + 15) // 16) * ((H + 15) // 16))   # 16 x 16 tiles: the small-frame kernel
buf = torch.zeros(ntiles * 4 * 16, dtype=torch.int64, device=dev)
for it in range(3):
    px, state = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C, keep_state=True)
    if it == 2:
        lib.dirt_debug_set_trace_grad(ctypes.c_void_p(buf.data_ptr()))
        lib.dirt_debug_set_trace_grad_small(ctypes.c_void_p(buf.data_ptr()))
    ops._op_rasterise_grad(t['vertices'], t['faces'], px, t['grad_pixels'], H, W, C, state=state)
    torch.cuda.synchronize()
a = buf.cpu().numpy().reshape(-1, 16)
a = a[a[:, 0] != 0]   # (the buffer is sized for the larger of the two tilings)
NT = 10 if (a[:, 9] != 0).any() else 8   # (10 timestamps since the segment-merging face loop)
tt = a[:, :NT].astype(np.float64)
d = np.diff(tt, axis=1)
names = ['issue loads + state tile', 'store planes', 'barrier', 'Scharr', 'dilation', 'list + roles', 'face loop']
if NT == 10:
    names = names[:6] + ['segments', 'iterations', 'stores + flush']
print('%s: %d waves; clocks per wave (s_memtime), mean / median / max' % (cfg, len(a)))
for i, n in enumerate(names):
    print('  %-26s %9.0f %9.0f %9.0f' % (n, d[:, i].mean(), np.median(d[:, i]), d[:, i].max()))
tot = tt[:, NT - 1] - tt[:, 0]
print('  %-26s %9.0f %9.0f %9.0f' % ('total', tot.mean(), np.median(tot), tot.max()))
print('  dilated pairs listed per wave: mean %.1f max %d;  face-loop iterations per wave: mean %.1f max %d' % (a[:, 12].mean(), a[:, 12].max(), a[:, 13].mean(), a[:, 13].max()))
print('  face loop clocks per iteration: %.0f' % (d[:, 7 if NT == 10 else 6].sum() / max(1, a[:, 13].sum())))
print('  kernel span (first start .. last end, clocks; not comparable across CUs): %d' % (tt[:, NT - 1].max() - tt[:, 0].min()))

w0 = a[:, 14].astype(np.float64); w1 = w0 + (a[:, 15] >> 20).astype(np.float64)
t0 = w0.min()
print('  wall clock (100 MHz): wave starts %.2f .. %.2f us after the first; ends %.2f .. %.2f us; wave duration mean %.2f us' % (
    0.0, (w0.max() - t0) / 100.0, (w1.min() - t0) / 100.0, (w1.max() - t0) / 100.0, ((w1 - w0).mean()) / 100.0))
st = np.sort((w0 - t0) / 100.0)
print('  start-time percentiles (us): 50%% %.2f  75%% %.2f  90%% %.2f  99%% %.2f' % tuple(np.percentile(st, [50, 75, 90, 99])))
hw = (a[:, 15] & 0xFFFFF).astype(np.int64)
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3
blk = np.arange(len(a)) // 4
key = se * 32 + sh * 16 + cu
print('  placement (XCD 0 = blocks 0, 8, 16, ...): block -> (se, sh, cu) of its first wave')
sel = [b for b in range(0, min(8 * 140, len(a) // 4), 8)]
print('   ', ' '.join('%d:%d.%d.%d' % (b, se[4 * b], sh[4 * b], cu[4 * b]) for b in sel[:48]))
from collections import defaultdict
g = defaultdict(list)
for b in range(0, len(a) // 4, 8):
    g[int(key[4 * b])].append(b)
print('    blocks sharing a CU on XCD 0:', [v for v in list(g.values())[:6]])
print('    SIMD of the 4 waves of block 0:', simd[0:4])
# how much of the spread of wave times is work (faces per region) and how much is placement
it = a[:, 13].astype(np.float64)
print('  corr(wave total, face-loop iterations) = %.2f;  totals: p50 %.0f p90 %.0f p99 %.0f max %.0f' % (
    np.corrcoef(tot, it)[0, 1], *np.percentile(tot, [50, 90, 99, 100])))
wg_it = it.reshape(-1, 4).sum(1); wg_end = tt[:, NT - 1].reshape(-1, 4).max(1) - tt[:, 0].reshape(-1, 4).min(1)
print('  per workgroup: corr(duration, iterations) = %.2f' % np.corrcoef(wg_end, wg_it)[0, 1])
if len(wg_it) == 1024:
    cu_it = wg_it.reshape(4, 256).sum(0); cu_end = wg_end.reshape(4, 256).max(0)
    print('  per CU (blocks b, b+256, b+512, b+768): iterations mean %.0f max %.0f (+%.0f%%); slowest workgroup mean %.0f max %.0f; corr %.2f' % (
        cu_it.mean(), cu_it.max(), 100 * (cu_it.max() / cu_it.mean() - 1), cu_end.mean(), cu_end.max(), np.corrcoef(cu_end, cu_it)[0, 1]))

# which phase makes a wave slow, and when (global clock) the waves end
for i, n in enumerate(names):
    print('  corr(total, %-24s) = %5.2f   p90 - p50 of the phase: %6.0f clocks' % (n, np.corrcoef(tot, d[:, i])[0, 1], np.percentile(d[:, i], 90) - np.percentile(d[:, i], 50)))
en = (w1 - t0) / 100.0
print('  end times (us after the first wave start): p10 %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f' % tuple(np.percentile(en, [10, 50, 90, 99, 100])))
wg_last = en.reshape(-1, 4).max(1)
if len(wg_last) == 1024:
    tiles_x = (W + 31) // 32
    # position of the tile inside its XCD band (xcd_tile: XCD x owns tiles x * 128 .. x * 128 + 127 = four tile rows)
    b = np.arange(1024); tile = (b & 7) * 128 + (b >> 3)
    row_in_band = (tile // tiles_x) % 4
    for r in range(4):
        print('    tile row %d of the XCD band: workgroup end mean %.2f us' % (r, wg_last[row_in_band == r].mean()))
    order = np.argsort(wg_last)[-12:]
    print('    last workgroups (block, tile, end us, iterations):', [(int(x), int(tile[x]), round(float(wg_last[x]), 2), int(wg_it[x])) for x in order])

# phase means by dispatch group (blocks b, b+256, b+512, b+768 share a CU; earlier groups are dispatched first)
if len(a) == 4096:
    grp = (np.arange(4096) // 4) // 256
    st = (w0 - t0) / 100.0
    print('  by dispatch group (blocks 0-255, 256-511, ...): start us | ' + ' | '.join(names) + ' | total clocks | end us')
    for gidx in range(4):
        m = grp == gidx
        print('    group %d: %5.2f | ' % (gidx, st[m].mean()) + ' | '.join('%6.0f' % d[m, i].mean() for i in range(len(names))) + ' | %6.0f | %5.2f' % (tot[m].mean(), en[m].mean()))
