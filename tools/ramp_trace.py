"""Where do the extra microseconds of a SHORT timed region go?  (`bench.py --steps 20` reads ~3.5 us per step more GPU time than
--steps 200.)  Run under `rocprofv3 --kernel-trace`: warm-up, then `regions` times { synchronize, `steps` steps }, then a long
region; `python tools/ramp_trace.py analyse <kernel_trace.csv>` then prints, per position of a step inside its region, the
kernels' durations and the gaps in front of them, averaged over the regions.
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/ramp_trace.py run [steps] [regions]"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(steps, regions):
    import numpy as np
    import torch
    from dirt_amd import rasterise_ops as ops
    from tests import scenes
    dev = torch.device('cuda:0')
    s = scenes.config_scene('K3')
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    H, W, C = s['height'], s['width'], s['channels']
    bg, v, vc, f, g = (t(s[k][None]) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels'))

    def step():
        px, state = ops._op_rasterise(bg, v, vc, f, H, W, C, keep_state=True, dense_grads=True)
        return ops._op_rasterise_grad(v, f, px, g, H, W, C, state=state, state_outputs='dense')

    for _ in range(300):
        step()
    for _ in range(regions):
        torch.cuda.synchronize()
        for _ in range(steps):
            step()
    torch.cuda.synchronize()
    for _ in range(400):
        step()
    torch.cuda.synchronize()


def analyse(path, steps):
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            name = r.get('Kernel_Name') or r.get('Name') or ''
            if 'dirt::' not in name:
                continue
            kind = 'grad' if 'grad_kernel' in name else ('raster' if 'raster_kernel' in name else ('setup' if 'setup_kernel' in name else None))
            if kind:
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), kind))
    rows.sort()
    # regions: a gap of more than 25 us in front of a set-up kernel
    regs, cur, prev_end = [], [], None
    for st, en, kind in rows:
        if prev_end is not None and kind == 'setup' and st - prev_end > 25000:
            regs.append(cur)
            cur = []
        cur.append((st, en, kind))
        prev_end = en
    regs.append(cur)
    short = [r for r in regs if len(r) == 3 * steps]
    longr = max(regs, key=len)
    print('%d regions, %d of %d steps; longest region %d steps' % (len(regs), len(short), steps, len(longr) // 3))

    def per_step(reg):
        out = []
        for i in range(0, len(reg) - 2, 3):
            (s0, e0, _), (s1, e1, _), (s2, e2, _) = reg[i:i + 3]
            nxt = reg[i + 3][0] if i + 3 < len(reg) else None
            out.append((e0 - s0, s1 - e0, e1 - s1, s2 - e1, e2 - s2, (nxt - e2) if nxt else 0, (nxt - s0) if nxt else 0))
        return out
    import numpy as np
    a = np.array([per_step(r) for r in short], dtype=np.float64) / 1e3      # [region, step, 7] us
    m = a.mean(0)
    print('step: setup | gap | raster | gap | grad | gap to next setup | step period   (us, mean over regions)')
    for i in range(steps):
        print('%3d: %6.2f | %5.2f | %6.2f | %5.2f | %6.2f | %5.2f | %6.2f' % ((i,) + tuple(m[i])))
    lp = np.array(per_step(longr), dtype=np.float64)[50:-1] / 1e3
    print('steady state (long region, steps 50..): %s' % ' | '.join('%6.2f' % x for x in lp.mean(0)))


if __name__ == '__main__':
    if sys.argv[1] == 'run':
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 20, int(sys.argv[3]) if len(sys.argv) > 3 else 20)
    else:
        analyse(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 20)
