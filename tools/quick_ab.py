"""A/B timing of the forward+backward step under flag sets, one process, several configurations (GPU box).

usage: python tools/quick_ab.py "K3 K3-2048" "0 0x10000" [dense|state|both] [steps]
Prints, per (config, flags, output mode): the event-timed step (median of 3 regions) and the library's per-kernel HIP-event
averages (raw readings: ~2 us above the kernel's own duration each).  DIRT_AMD_LIBRARY picks the build (tools/variants.sh)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dirt_amd import _lib, rasterise_ops as ops
from tests import scenes  # noqa: E402

configs = (sys.argv[1] if len(sys.argv) > 1 else 'K3').split()
flagsets = [int(x, 0) for x in (sys.argv[2] if len(sys.argv) > 2 else '0').split()]
modes = {'dense': ['dense'], 'state': ['state'], 'both': ['dense', 'state']}[sys.argv[3] if len(sys.argv) > 3 else 'dense']
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
dev = torch.device('cuda:0')
tag = os.path.basename(os.environ.get('DIRT_AMD_LIBRARY', 'product'))
for cfg in configs:
    F, H, W, C, seed, rlo, rhi = scenes.CONFIGS[cfg]
    spg = int(os.environ.get('SCENES', '1'))
    b = scenes.batch_scene(F, H, W, C, [seed + i for i in range(spg)], r_lo=rlo, r_hi=rhi)
    t = {k: torch.from_numpy(np.ascontiguousarray(b[k])).to(dev) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    for flags in flagsets:
        for mode in modes:
            def step(fl):
                px, st = ops._op_rasterise(t['background'], t['vertices'], t['vertex_colors'], t['faces'], H, W, C, flags=fl,
                                           keep_state=True, dense_grads=(mode == 'dense'))
                return ops._op_rasterise_grad(t['vertices'], t['faces'], px, t['grad_pixels'], H, W, C, flags=fl, state=st,
                                              state_outputs='dense' if mode == 'dense' else True)
            for _ in range(30):
                step(flags)
            regs = []
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(10):
                    step(flags)
                e0.record()
                for _ in range(steps):
                    step(flags)
                e1.record()
                e1.synchronize()
                regs.append(e0.elapsed_time(e1) / steps * 1e3)
            _lib.profile_reset()
            torch.cuda.synchronize()
            for _ in range(100):
                step(flags | _lib.FLAG_PROFILE)
            torch.cuda.synchronize()
            prof = {k.replace('_kernel', '').replace('<shade>', ''): round(ms / n * 1e3, 2) for k, (ms, n) in _lib.profile_read().items() if n}
            print('%-22s %-8s flags=%-8s %-5s step %.2f us (%.2f..%.2f)  %s' % (tag, cfg, hex(flags), mode, sorted(regs)[1], min(regs), max(regs), prof), flush=True)
