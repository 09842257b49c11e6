"""One configuration's rasterise_deferred forward + backward, a few steps, for rocprofv3 --kernel-trace --stats (GPU box):
which kernels a deferred step is made of (dirt/rasterise_ops.py:204-237).
usage: rocprofv3 --kernel-trace --stats -d out -o x --output-format csv -- python tools/prof_deferred.py [config] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dirt_amd import rasterise_ops as ops  # noqa: E402
from tests import scenes  # noqa: E402

config = sys.argv[1] if len(sys.argv) > 1 else 'K5'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device('cuda', 0)
F, H, W, C, seed, r_lo, r_hi = scenes.CONFIGS[config]
s = scenes.rand_scene(F, H, W, C, seed, r_lo, r_hi)
bg, v, a = (torch.from_numpy(s[k]).to(dev).requires_grad_(True) for k in ('background', 'vertices', 'vertex_colors'))
f = torch.from_numpy(s['faces']).to(dev)
d = torch.from_numpy(np.random.default_rng(0).standard_normal((H, W, 3)).astype(np.float32)).to(dev)


def shader(g):
    return g[..., :3] * g[..., 3:4]


for _ in range(steps):
    bg.grad = v.grad = a.grad = None
    ops.rasterise_deferred(bg, v, a, f, shader).backward(d)
torch.cuda.synchronize()
print('prof_deferred: %d steps of %s' % (steps, config))
