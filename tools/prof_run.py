import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dirt_amd import rasterise_ops as ops
from tests import scenes
dev = torch.device('cuda:0')
cfg = sys.argv[1] if len(sys.argv) > 1 else 'K3'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
flags = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0   # DIRT_FLAG_* bits (kernel shapes)
s = scenes.config_scene(cfg)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
H, W, C = s['height'], s['width'], s['channels']
bg, v, vc, f, g = t(s['background'][None]), t(s['vertices'][None]), t(s['vertex_colors'][None]), t(s['faces'][None]), t(s['grad_pixels'][None])
for it in range(n):
    # the step bench.py times: the forward leaves its state, the backward consumes it
    px, state = ops._op_rasterise(bg, v, vc, f, H, W, C, flags=flags, keep_state=True, dense_grads=True)
    out = ops._op_rasterise_grad(v, f, px, g, H, W, C, flags=flags, state=state, state_outputs='dense')
torch.cuda.synchronize()
