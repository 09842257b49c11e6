"""Is anything read by the forward that this call did not write?  The forward on a workspace pre-filled with a byte pattern."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dirt_amd import _lib
from tests import scenes
dev = torch.device('cuda:0')
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
lib = _lib.load()
cfg, fill, flags = sys.argv[1], int(sys.argv[2], 0), int(sys.argv[3], 0)
F, H, W, C, seed, rlo, rhi = scenes.CONFIGS[cfg]
b = scenes.batch_scene(F, H, W, C, [seed], r_lo=rlo, r_hi=rhi)
d = {k: t(b[k]) for k in ('background', 'vertices', 'vertex_colors', 'faces')}
V = d['vertices'].shape[1]
n = lib.dirt_workspace_bytes(1, V, F, H, W, C)
ws = torch.full((n,), fill, dtype=torch.uint8, device=dev)
px = torch.empty_like(d['background'])
torch.cuda.synchronize()
rc = lib.dirt_rasterise_forward(d['background'].data_ptr(), d['vertices'].data_ptr(), d['vertex_colors'].data_ptr(), d['faces'].data_ptr(), px.data_ptr(),
                                1, V, F, H, W, C, ws.data_ptr(), n, flags, None)
torch.cuda.synchronize()
print(cfg, 'fill', hex(fill), 'flags', hex(flags), 'rc', rc, 'sum', float(px.sum()), flush=True)
