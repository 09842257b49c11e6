"""GPU: per-component error / mass of the HIP gradients against the oracle on a few scenes (diagnostic)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from dirt_amd import rasterise_ops as ops
from tests import scenes
dev = torch.device('cuda:0')
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)

def probe(name, s, flags=0):
    b = {k: (s[k] if s['background'].ndim == 4 else s[k][None]) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    B, H, W, C = b['background'].shape
    want = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    ow = oracle.backward(b['vertices'], b['faces'], want, b['grad_pixels'], want_mass=True)
    gb, gv, gvc, _ = ops._op_rasterise_grad(t(b['vertices']), t(b['faces']), t(want), t(b['grad_pixels']), H, W, C, flags=flags)
    gv, gvc = gv.cpu().numpy().astype(np.float64), gvc.cpu().numpy().astype(np.float64)
    out = []
    for lab, g, w, m in (('x', gv[..., 0], ow['grad_vertices'][..., 0], ow['mass_vertices'][..., 0]),
                         ('y', gv[..., 1], ow['grad_vertices'][..., 1], ow['mass_vertices'][..., 1]),
                         ('w', gv[..., 3], ow['grad_vertices'][..., 3], ow['mass_vertices'][..., 3]),
                         ('col', gvc, ow['grad_vertex_colors'], ow['mass_vertex_colors'])):
        ok = np.isfinite(w) & np.isfinite(m) & np.isfinite(g)
        r = np.abs(g - w)[ok] / np.maximum(m[ok], 1e-300)
        r = r[m[ok] > 0]
        zero_bad = int(np.sum((m == 0) & (g != 0)))
        out.append('%s: max %.2e p99.9 %.2e n>1e-4 %d n>1e-5 %d zero-mass-nonzero %d nonfinite-mismatch %d' % (
            lab, r.max() if r.size else 0, np.quantile(r, 0.999) if r.size else 0, int((r > 1e-4).sum()), int((r > 1e-5).sum()), zero_bad,
            int(np.sum(np.isfinite(w) != np.isfinite(g)))))
    print(name, '|', ' | '.join(out), flush=True)

probe('K3', scenes.config_scene('K3'))
probe('K3-256', scenes.config_scene('K3-256'))
probe('K3-2048', scenes.config_scene('K3-2048'))
probe('K5', scenes.config_scene('K5'))
for H, W, C, seed, n in ((96, 80, 4, 1, 200), (70, 50, 3, 2, 1200), (33, 65, 1, 3, 400)):
    probe('hostile %d' % seed, scenes.hostile_scene(H, W, C, seed, n))
probe('tiny tris', scenes.rand_scene(3000, 256, 256, 4, 8, 0.005, 0.04))
probe('cyl', scenes.cylinder_scene())
