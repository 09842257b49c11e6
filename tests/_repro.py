import os, sys; sys.path.insert(0, '.')
import numpy as np, torch
import oracle
from dirt_amd import scenes, rasterise_ops as ops
dev = torch.device('cuda:0')
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
H, W, C, seed = 376, 16, 5, 875185112
# the generator call order of fuzz_parity for 'hostile': n_small drawn after seed
rng = np.random.default_rng(99)
found = None
while found is None:
    h, w = int(rng.integers(1, 400)), int(rng.integers(1, 400))
    c = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 10]))
    kind = rng.choice(['split', 'shared', 'hostile', 'tiny'])
    sd = int(rng.integers(0, 1 << 30))
    if kind == 'hostile':
        ns = int(rng.integers(10, 1500)); s = (h, w, c, sd, ns)
    elif kind == 'tiny':
        rng.integers(1, 4000); s = None
    else:
        rng.integers(1, 3000); rng.uniform(0.005, 0.1); rng.uniform(0.1, 0.8); s = None
    fl = int(rng.choice([0, 0x200, 0x400])) | int(rng.choice([0, 1]))
    tp = int(rng.choice([1, 2, 3, 4, 5])); sl = int(rng.choice([32, 64]))
    if kind in ('split', 'shared') and rng.random() < 0.4:
        rng.integers(2, 4)
    us = rng.random() < 0.5
    if sd == seed: found = (s, fl, tp, sl, us)
print('case', found)
(h, w, c, sd, ns), fl, tp, sl, us = found
s = scenes.hostile_scene(h, w, c, sd, ns)
b = {k: v[None] for k, v in s.items() if isinstance(v, np.ndarray)}
want = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
ow = oracle.backward(b['vertices'], b['faces'], want, b['grad_pixels'], flags=fl & 1)
for slots in ('32', '64'):
    for tiles in (0, 0x200, 0x400):
        os.environ['DIRT_GRAD_SLOTS'] = slots
        gb, gv, gvc, _ = ops._op_rasterise_grad(t(b['vertices']), t(b['faces']), t(want), t(b['grad_pixels']), h, w, c, flags=(fl & 1) | tiles)
        g = gv.cpu().numpy(); o = ow['grad_vertices']
        err = np.abs(g - o); i = np.unravel_index(err.argmax(), err.shape)
        print('slots', slots, 'tiles', hex(tiles), 'max err %.4g at %s  gpu %.6g oracle %.6g  scale %.4g  rel %.3g' % (err.max(), i, g[i], o[i], np.abs(o).max(), err.max() / np.abs(o).max()))
v = b['vertices'][0]
print('vertex', i[1], v[i[1]], 'min |w| %.3g' % np.abs(v[:, 3]).min())
