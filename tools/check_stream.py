"""GPU box: parity of the streaming gradient kernel (DIRT_FLAG_GRAD_STREAM) against the CPU oracle on frames whose sides are
multiples of 32 -- random sizes, meshes (split / shared / hostile / tiny), both quirk-Q1 modes, batches, with and without the
forward's state -- at the tight tolerance; then, optionally, the K3 step with and without it.
usage: python tools/check_stream.py [seconds] [seed]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import oracle  # noqa: E402
from dirt_amd import _lib, rasterise_ops as ops
from tests import scenes  # noqa: E402
from tests import parity  # noqa: E402

STREAM = _lib.FLAG_GRAD_STREAM


def run(budget, seed, sizes=(32, 64, 96, 128, 160, 192, 256, 320)):
    rng = np.random.default_rng(seed)
    dev = torch.device('cuda', 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    t0, n, fails = time.time(), 0, []
    while time.time() - t0 < budget:
        H, W, C = int(rng.choice(sizes)), int(rng.choice(sizes)), 4
        kind = rng.choice(['split', 'shared', 'hostile', 'tiny'])
        seed_ = int(rng.integers(0, 1 << 30))
        if kind == 'hostile':
            s = scenes.hostile_scene(H, W, C, seed_, int(rng.integers(10, 1500)))
        elif kind == 'tiny':
            s = scenes.rand_scene(int(rng.integers(1, 4000)), H, W, C, seed_, 0.001, 0.02)
        else:
            s = scenes.rand_scene(int(rng.integers(1, 3000)), H, W, C, seed_, float(rng.uniform(0.005, 0.1)), float(rng.uniform(0.1, 0.8)), kind == 'shared')
        q1 = int(rng.choice([0, 1]))
        b = {k: v[None] for k, v in s.items() if isinstance(v, np.ndarray)}
        if kind in ('split', 'shared') and rng.random() < 0.4:
            B = int(rng.integers(2, 4))
            F = b['faces'].shape[1]
            bs = scenes.batch_scene(F, H, W, C, [seed_ + i for i in range(B)], r_lo=0.01, r_hi=0.3, shared=(kind == 'shared'))
            b = {k: bs[k] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
        want = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
        use_state = rng.random() < 0.5
        got = ops._op_rasterise(t(b['background']), t(b['vertices']), t(b['vertex_colors']), t(b['faces']), H, W, C, keep_state=use_state)
        got, state = got if use_state else (got, None)
        tag = (kind, b['vertices'].shape, H, W, seed_, q1, use_state)
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), ('forward', tag)
        ow = oracle.backward(b['vertices'], b['faces'], want, b['grad_pixels'], flags=q1, want_mass=True)
        gb, gv, gvc, _ = ops._op_rasterise_grad(t(b['vertices']), t(b['faces']), t(want), t(b['grad_pixels']), H, W, C, flags=STREAM | q1, state=state)
        if not np.array_equal(gb.cpu().numpy(), ow['grad_background']):
            fails.append('grad_background %s' % (tag,))
        try:
            parity.grads_close(gv, gvc, ow, str(tag), tol=parity.TIGHT_TOL)
        except AssertionError as e:
            fails.append(str(e))
        n += 1
    return n, fails


if __name__ == '__main__':
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.
    n, fails = run(budget, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    for f in fails[:20]:
        print('MISMATCH', f)
    print('check_stream: %d cases, %d mismatches' % (n, len(fails)))
    sys.exit(1 if fails else 0)
