"""Deferred shading at K5 (2048x2048, 16-channel G-buffer, 50 000 triangles): time of one
rasterise_deferred forward + backward with a 3-channel shader, with the forward's state shared by both gradient
calls (what dirt_amd.rasterise_deferred does) and with every call rendering again (the reference's structure).
usage (GPU box): python tools/bench_deferred.py [config] [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from dirt_amd import rasterise_ops as ops
from tests import scenes  # noqa: E402


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else 'K5'
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    dev = torch.device('cuda', 0)
    F, H, W, C, seed, r_lo, r_hi = scenes.CONFIGS[config]
    s = scenes.rand_scene(F, H, W, C, seed, r_lo, r_hi)
    bg, v, a = (torch.from_numpy(s[k]).to(dev).requires_grad_(True) for k in ('background', 'vertices', 'vertex_colors'))
    f = torch.from_numpy(s['faces']).to(dev)
    d = torch.from_numpy(np.random.default_rng(0).standard_normal((H, W, 3)).astype(np.float32)).to(dev)

    def shader(g):
        return g[..., :3] * g[..., 3:4]

    def shader_split(g):
        # the same arithmetic with ONE split: torch's backward of k slices of the G-buffer is k zero-filled G-buffer-sized
        # gradients added up; the backward of a split is a single concatenation
        rgb, k, _ = g.split([3, 1, C - 4], dim=-1)
        return rgb * k

    def step_shared():
        bg.grad = v.grad = a.grad = None   # (as an optimiser's zero_grad(set_to_none=True): no 268 MB accumulate-into-.grad kernel)
        px = ops.rasterise_deferred(bg, v, a, f, shader)
        px.backward(d)

    def step_shared_split():
        bg.grad = v.grad = a.grad = None
        px = ops.rasterise_deferred(bg, v, a, f, shader_split)
        px.backward(d)

    def step_rerender():  # the same arithmetic with a fresh set-up + visibility render in each of the three calls
        with torch.no_grad():
            g = ops._op_rasterise(bg[None], v[None], a[None], f[None], H, W, C)
        gi = g.detach().requires_grad_(True)
        px = shader(gi)
        ops._op_rasterise_grad(v[None], f[None], px.detach().contiguous(), d[None], H, W, 3)
        dg, = torch.autograd.grad(px, [gi], d[None])
        ops._op_rasterise_grad(v[None], f[None], g, dg.contiguous(), H, W, C)

    def step_shared_raw():  # the raw ops of step_rerender with the forward's state shared: no autograd engine in either leg
        with torch.no_grad():
            g, state = ops._op_rasterise(bg[None], v[None], a[None], f[None], H, W, C, keep_state=True, state_channels=4)
        gi = g.detach().requires_grad_(True)
        px = shader(gi)
        ops._op_rasterise_grad(v[None], f[None], px.detach().contiguous(), d[None], H, W, 3, state=state, state_outputs=False)
        dg, = torch.autograd.grad(px, [gi], d[None])
        ops._op_rasterise_grad(v[None], f[None], g, dg.contiguous(), H, W, C, state=state, state_outputs=False)

    out = {'config': config, 'steps': steps,
           'legs': 'shared_state: dirt_amd.rasterise_deferred(...).backward() (autograd engine: ~0.1 ms of host time per step); '
                   'shared_state_raw / rerender: the same arithmetic as raw op calls, with the forward state shared / with a fresh '
                   'set-up + visibility render in each gradient call; shared_state_split_shader: shared_state with the shader '
                   'written with one split of the G-buffer instead of two slices (torch autograd cost of the shader, not of the path)'}
    for name, fn in (('shared_state', step_shared), ('shared_state_split_shader', step_shared_split), ('shared_state_raw', step_shared_raw),
                     ('rerender', step_rerender)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[name] = {'ms_per_step': dt * 1e3, 'Mpixels_per_s': H * W / dt / 1e6}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
