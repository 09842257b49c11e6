"""The BASELINE.json configurations and the reference's own test scenes, HIP path against the CPU oracle (not against
itself): forward bit for bit, gradients per element within 1e-4 of the L1 mass of their terms (tests/parity.py), with the kernel variants the library selects on
its own (no flags), both through the state the forward pass keeps (what bench.py and autograd time) and statelessly."""
import os
import threading

import numpy as np
import pytest
import torch

from dirt_amd import sharding
from tests import scenes
from dirt_amd import rasterise_ops as ops
from tests import parity

pytestmark = pytest.mark.gpu
GRAD_TOL = 1e-4  # BASELINE.json north_star
TIGHT_TOL = parity.TIGHT_TOL


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _close(got, ow, key, what, index=None):
    """HIP gradient against the oracle's, per element (tests/parity.py)."""
    parity.grad_close(got, ow, key, what, index)


def _close_scale(got, want, what, tol=GRAD_TOL):
    """torch-against-torch comparisons of quantities downstream of the op (shader inputs, dense copies)."""
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else got
    scale = max(1.0, float(np.abs(want).max()))
    err = float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max())
    assert err <= tol * scale, '%s: max abs err %g > %g (scale %g)' % (what, err, tol * scale, scale)


def _check_scene(s, dev, oracle, what):
    """s: batched numpy scene.  Forward + both backward paths against the oracle."""
    H, W, C = s['background'].shape[1:]
    want = oracle.forward(s['background'], s['vertices'], s['vertex_colors'], s['faces'])
    ow = oracle.backward(s['vertices'], s['faces'], want, s['grad_pixels'])
    d = {k: _t(s[k], dev) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    px, state = ops._op_rasterise(d['background'], d['vertices'], d['vertex_colors'], d['faces'], H, W, C, keep_state=True)
    got = px.cpu().numpy()
    nbad = int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
    assert nbad == 0, '%s: %d of %d pixel values differ from the oracle' % (what, nbad, got.size)
    for name, st in (('state-reusing', state), ('stateless', None)):
        gb, gv, gvc, _ = ops._op_rasterise_grad(d['vertices'], d['faces'], px, d['grad_pixels'], H, W, C, state=st)
        assert np.array_equal(gb.cpu().numpy(), ow['grad_background']), '%s %s grad_background' % (what, name)
        _close(gv, ow, 'grad_vertices', '%s %s grad_vertices' % (what, name))
        _close(gvc, ow, 'grad_vertex_colors', '%s %s grad_vertex_colors' % (what, name))
    vis = ops._op_visibility(d['vertices'], d['faces'], H, W).cpu().numpy()
    for i in range(vis.shape[0]):
        assert np.array_equal(vis[i], oracle.visibility(s['vertices'][i], s['faces'][i], H, W)[0]), what + ' visibility'
    return want, ow


@pytest.mark.parametrize('config', ['K3', 'K3-256', 'K3-2048', 'K5'])
def test_baseline_config_matches_oracle(gpu, oracle, config):
    """SURVEY.md 8d / BASELINE.json: the benchmarked configurations themselves, against the oracle."""
    s = scenes.config_scene(config)
    b = {k: s[k][None] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    want, ow = _check_scene(b, gpu, oracle, config)
    # the measured margin, on every configuration: each element within 5e-6 of its terms' mass, 20 x inside the bound
    # (round 4, tools/tol_probe.py -> profiles/r04_tolerance_probe.txt: worst element 4.1e-7 at K3-256, 4.1e-7 at K5)
    d = {k: _t(b[k], gpu) for k in ('vertices', 'faces', 'grad_pixels')}
    _, gv, gvc, _ = ops._op_rasterise_grad(d['vertices'], d['faces'], _t(want, gpu), d['grad_pixels'], *want.shape[1:])
    parity.grads_close(gv, gvc, ow, config + ' tight', tol=TIGHT_TOL)


@pytest.mark.parametrize('config,flags', [('K3', 0x8000), ('K3-768', 0), ('K3-3ch', 0), ('K3-1ch', 0x8000), ('K3-2048', 0x8000)])
def test_two_pixel_gradient_kernel_at_full_size(gpu, oracle, config, flags):
    """grad_kernel_px2 (two pixels per lane, 32 x 16 tiles; round 5) on the benchmarked meshes at full size: pinned with
    DIRT_FLAG_GRAD_PX2 where the library would pick the 4-pixel kernel, unpinned where it is the library's own choice
    (K3-768, K3-3ch) -- dense outputs cleared by the forward launch, as the autograd path uses them: grad_background bit for
    bit, every vertex gradient within 5e-6 of its terms' mass."""
    s = scenes.config_scene(config)
    b = {k: s[k][None] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    H, W, C = b['background'].shape[1:]
    want = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], b['faces'])
    ow = oracle.backward(b['vertices'], b['faces'], want, b['grad_pixels'], want_mass=True)
    d = {k: _t(b[k], gpu) for k in b}
    px, state = ops._op_rasterise(d['background'], d['vertices'], d['vertex_colors'], d['faces'], H, W, C, keep_state=True, dense_grads=True)
    assert np.array_equal(px.cpu().numpy().view(np.uint32), want.view(np.uint32))
    gb, gv, gvc, _ = ops._op_rasterise_grad(d['vertices'], d['faces'], px, d['grad_pixels'], H, W, C, flags=flags, state=state, state_outputs='dense')
    assert gv.is_contiguous() and gvc.is_contiguous()
    assert np.array_equal(gb.cpu().numpy(), ow['grad_background'])
    parity.grads_close(gv, gvc, ow, config + ' px2', tol=TIGHT_TOL)


def test_cube_k2_with_gradients(gpu, oracle):
    """K2: the Gouraud cube of samples/simple.py at 256 x 256 x 3, forward and gradients."""
    s = scenes.cube_scene(256, 256)
    rng = np.random.default_rng(3)
    s['grad_pixels'] = rng.standard_normal(s['background'].shape).astype(np.float32)
    s['background'] = rng.uniform(0, 1, s['background'].shape).astype(np.float32)
    b = {k: s[k][None] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    _check_scene(b, gpu, oracle, 'K2')


def test_k4_slice_batch_of_eight(gpu, oracle):
    """One rank's share of K4: eight K3 scenes in one batch (grid.y = scene); two of them against the oracle."""
    F, H, W, C, seed, r_lo, r_hi = scenes.CONFIGS['K3']
    b = scenes.batch_scene(F, H, W, C, seeds=list(range(8)), r_lo=r_lo, r_hi=r_hi)
    d = {k: _t(b[k], gpu) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    px, state = ops._op_rasterise(d['background'], d['vertices'], d['vertex_colors'], d['faces'], H, W, C, keep_state=True)
    gb, gv, gvc, _ = ops._op_rasterise_grad(d['vertices'], d['faces'], px, d['grad_pixels'], H, W, C, state=state)
    for i in (0, 7):
        one = {k: b[k][i:i + 1] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
        want = oracle.forward(one['background'], one['vertices'], one['vertex_colors'], one['faces'])
        assert np.array_equal(px[i:i + 1].cpu().numpy().view(np.uint32), want.view(np.uint32))
        ow = oracle.backward(one['vertices'], one['faces'], want, one['grad_pixels'])
        assert np.array_equal(gb[i:i + 1].cpu().numpy(), ow['grad_background'])
        _close(gv[i:i + 1], ow, 'grad_vertices', 'scene %d grad_vertices' % i)
        _close(gvc[i:i + 1], ow, 'grad_vertex_colors', 'scene %d grad_vertex_colors' % i)


def test_k4_whole_batch_of_64_on_one_gpu(gpu, oracle):
    """K4 as BASELINE.json states it -- 64 scenes of K3 -- in ONE launch on one GPU (grid.y = scene; 8 GPUs take 8 each):
    three of the 64 against the oracle, the rest through the invariants (uncovered pixels are the background; .z = 0)."""
    F, H, W, C, seed, r_lo, r_hi = scenes.CONFIGS['K3']
    b = scenes.batch_scene(F, H, W, C, seeds=list(range(64)), r_lo=r_lo, r_hi=r_hi)
    d = {k: _t(b[k], gpu) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    px, state = ops._op_rasterise(d['background'], d['vertices'], d['vertex_colors'], d['faces'], H, W, C, keep_state=True)
    gb, gv, gvc, _ = ops._op_rasterise_grad(d['vertices'], d['faces'], px, d['grad_pixels'], H, W, C, state=state)
    for i in (0, 31, 63):
        one = {k: b[k][i:i + 1] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
        want = oracle.forward(one['background'], one['vertices'], one['vertex_colors'], one['faces'])
        assert np.array_equal(px[i:i + 1].cpu().numpy().view(np.uint32), want.view(np.uint32))
        ow = oracle.backward(one['vertices'], one['faces'], want, one['grad_pixels'])
        assert np.array_equal(gb[i:i + 1].cpu().numpy(), ow['grad_background'])
        _close(gv[i:i + 1], ow, 'grad_vertices', 'scene %d grad_vertices' % i)
        _close(gvc[i:i + 1], ow, 'grad_vertex_colors', 'scene %d grad_vertex_colors' % i)
    assert bool((gv[..., 2] == 0).all())
    uncovered = (gb != 0).any(-1)
    assert torch.equal(px[uncovered], d['background'][uncovered])
    assert torch.equal(gb[uncovered], d['grad_pixels'][uncovered])


def test_two_gpus_over_rccl(gpu):
    """tests/multi_gpu_test.py:22-29 of the reference runs the op on two devices; here two ranks, one per GPU, over RCCL:
    `broadcast_shared`, sharded render + gradient, `gather_batch` (tests/nccl_worker.py).  Needs two GPUs."""
    if torch.cuda.device_count() < 2:
        print('test_two_gpus_over_rccl: torch.cuda.device_count() = %d on %s -- the RCCL leg needs two GPUs' % (torch.cuda.device_count(), torch.cuda.get_device_name(0)))
        pytest.skip('torch.cuda.device_count() = %d: one GPU on this box (the gloo world-size-2 tests cover the host logic)' % torch.cuda.device_count())
    import subprocess
    import sys
    import socket
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nccl_worker.py')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), worker], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'nccl_worker ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_one_rank_nccl_group_on_one_gpu(gpu):
    """A process group of ONE rank over RCCL on the one GPU every box has: `init_process_group('nccl', device_id=...)`, the
    dmabuf IPC setting, an all-reduce, a broadcast and a gather on device tensors, the sharded render + gather of
    tests/nccl_worker.py -- everything `bench.py --gpus N` does first, run for real (the two-GPU test above is skipped on
    one-GPU boxes; the host logic at world sizes 2 and 4 is tests/test_distributed.py over gloo)."""
    import subprocess
    import sys
    import socket
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nccl_worker.py')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=1', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), worker], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and 'nccl_worker ok' in out.stdout and 'nccl_worker collectives ok' in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


# ---- the reference's own test scenes ----------------------------------------------------------------------------------

def test_reference_cylinder_scene(gpu, oracle):
    """tests/rasterise_tests.py:50-99,115: the bevelled cylinder at 48 x 36 x 3 under perspective, split vertices."""
    s = scenes.cylinder_scene()
    b = {k: s[k][None] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
    want, _ = _check_scene(b, gpu, oracle, 'cylinder')
    assert 200 < int((want[0] != s['background']).any(-1).sum()) < 400   # the cylinder is there


def test_reference_cylinder_batch_of_two(gpu, oracle):
    """tests/rasterise_tests.py:89,123-132: the same geometry twice, over a black and a blue background."""
    b = scenes.cylinder_batch_scene()
    _check_scene({k: b[k] for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}, gpu, oracle, 'cylinder x2')


def test_reference_cylinder_per_pixel_jacobians(gpu, oracle):
    """The harness of tests/rasterise_tests.py:108-116 -- one backward pass per one-hot d_loss_by_pixels -- on a sample of
    pixels (silhouette, interior, background, frame border), every output against the oracle; the reference only looks
    at these images (cv2.imshow)."""
    s = scenes.cylinder_scene()
    H, W, C = 36, 48, 3
    want = oracle.forward(s['background'][None], s['vertices'][None], s['vertex_colors'][None], s['faces'][None])
    covered = (want[0] != s['background']).any(-1)
    edge = covered & ~(np.roll(covered, 1, 0) & np.roll(covered, -1, 0) & np.roll(covered, 1, 1) & np.roll(covered, -1, 1))
    rng = np.random.default_rng(1)
    picks = [tuple(p) for kind in (np.argwhere(edge), np.argwhere(covered & ~edge), np.argwhere(~covered)) for p in kind[rng.permutation(len(kind))[:8]]]
    picks += [(0, 0), (H - 1, W - 1), (0, W // 2), (H // 2, 0)]
    d = {k: _t(s[k][None], gpu) for k in ('vertices', 'faces')}
    pxd = _t(want, gpu)
    for (y, x) in picks:
        for c in range(C):
            g = np.zeros((1, H, W, C), np.float32)
            g[0, y, x, c] = 1.0
            ow = oracle.backward(s['vertices'][None], s['faces'][None], want, g)
            gb, gv, gvc, _ = ops._op_rasterise_grad(d['vertices'], d['faces'], pxd, _t(g, gpu), H, W, C)
            assert np.array_equal(gb.cpu().numpy(), ow['grad_background'])
            _close(gv, ow, 'grad_vertices', 'pixel (%d,%d,%d) grad_vertices' % (y, x, c))
            _close(gvc, ow, 'grad_vertex_colors', 'pixel (%d,%d,%d) grad_vertex_colors' % (y, x, c))


def test_reference_bent_square_deferred(gpu, oracle):
    """tests/deferred_grad_test.py:121-142 (`get_pixels_deferred_v2`): the bent square's 7-channel G-buffer (mask,
    colours, normals) shaded per pixel through `rasterise_deferred`, with the light intensity and the background as
    shader inputs; every gradient against the manual composition on the oracle (dirt/rasterise_ops.py:204-237)."""
    from dirt_amd import lighting
    clip, faces, world, colours = scenes.bent_square_geometry()
    v = _t(clip, gpu).requires_grad_(True)
    f = _t(faces, gpu)
    normals = lighting.vertex_normals(_t(world[:, :3], gpu), f)
    attrs = torch.cat([torch.ones(6, 1, device=gpu), _t(colours, gpu), normals], dim=1).detach().requires_grad_(True)
    bg_attrs = torch.zeros(32, 32, 7, device=gpu, requires_grad=True)
    light_intensity = torch.tensor(1.0, device=gpu, requires_grad=True)
    background = torch.tensor([0.1, 0.1, 0.1], device=gpu, requires_grad=True)
    light_direction = torch.nn.functional.normalize(torch.tensor([1., -0.3, -0.5], device=gpu), dim=0)

    def calculate_shading(colours_, normals_, li):   # tests/deferred_grad_test.py:58-70
        ambient = colours_ * torch.tensor([0.4, 0.4, 0.4], device=gpu)
        diffuse = lighting.diffuse_directional(normals_.reshape(-1, 3), colours_.reshape(-1, 3), light_direction,
                                               light_color=torch.tensor([0., 1., 0.], device=gpu) * li, double_sided=True)
        return ambient + diffuse.reshape(colours_.shape)

    def shader_fn(gbuffer, li, bgc):
        mask, cols, nrm = gbuffer[..., :1], gbuffer[..., 1:4], gbuffer[..., 4:7]
        return mask * calculate_shading(cols, nrm, li) + (1. - mask) * bgc

    px = ops.rasterise_deferred(bg_attrs, v, attrs, f, shader_fn, [light_intensity, background])
    d = _t(np.random.default_rng(2).standard_normal((32, 32, 3)).astype(np.float32), gpu)
    px.backward(d)

    a_np = attrs.detach().cpu().numpy()
    gbuf = oracle.forward(np.zeros((1, 32, 32, 7), np.float32), clip[None], a_np[None], faces[None])
    gt = _t(gbuf[0], gpu).requires_grad_(True)
    li2 = light_intensity.detach().clone().requires_grad_(True)
    bg2 = background.detach().clone().requires_grad_(True)
    shaded = shader_fn(gt, li2, bg2)
    assert torch.allclose(px, shaded.detach(), atol=1e-6)
    shaded.backward(d)
    want_v = oracle.backward(clip[None], faces[None], shaded.detach().cpu().numpy()[None], d.cpu().numpy()[None])
    want_a = oracle.backward(clip[None], faces[None], gbuf, gt.grad.cpu().numpy()[None])
    _close(v.grad, want_v, 'grad_vertices', 'vertices', 0)
    _close(attrs.grad, want_a, 'grad_vertex_colors', 'attributes', 0)
    assert np.array_equal(bg_attrs.grad.cpu().numpy(), want_a['grad_background'][0]), 'background attributes'
    _close_scale(light_intensity.grad, li2.grad.cpu().numpy(), 'light intensity', tol=1e-5)
    _close_scale(background.grad, bg2.grad.cpu().numpy(), 'background colour', tol=1e-5)


# ---- randomised sweep, sharding, re-entrancy -----------------------------------------------------------------------------

def test_fuzz_parity_slice(gpu, oracle):
    """A fixed-seed slice of tests/fuzz_parity.py: random frame sizes, channel counts, meshes (split / shared / hostile /
    tiny), batches, tile-shape flags, with and without the forward's state."""
    from tests import fuzz_parity
    assert fuzz_parity.run(max_cases=120, seed=2024, max_dim=300) == 120
    assert fuzz_parity.run(max_cases=30, seed=7, hard=True, max_dim=500) == 30


def test_sharded_batch_reassembles_bit_exactly(gpu, oracle):
    """`rasterise_batch_sharded` for ranks 0 and 1 of a world of 2, run one after the other on this GPU: the shards, put
    back in round-robin order, are the unsharded batch bit for bit -- also with one [F,3] topology shared by every scene."""
    b = scenes.batch_scene(300, 64, 96, 4, seeds=[31, 32, 33, 34, 35], r_lo=0.03, r_hi=0.2)
    d = {k: _t(b[k], gpu) for k in ('background', 'vertices', 'vertex_colors', 'faces')}
    full = ops.rasterise_batch(d['background'], d['vertices'], d['vertex_colors'], d['faces'])
    out = torch.empty_like(full)
    for rank in range(2):
        local = sharding.rasterise_batch_sharded(d['background'], d['vertices'], d['vertex_colors'], d['faces'], rank=rank, world_size=2)
        out[sharding.scenes_for_rank(5, rank, 2)] = local
    assert torch.equal(out, full)
    shared_faces = d['faces'][0].contiguous()
    full2 = ops.rasterise_batch(d['background'], d['vertices'], d['vertex_colors'], shared_faces)
    for rank in range(2):
        local = sharding.rasterise_batch_sharded(d['background'], d['vertices'], d['vertex_colors'], shared_faces, rank=rank, world_size=2)
        assert torch.equal(local, full2[sharding.scenes_for_rank(5, rank, 2)])
    want = oracle.forward(b['background'], b['vertices'], b['vertex_colors'], np.tile(b['faces'][:1], [5, 1, 1]))
    assert np.array_equal(full2.cpu().numpy().view(np.uint32), want.view(np.uint32))


def test_two_threads_two_streams(gpu, oracle):
    """The library is re-entrant (include/dirt_hip.h: no device state, thread-local error / profile state): two host
    threads render and differentiate different scenes on their own streams at the same time, repeatedly -- the analogue
    of the reference's tests/multi_gpu_test.py:22-29 (the same op on two devices in one process)."""
    ss = [scenes.rand_scene(400 + 300 * i, 160 + 32 * i, 200, 4 - i, 51 + i, 0.02, 0.2) for i in range(2)]
    wants = []
    for s in ss:
        px = oracle.forward(s['background'][None], s['vertices'][None], s['vertex_colors'][None], s['faces'][None])
        wants.append((px, oracle.backward(s['vertices'][None], s['faces'][None], px, s['grad_pixels'][None])))
    errors = []

    def work(i):
        try:
            s = ss[i]
            stream = torch.cuda.Stream(device=gpu)
            with torch.cuda.stream(stream):
                d = {k: _t(s[k][None], gpu) for k in ('background', 'vertices', 'vertex_colors', 'faces', 'grad_pixels')}
                for _ in range(20):
                    bg, v, vc = (d[k].clone().requires_grad_(True) for k in ('background', 'vertices', 'vertex_colors'))
                    px = ops.rasterise_batch(bg, v, vc, d['faces'])
                    px.backward(d['grad_pixels'])
                    stream.synchronize()
                    assert np.array_equal(px.detach().cpu().numpy().view(np.uint32), wants[i][0].view(np.uint32))
                    assert np.array_equal(bg.grad.cpu().numpy(), wants[i][1]['grad_background'])
                    _close(v.grad, wants[i][1], 'grad_vertices', 'thread %d grad_vertices' % i)
                    _close(vc.grad, wants[i][1], 'grad_vertex_colors', 'thread %d grad_vertex_colors' % i)
        except Exception as e:  # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    assert not errors, errors
