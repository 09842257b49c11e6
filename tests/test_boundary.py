"""The drop-in boundary, checked without a GPU: the C-ABI library loads and exports every symbol
include/dirt_hip.h declares, argument validation returns the reference's error conditions before any
device work, the Python API mirrors dirt/rasterise_ops.py, and the product path never touches the
oracle or a CPU fallback."""
import ctypes
import inspect
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from dirt_amd import build, _lib
    build.build_library()
    return _lib.load()


def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'dirt_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dirt_[a-z_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol(lib):
    from dirt_amd import _lib
    syms = _header_symbols()
    assert set(syms) == set(_lib.SYMBOLS), (syms, _lib.SYMBOLS)
    for s in syms:
        assert hasattr(lib, s), 'libdirt_hip.so does not export %s' % s
    assert lib.dirt_abi_version() == _lib.ABI_VERSION == 4


def test_python_flag_constants_match_the_header():
    """dirt_amd/_lib.py restates the DIRT_FLAG_* bits of include/dirt_hip.h: every flag of the header has its Python
    constant with the same value, and no two flags share a bit."""
    import re
    from dirt_amd import _lib
    header = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'dirt_hip.h')).read()
    flags = {m.group(1): int(m.group(2), 0) for m in re.finditer(r'#define DIRT_FLAG_(\w+)\s+(0x[0-9a-fA-F]+|\d+)u', header)}
    assert len(flags) >= 11
    for name, value in flags.items():
        assert getattr(_lib, 'FLAG_' + name) == value, name
    assert len(set(flags.values())) == len(flags)
    bits = 0
    for v in flags.values():
        assert v & (v - 1) == 0 and not (bits & v), 'flags are single, distinct bits'
        bits |= v


def test_header_cites_the_reference_interfaces():
    text = open(os.path.join(ROOT, 'include', 'dirt_hip.h')).read()
    for cite in ('csrc/rasterise_egl.cpp:32-51', 'csrc/rasterise_egl.cpp:276-407', 'csrc/rasterise_grad_egl.cpp:33-53',
                 'csrc/rasterise_grad_egl.cu:93-278', 'dirt/rasterise_ops.py:132-177', 'csrc/hwc.h:27-28'):
        assert cite in text, cite


def test_workspace_bytes_and_size_validation(lib):
    assert lib.dirt_workspace_bytes(1, 30000, 10000, 1024, 1024, 4) >= 10000 * 136 + 4 * 1024 * 1024
    assert lib.dirt_workspace_bytes(1, 3, 1, 0, 16, 3) == 0          # CHECK(width > 0 && height > 0), csrc/hwc.h:28
    assert lib.dirt_workspace_bytes(1, 3, 1, 16, 16, 0) == 0
    assert lib.dirt_workspace_bytes(-1, 3, 1, 16, 16, 3) == 0
    assert lib.dirt_workspace_bytes(1, 3, 1, 1 << 20, 16, 3) == 0
    assert b'height and width' in lib.dirt_last_error() or b'frame larger' in lib.dirt_last_error()


def test_error_codes_before_any_device_work(lib):
    from dirt_amd import _lib
    one = ctypes.c_void_p(16)  # never dereferenced: validation fails first
    rc = lib.dirt_rasterise_forward(one, one, one, one, one, 1, 3, 1, 0, 8, 3, one, 1 << 20, 0, None)
    assert rc == _lib.E_INVALID_ARGUMENT
    rc = lib.dirt_rasterise_forward(None, one, one, one, one, 1, 3, 1, 8, 8, 3, one, 1 << 20, 0, None)
    assert rc == _lib.E_INVALID_ARGUMENT and b'NULL' in lib.dirt_last_error()
    rc = lib.dirt_rasterise_forward(one, one, one, one, one, 1, 3, 1, 8, 8, 3, None, 0, 0, None)
    assert rc == _lib.E_WORKSPACE
    rc = lib.dirt_rasterise_forward(one, one, one, one, one, 1, 3, 1, 8, 8, 3, one, 16, 0, None)
    assert rc == _lib.E_WORKSPACE and b'too small' in lib.dirt_last_error()
    rc = lib.dirt_rasterise_forward(one, one, one, one, one, 1, 3, 1, 8, 8, 3, ctypes.c_void_p(20), 1 << 20, 0, None)
    assert rc == _lib.E_WORKSPACE and b'aligned' in lib.dirt_last_error()
    # V > 2^24: csrc/rasterise_grad_egl.cpp:399-405
    rc = lib.dirt_rasterise_backward(one, one, one, one, one, one, one, None, 1, (1 << 24) + 1, 1, 8, 8, 3, one, 1 << 40, 0, None)
    assert rc == _lib.E_TOO_MANY_VERTICES and b'maximum of 16777216 vertices' in lib.dirt_last_error()
    with pytest.raises(ValueError):
        _lib.check(rc)
    # B == 0 is a no-op, as an empty batch is for the reference
    assert lib.dirt_rasterise_forward(None, None, None, None, None, 0, 3, 1, 8, 8, 3, None, 0, 0, None) == 0


def test_python_api_mirrors_the_reference_signatures():
    """dirt/rasterise_ops.py:13,51,260,313 and dirt/__init__.py:2."""
    import dirt_amd
    from dirt_amd import rasterise_ops as ops
    def names(fn):
        return list(inspect.signature(fn).parameters)
    assert names(ops.rasterise) == ['background', 'vertices', 'vertex_colors', 'faces', 'height', 'width', 'channels', 'name']
    assert names(ops.rasterise_batch) == names(ops.rasterise)
    want = ['background_attributes', 'vertices', 'vertex_attributes', 'faces', 'shader_fn', 'shader_additional_inputs', 'name']
    assert names(ops.rasterise_deferred)[:7] == want and names(ops.rasterise_batch_deferred)[:7] == want
    for n in ('rasterise', 'rasterise_batch', 'rasterise_deferred', 'rasterise_batch_deferred'):
        assert getattr(dirt_amd, n) is getattr(ops, n)
    assert all(p.default is None for k, p in inspect.signature(ops.rasterise).parameters.items() if k in ('height', 'width', 'channels', 'name'))


def test_shape_errors_match_the_reference_conditions():
    """OP_REQUIRES of csrc/rasterise_egl.cpp:301-316 -> ValueError with the reference's messages."""
    from dirt_amd import rasterise_ops as ops
    bg = torch.zeros(1, 8, 8, 3)
    v, vc, f = torch.zeros(1, 4, 4), torch.zeros(1, 4, 3), torch.zeros(1, 2, 3, dtype=torch.int32)
    with pytest.raises(ValueError, match='vertices to be 3D'):
        ops.rasterise_batch(bg, torch.zeros(1, 4, 3), vc, f)
    with pytest.raises(ValueError, match='vertex_colors to be 3D'):
        ops.rasterise_batch(bg, v, torch.zeros(1, 5, 3), f)
    with pytest.raises(ValueError, match='faces to be 3D'):
        ops.rasterise_batch(bg, v, vc, torch.zeros(1, 2, 4, dtype=torch.int32))
    with pytest.raises(ValueError, match='same leading'):
        ops.rasterise_batch(bg, torch.zeros(2, 4, 4), torch.zeros(2, 4, 3), f)
    with pytest.raises(ValueError, match='background_tensor to be 4D'):
        ops.rasterise_batch(bg, v, vc, f, height=9)
    with pytest.raises(ValueError, match='RasteriseGrad expects grad_pixels'):
        ops._op_rasterise_grad(v, f, bg, torch.zeros(1, 8, 9, 3), 8, 8, 3)


def test_no_cpu_fallback():
    """CPU tensors are refused (the reference registers DEVICE_GPU kernels only, csrc/rasterise_egl.cpp:410)."""
    from dirt_amd import rasterise_ops as ops
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        ops.rasterise(torch.zeros(8, 8, 1), torch.zeros(3, 4), torch.zeros(3, 1), torch.zeros(1, 3, dtype=torch.int32))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dirt_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'libdirt_hip.so'))
    with pytest.raises(_lib.DirtLibraryError, match='no fallback'):
        _lib.load()


def test_product_code_never_touches_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/."""
    pkg = os.path.join(ROOT, 'dirt_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h', '.cpp')):
                text = open(os.path.join(dirpath, fn), errors='replace').read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', text, flags=re.M), fn
                assert not re.search(r'#\s*include[^\n]*oracle', text), fn
                assert 'dirt_oracle_' not in text and 'libdirt_oracle' not in text, fn   # symbols / library of the oracle
    import subprocess
    out = subprocess.run(['bash', '-c', 'ldd %s | grep -c oracle || true' % os.path.join(pkg, 'libdirt_hip.so')],
                         capture_output=True, text=True).stdout.strip()
    assert out == '0'


def test_channel_groups_follow_the_reference():
    """The group boundaries the kernels and the oracle implement are those of dirt/rasterise_ops.py:90-95."""
    def ref_groups(channels):
        out, begin = [], 0
        while begin < channels:
            end = begin + 3 if begin + 3 <= channels else begin + 1
            out.append((begin, end))
            begin = end
        return out
    assert ref_groups(4) == [(0, 3), (3, 4)]
    assert ref_groups(16) == [(0, 3), (3, 6), (6, 9), (9, 12), (12, 15), (15, 16)]
    assert ref_groups(5) == [(0, 3), (3, 4), (4, 5)]
    src = open(os.path.join(ROOT, 'dirt_amd', 'csrc', 'dirt_grad.hip')).read()
    # launch_grad: groups of 3 while >= 3 channels remain, then singles (the last 3-group and the first single share a pass)
    assert "const int groups3 = p.C / 3, singles = p.C % 3;" in src


def test_scene_generators_are_deterministic():
    from tests import scenes
    a, b = scenes.config_scene('K3'), scenes.config_scene('K3')
    assert all(np.array_equal(a[k], b[k]) for k in ('vertices', 'faces', 'vertex_colors', 'background', 'grad_pixels'))
    assert a['vertices'].shape == (30000, 4) and a['faces'].shape == (10000, 3) and a['background'].shape == (1024, 1024, 4)
    assert np.all(a['vertices'][:, 3] > 0)
    v, f = scenes.rand_mesh(1000, 3, 0, 0, shared=True)
    assert f.max() < len(v) and len(v) < len(f)


def test_state_gradient_accumulator_layout(lib):
    """dirt_state_grad_buffers (ABI 2): the two accumulators share rows of 4 + C (rounded up to 4) floats -- all of a
    vertex's values in one row: what float atomics are priced by.  Host arithmetic only: no device work."""
    import ctypes
    B, V, F, H, W = 2, 50, 30, 40, 24
    for C, want in ((1, 8), (3, 8), (4, 8), (5, 12), (16, 20)):
        n = lib.dirt_workspace_bytes(B, V, F, H, W, C)
        assert n > 0
        buf = np.zeros(n + 256, dtype=np.uint8)
        base = (buf.ctypes.data + 15) // 16 * 16
        gv, gvc, s1, s2 = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int()
        assert lib.dirt_state_grad_buffers(base, n, B, V, F, H, W, C, ctypes.byref(gv), ctypes.byref(gvc), ctypes.byref(s1), ctypes.byref(s2)) == 0
        assert (s1.value, s2.value) == (want, want)
        assert gv.value % 16 == 0 and gvc.value == gv.value + 16    # colours 16 bytes behind the positions of the same vertex
        assert gv.value + B * V * want * 4 <= base + n


def test_no_kernel_uses_scratch_memory_and_the_big_kernels_keep_their_occupancy():
    """Scratch (register spills) is ruinous on this path -- round 5 measured 27 -> 35 us and 414 -> 564 us for kernels with
    32-96 bytes of it (profiles/EXPERIMENTS.md) -- and a register more than the bound costs a wave per SIMD: every kernel
    of the library compiles to 0 bytes of scratch, the headline's kernels to four workgroups per compute unit (<= 128 VGPRs,
    <= 40 KB of LDS), the two-pixels-per-lane gradient kernel to five (4 channels) and eight (1, 3 channels)."""
    from dirt_amd import build
    res = build.kernel_resources()
    assert len(res) >= 40, sorted(res)
    spilling = {k: v['scratch'] for k, v in res.items() if v['scratch'] != 0}
    assert not spilling, spilling
    for name, v in res.items():
        if 'raster_kernel' in name or ('grad_kernel<' in name and 'grad_kernel<6' not in name):
            assert v['vgpr'] <= 128 and v['lds'] <= 40960, (name, v)
    assert res['void dirt::grad_kernel_px2<4, false>']['vgpr'] <= 96 and res['void dirt::grad_kernel_px2<4, false>']['lds'] <= 27306
    for c in (1, 3):
        v = res['void dirt::grad_kernel_px2<%d, false>' % c]
        assert v['vgpr'] <= 64 and v['lds'] <= 20480, (c, v)


def test_graphed_step_refuses_what_it_cannot_bind():
    """dirt_amd.GraphedStep binds float32, contiguous GPU tensors in place (they become the captured graph's inputs) and needs
    exactly one of loss_fn / grad_pixels: anything else is refused before any device work, as the ops refuse CPU tensors."""
    import dirt_amd
    bg, v, vc = torch.zeros(1, 8, 8, 3), torch.zeros(1, 4, 4), torch.zeros(1, 4, 3)
    f = torch.zeros(1, 2, 3, dtype=torch.int32)
    with pytest.raises(ValueError, match='GPU tensors'):
        dirt_amd.GraphedStep(bg, v, vc, f, loss_fn=lambda p: p.sum())
    assert callable(dirt_amd.backward)
    with pytest.raises(RuntimeError, match='no CPU fallback|MI355X'):
        dirt_amd.rasterise_batch(bg, v, vc, f)
