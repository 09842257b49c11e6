"""Builds dirt_amd/libdirt_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo).

`python -m dirt_amd.build` or `dirt_amd.build.build_library()`.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(_HERE, 'libdirt_hip.so')
SOURCES = ['dirt_capi.hip', 'dirt_raster.hip', 'dirt_forward.hip', 'dirt_grad.hip', 'dirt_grad_small.hip', 'dirt_grad_px2.hip', 'dirt_grad_stream.hip', 'dirt_texture.hip']
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join('..', '..', 'include', 'dirt_hip.h')]

# -ffp-contract=off: the numeric specification is a sequence of IEEE basic operations
# shared with the CPU oracle; only the fma()s written in the source may be fused.
# -munsafe-fp-atomics: float atomicAdd lowers to global_atomic_add_f32 (no CAS loop).
HIPCC_FLAGS = [
    '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
    '-ffp-contract=off', '-fno-fast-math', '-fhip-fp32-correctly-rounded-divide-sqrt',
    '-munsafe-fp-atomics', '-Wall', '-Wno-unused-function',
    # gfx950 loads the first kernel arguments into SGPRs while a wave is launched: kernels that list, before their parameter
    # struct, the scalars their first memory trip is addressed with start that trip without a scalar load of the argument segment
    '-mllvm', '-amdgpu-kernarg-preload-count=16',
]


def hipcc_path():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def _flags_stamp(extra_flags=()):
    """What the objects were compiled with: a flag change must trigger a rebuild as a source change does."""
    import hashlib
    return hashlib.sha256(repr((HIPCC_FLAGS, sorted(PER_SOURCE_FLAGS.items()), list(extra_flags), SOURCES)).encode()).hexdigest()


def needs_build(extra_flags=()):
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    try:
        return open(os.path.join(OBJ_DIR, 'flags.sha256')).read().strip() != _flags_stamp(extra_flags)
    except OSError:
        return True


# per-source flags.  dirt_grad.hip: the SLP vectoriser turns two of the four per-pixel barycentric triples into
# <2 x float> + float stores to a stack slot that is never promoted back to registers (scratch memory + a 3 KB LDS
# "promoted alloca" in the 1- and 3-channel instantiations); its packed arithmetic is written by hand anyway.
PER_SOURCE_FLAGS = {'dirt_grad.hip': ['-fno-slp-vectorize'], 'dirt_grad_px2.hip': ['-fno-slp-vectorize'], 'dirt_grad_stream.hip': ['-fno-slp-vectorize']}
OBJ_DIR = os.path.join(_HERE, 'csrc', '.obj')


def build_library(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every HIP source (one hipcc -c each, in parallel) and link them into one shared library.  Raises on failure.
    `out`: another path for the library (instrumented / experimental builds under tools/_bin: tools/variants.sh)."""
    if out is None and not force and not needs_build(extra_flags):
        return LIB_PATH
    lib_path = out or LIB_PATH
    obj_dir = OBJ_DIR if out is None else os.path.join(OBJ_DIR, os.path.basename(out))
    os.makedirs(obj_dir, exist_ok=True)
    compile_flags = [f for f in HIPCC_FLAGS if f != '-shared']
    procs = []
    for src in SOURCES:
        obj = os.path.join(obj_dir, src + '.o')
        cmd = [hipcc_path()] + compile_flags + PER_SOURCE_FLAGS.get(src, []) + list(extra_flags) + ['-c', os.path.join(CSRC, src), '-o', obj]
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        procs.append((cmd, obj, subprocess.Popen(cmd)))
    # every compiler is waited for before anything is raised (none is left running behind an exception), and the objects of
    # failed compilations are removed so that no later link can pick up a stale or partial one
    objs, failed = [], None
    for cmd, obj, pr in procs:
        if pr.wait() != 0:
            failed = failed or (pr.returncode, cmd)
            try:
                os.remove(obj)
            except OSError:
                pass
        objs.append(obj)
    if failed:
        raise subprocess.CalledProcessError(failed[0], failed[1])
    link = [hipcc_path(), '--offload-arch=gfx950', '-fPIC', '-shared'] + objs + ['-o', lib_path]
    if verbose:
        print(' '.join(link), file=sys.stderr)
    subprocess.check_call(link)
    if out is None:
        with open(os.path.join(obj_dir, 'flags.sha256'), 'w') as fh:
            fh.write(_flags_stamp(extra_flags))
    return lib_path


def kernel_resources(extra_flags=()):
    """{demangled kernel name: {'vgpr', 'sgpr', 'scratch', 'occupancy', 'lds'}} of every kernel of the library, from hipcc's
    -Rpass-analysis=kernel-resource-usage (every source recompiled with the flags build_library gives it, in parallel; no GPU
    needed).  tools/kernel_resources.sh prints it; tests/test_boundary.py asserts that no kernel uses scratch memory."""
    import re
    import tempfile
    procs = []
    tmp = tempfile.mkdtemp(prefix='dirt_res_')
    for src in SOURCES:
        cmd = [hipcc_path()] + [f for f in HIPCC_FLAGS if f != '-shared'] + PER_SOURCE_FLAGS.get(src, []) + list(extra_flags) + \
              ['-c', os.path.join(CSRC, src), '-o', os.path.join(tmp, src + '.o'), '-Rpass-analysis=kernel-resource-usage']
        procs.append(subprocess.Popen(cmd, stderr=subprocess.PIPE, stdout=subprocess.DEVNULL, text=True))
    rows = {}
    for pr in procs:
        err = pr.communicate()[1]
        if pr.returncode != 0:
            raise RuntimeError('hipcc failed:\n' + err[-2000:])
        cur = None
        for line in err.splitlines():
            m = re.search(r'remark:\s+(.*?)\s*\[-Rpass', line)
            if not m:
                continue
            t = m.group(1)
            if t.startswith('Function Name:'):
                cur = t.split(':', 1)[1].strip()
                rows[cur] = {}
            elif cur and ':' in t:
                k, v = t.split(':', 1)
                rows[cur][k.strip()] = v.strip()
    out = {}
    for mangled, r in rows.items():
        name = subprocess.run(['c++filt', mangled], capture_output=True, text=True).stdout.strip().split('(')[0]
        out[name] = {'vgpr': int(r.get('VGPRs', -1)), 'sgpr': int(r.get('TotalSGPRs', -1)), 'scratch': int(r.get('ScratchSize [bytes/lane]', -1)),
                     'occupancy': int(r.get('Occupancy [waves/SIMD]', -1)), 'lds': int(r.get('LDS Size [bytes/block]', -1))}
    return out


if __name__ == '__main__':
    # python -m dirt_amd.build [--force] [--out path.so] [--flags "-DX=1 ..."]
    argv = sys.argv[1:]
    out_path = argv[argv.index('--out') + 1] if '--out' in argv else None
    flags = argv[argv.index('--flags') + 1].split() if '--flags' in argv else []
    print(build_library(force='--force' in argv, verbose='--quiet' not in argv, extra_flags=flags, out=out_path))
