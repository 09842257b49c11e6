"""Builds dirt_amd/libdirt_hip.so for gfx950 with hipcc (in-tree, so the .so travels with the repo).

`python -m dirt_amd.build` or `dirt_amd.build.build_library()`.  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, 'csrc')
LIB_PATH = os.path.join(_HERE, 'libdirt_hip.so')
SOURCES = ['dirt_capi.hip', 'dirt_raster.hip', 'dirt_grad.hip', 'dirt_grad_small.hip', 'dirt_texture.hip']
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join('..', '..', 'include', 'dirt_hip.h')]

# -ffp-contract=off: the numeric specification is a sequence of IEEE basic operations
# shared with the CPU oracle; only the fma()s written in the source may be fused.
# -munsafe-fp-atomics: float atomicAdd lowers to global_atomic_add_f32 (no CAS loop).
HIPCC_FLAGS = [
    '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
    '-ffp-contract=off', '-fno-fast-math', '-fhip-fp32-correctly-rounded-divide-sqrt',
    '-munsafe-fp-atomics', '-Wall', '-Wno-unused-function',
]


def hipcc_path():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return 'hipcc'


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False, extra_flags=()):
    """Compile every HIP source into one shared library.  Raises on failure."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [hipcc_path()] + HIPCC_FLAGS + list(extra_flags) + [os.path.join(CSRC, s) for s in SOURCES] + ['-o', LIB_PATH]
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == '__main__':
    build_library(force='--force' in sys.argv, verbose=True)
    print(LIB_PATH)
