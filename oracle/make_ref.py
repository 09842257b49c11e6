"""Build oracle/_ref/libdirt_ref.so: the reference's OWN gradient kernel compiled for the host.

TEST INFRASTRUCTURE.  /root/reference/csrc/rasterise_grad_egl.cu (Vec3, assemble_grads,
launch_grad_assembly, upload_vertices, launch_vertex_upload) and csrc/rasterise_egl.cu (upload_background,
download_pixels: the forward op's two data movers, with the vertical flip and the atlas tiling) are compiled
from where they lie, together with their own headers csrc/tf_cuda_utils.h and csrc/rasterise_grad_common.h, against the host shim in
oracle/ref_shim/ (which stands in for <tensorflow/...> and the CUDA vocabulary).  The ONE textual edit
is the launch syntax g++ cannot parse: `kernel<<<grid, block, shm, stream>>>(` becomes
`REF_SHIM_LAUNCH(kernel, grid, block, shm, stream)(`; the edited copy goes to the git-ignored
oracle/_ref/ and nowhere else.  The OpenGL render that feeds the kernel (csrc/rasterise_grad_egl.cpp:
432-456) lives in NVIDIA's driver and cannot be built: oracle/ref.py feeds the kernel the surfaces of
the specification-pinned oracle visibility instead.

Without /root/reference (the GPU box) this is a no-op: the prebuilt .so travels with the snapshot.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_CSRC = '/root/reference/csrc'
OUT = os.path.join(HERE, '_ref')
SO = os.path.join(OUT, 'libdirt_ref.so')
_LAUNCH = re.compile(r'(\b\w+)<<<(.*?)>>>\(', re.S)


def available():
    return all(os.path.exists(os.path.join(REF_CSRC, u)) for u in ('rasterise_grad_egl.cu', 'rasterise_egl.cu'))


UNITS = {'rasterise_grad_egl.cu': 2, 'rasterise_egl.cu': 2}   # translation unit -> kernel launches in it


def build(force=False, verbose=False):
    """-> path of the .so, or None when neither the reference nor a prebuilt library is present."""
    if not available():
        return SO if os.path.exists(SO) else None
    deps = [os.path.join(REF_CSRC, u) for u in UNITS] + [
        os.path.join(REF_CSRC, 'tf_cuda_utils.h'), os.path.join(REF_CSRC, 'rasterise_grad_common.h'),
        os.path.join(HERE, 'ref_driver.cpp'), os.path.join(HERE, 'ref_shim/tensorflow/core/framework/tensor.h'),
        os.path.join(HERE, 'ref_shim/tensorflow/core/util/cuda_launch_config.h'), os.path.abspath(__file__)]
    if not force and os.path.exists(SO) and all(os.path.getmtime(SO) >= os.path.getmtime(d) for d in deps):
        return SO
    os.makedirs(OUT, exist_ok=True)
    host_srcs = []
    for unit, launches in UNITS.items():
        text = open(os.path.join(REF_CSRC, unit)).read()
        text, n = _LAUNCH.subn(lambda m: 'REF_SHIM_LAUNCH(%s, %s)(' % (m.group(1), m.group(2)), text)
        assert n == launches, 'expected %d kernel launches in %s, found %d' % (launches, unit, n)
        host_src = os.path.join(OUT, unit.replace('.cu', '_host.cpp'))
        with open(host_src, 'w') as f:
            f.write(text)
        host_srcs.append(host_src)
    # -ffp-contract=off: nvcc would contract a*b+c into fma where it likes; the reference's results are
    # only defined up to that, and the oracle restates the uncontracted expression order.
    cmd = ['g++', '-O2', '-std=c++14', '-fPIC', '-shared', '-ffp-contract=off', '-fno-fast-math', '-w',
           '-I', os.path.join(HERE, 'ref_shim'), '-I', REF_CSRC] + host_srcs + [os.path.join(HERE, 'ref_driver.cpp'), '-o', SO]
    if verbose:
        print(' '.join(cmd))
    try:
        subprocess.check_call(cmd)
    finally:
        for host_src in host_srcs:
            os.remove(host_src)  # the edited copies of the reference text do not stay in the tree, not even git-ignored
    return SO


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
