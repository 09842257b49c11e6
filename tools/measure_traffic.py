"""HBM traffic per kernel launch of one configuration's step, measured with rocprofv3 PMC counters.

Two SEPARATE `rocprofv3 --pmc` passes (FETCH_SIZE, then WRITE_SIZE -- they do not fit one pass, and counter
collection is never combined with tracing: /opt/skills/guides/MI355X_MICROARCH.md "rocprofv3 PMC slots") over
`tools/prof_run.py <config> 3`, averaged per dispatch and corrected as the guide's HBM section prescribes for
gfx950: bytes = 2 * FETCH_SIZE + WRITE_SIZE, both reported in KiB (FETCH_SIZE tallies 128-byte requests at
64 bytes; calibrated there for wide coalesced reads, other widths and WRITE_SIZE are uncalibrated).

Used two ways:
  * bench.py calls `measure(config)` after its timed region (N = 1, rocprofv3 on PATH): `roofline.traffic` is then
    measured in the same run, on the same library build;
  * `python tools/measure_traffic.py K3 K5 ...` rewrites profiles/pmc_traffic.json, stamped with the sha256 of the
    kernel sources it was collected on (bench.py refuses the file when the stamp does not match the tree).
"""
import collections
import csv
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'dirt_amd', 'csrc')


def source_stamp():
    """sha256 over the kernel sources and headers, in name order."""
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith(('.hip', '.h')):
            h.update(name.encode())
            h.update(open(os.path.join(CSRC, name), 'rb').read())
    return h.hexdigest()


def kernel_key(name):
    """rocprofv3's demangled kernel name -> the names bench.py uses."""
    if 'grad_kernel' in name:
        return 'grad_kernel'
    if 'setup_kernel' in name:
        return 'setup_kernel'
    if 'raster_kernel<0' in name or 'raster_kernel_v2<0' in name:
        return 'raster_kernel<shade>'
    if 'raster_kernel<1' in name or 'raster_kernel_v2<1' in name:
        return 'raster_kernel<visibility>'
    if 'zero_kernel' in name:
        return 'zero_kernel'
    return None


def _pmc_pass(counter, config, steps, out_dir, timeout):
    env = dict(os.environ, TMPDIR='/tmp')
    cmd = ['rocprofv3', '--pmc', counter, '--output-format', 'csv', '-d', out_dir, '-o', 'pmc', '--',
           sys.executable, os.path.join(ROOT, 'tools', 'prof_run.py'), config, str(steps)]
    subprocess.run(cmd, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
    per_kernel = collections.defaultdict(list)
    for path in glob.glob(os.path.join(out_dir, '**', '*_counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(path)):
            k = kernel_key(r['Kernel_Name'])
            if k is not None and r['Counter_Name'] == counter:
                per_kernel[k].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in per_kernel.items()}


def measure(config='K3', steps=3, timeout=150):
    """-> {kernel: HBM bytes per launch}, or raises (no rocprofv3, no GPU, time-out)."""
    if shutil.which('rocprofv3') is None:
        raise RuntimeError('rocprofv3 is not on PATH')
    tmp = tempfile.mkdtemp(prefix='dirt_pmc_', dir='/tmp')
    try:
        fetch = _pmc_pass('FETCH_SIZE', config, steps, os.path.join(tmp, 'fetch'), timeout)
        write = _pmc_pass('WRITE_SIZE', config, steps, os.path.join(tmp, 'write'), timeout)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    out = {}
    for k in fetch:
        out[k] = int(round((2.0 * fetch[k] + write.get(k, 0.0)) * 1024.0))
    if not out:
        raise RuntimeError('rocprofv3 produced no counter rows')
    return out


def main():
    configs = sys.argv[1:] or ['K3']
    path = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')
    data = json.load(open(path)) if os.path.exists(path) else {}
    for cfg in configs:
        data[cfg] = measure(cfg)
        data[cfg]['_sources_sha256'] = source_stamp()   # per configuration: bench.py refuses an entry collected on other sources
        print(cfg, data[cfg])
    data['_note'] = ('HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the gfx950 FETCH_SIZE correction of '
                     'MI355X_MICROARCH.md (calibrated there for wide coalesced reads; narrower accesses and WRITE_SIZE are uncalibrated)')
    data.pop('_collected', None)
    json.dump(data, open(path, 'w'), indent=1)


if __name__ == '__main__':
    main()
