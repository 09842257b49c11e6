"""Turns gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the tracked summaries under profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py --steps 50 --warmup 10`
   profiles/<tag>_pmc.txt            per-kernel PMC averages (separate --pmc passes)
   profiles/<tag>_bench.json         the bench line of the same build
   profiles/pmc_traffic.json         HBM bytes per launch per kernel (tools/measure_traffic.py on the box: FETCH_SIZE doubled as
                                     /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950; stamped with the sources' sha256)
usage: python tools/summarise_profiles.py r01 [config]"""
import collections
import csv
import glob
import io
import json
import os
import shutil
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import pmc_summary  # noqa: E402

def slot(kernel):
    """rocprofv3 kernel name -> the library's profiling slot (dirt_profile_name), whatever the tile-shape template."""
    if kernel.startswith('grad_kernel'):
        return 'grad_kernel'
    if kernel.startswith('raster_kernel<0') or kernel.startswith('raster_kernel_v2<0'):
        return 'raster_kernel<shade>'
    if kernel.startswith('setup_kernel'):
        return 'setup_kernel'
    if kernel.startswith('raster_kernel<1') or kernel.startswith('raster_kernel_v2<1'):
        return 'raster_kernel<visibility>'
    return kernel


def counter(path, name):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == name:
            k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('dirt::', '')
            agg[k].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    tag = sys.argv[1]
    config = sys.argv[2] if len(sys.argv) > 2 else 'K3'
    src = os.path.join(ROOT, 'gpurun_out', 'prof_' + tag)
    dst = os.path.join(ROOT, 'profiles')
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, 'trace', 'trace_kernel_stats.csv'), os.path.join(dst, tag + '_kernel_stats.csv'))
    buf = io.StringIO()
    with redirect_stdout(buf):
        sys.argv = ['pmc_summary'] + [os.path.join(src, d) for d in ('trace', 'pmc_sq', 'pmc_act', 'pmc_lds', 'pmc_fetch', 'pmc_write', 'pmc_l2') if os.path.isdir(os.path.join(src, d))]
        pmc_summary.main()
    open(os.path.join(dst, tag + '_pmc.txt'), 'w').write(buf.getvalue().replace(ROOT + '/', ''))
    line = [l for l in open(os.path.join(src, 'bench.json')).read().splitlines() if l.startswith('{')][-1]
    json.dump(json.loads(line), open(os.path.join(dst, tag + '_bench.json'), 'w'), indent=1)
    for extra in sorted(glob.glob(os.path.join(src, 'bench_*.json'))):
        name = os.path.basename(extra)[len('bench_'):-len('.json')]
        lines = [l for l in open(extra).read().splitlines() if l.startswith('{')]
        if name != 'under_rocprof' and lines:
            json.dump(json.loads(lines[-1]), open(os.path.join(dst, '%s_bench_%s.json' % (tag, name)), 'w'), indent=1)

    for cfg in ('K5', 'K3-256'):
        stats = os.path.join(src, 'trace_' + cfg, 'trace_kernel_stats.csv')
        if os.path.exists(stats):
            shutil.copy(stats, os.path.join(dst, '%s_kernel_stats_%s.csv' % (tag, cfg)))
    # HBM traffic per launch: written on the box by tools/measure_traffic.py, stamped with the kernel sources' sha256
    # (bench.py uses an entry only when its stamp matches the tree; its default run measures the traffic itself)
    t = os.path.join(src, 'pmc_traffic.json')
    if os.path.exists(t):
        shutil.copy(t, os.path.join(dst, 'pmc_traffic.json'))
    for name in ('soak.log', 'traffic.log', 'traces.log', 'fuzz.log', 'step_ramp.log', 'texture.log'):
        if os.path.exists(os.path.join(src, name)):
            text = open(os.path.join(src, name)).read().replace('/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n', '')
            open(os.path.join(dst, '%s_%s' % (tag, name.replace('.log', '.txt'))), 'w').write(text)
    if os.path.exists(os.path.join(src, 'texture.json')):
        shutil.copy(os.path.join(src, 'texture.json'), os.path.join(dst, tag + '_texture.json'))
    stats = os.path.join(src, 'trace_deferred_K5', 'trace_kernel_stats.csv')
    if os.path.exists(stats):
        shutil.copy(stats, os.path.join(dst, tag + '_kernel_stats_K5_deferred.csv'))
    drv = []
    for i in (1, 2, 3):
        f = os.path.join(src, 'bench_driver_cmd_%d.json' % i)
        if os.path.exists(f) and open(f).read().strip():
            drv.append(json.loads(open(f).read().strip().splitlines()[-1]))
    if drv:
        json.dump({'command': 'python bench.py --gpus 1 --steps 20 --warmup 5 (three runs on one box; --no-cpu-baseline --traffic off)',
                   'ms_per_step': [d['ms_per_step'] for d in drv], 'ms_per_step_events_median': [d['ms_per_step_events_median'] for d in drv],
                   'value': [d['value'] for d in drv]}, open(os.path.join(dst, tag + '_bench_driver_command.json'), 'w'), indent=1)
    print(open(os.path.join(dst, tag + '_kernel_stats.csv')).read())


if __name__ == '__main__':
    main()
