"""Turns gpurun_out/prof_<tag>/ (tools/collect_profiles.sh) into the tracked summaries under profiles/:
   profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of `python bench.py --steps 50 --warmup 10`
   profiles/<tag>_pmc.txt            per-kernel PMC averages (separate --pmc passes)
   profiles/<tag>_bench.json         the bench line of the same build
   profiles/pmc_traffic.json         HBM bytes per launch per kernel, FETCH_SIZE doubled as
                                     /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950
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
    if kernel.startswith('raster_kernel<0'):
        return 'raster_kernel<shade>'
    if kernel.startswith('setup_kernel'):
        return 'setup_kernel'
    if kernel.startswith('raster_kernel<1'):
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

    def traffic_of(suffix):
        fetch = counter(glob.glob(os.path.join(src, 'pmc_fetch' + suffix, '*counter_collection.csv'))[0], 'FETCH_SIZE')
        write = counter(glob.glob(os.path.join(src, 'pmc_write' + suffix, '*counter_collection.csv'))[0], 'WRITE_SIZE')
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes
        out = collections.defaultdict(int)   # launches of one slot (the per-shape gradient launches of many-channel images) add up
        for k in fetch:
            out[slot(k)] += int(2 * fetch[k] * 1024 + write.get(k, 0) * 1024)
        return dict(out)

    traffic = traffic_of('')
    path = os.path.join(dst, 'pmc_traffic.json')
    allt = json.load(open(path)) if os.path.exists(path) else {}
    allt[config] = traffic
    allt['_collected'] = tag
    if os.path.isdir(os.path.join(src, 'pmc_fetch_K5')):
        allt['K5'] = traffic_of('_K5')
        stats = os.path.join(src, 'trace_K5', 'trace_kernel_stats.csv')
        if os.path.exists(stats):
            shutil.copy(stats, os.path.join(dst, tag + '_kernel_stats_K5.csv'))
    allt['_note'] = ('HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes); the factor 2 is the gfx950 FETCH_SIZE '
                     'correction of MI355X_MICROARCH.md (calibrated there for wide coalesced reads; narrower accesses are uncalibrated)')
    json.dump(allt, open(path, 'w'), indent=1)
    # the bench lines were written before this summary existed: give them the traffic figures of this collection
    for f in glob.glob(os.path.join(dst, tag + '_bench*.json')):
        d = json.load(open(f))
        cfg = d['config']['workload'].split(':')[0]
        spg = d['config'].get('scenes_per_gpu', 1)
        t = allt.get(cfg, {}).get(d['roofline']['kernel'])
        d['roofline']['traffic'] = t * spg if t is not None else None
        d['roofline']['traffic_source'] = ('profiles/pmc_traffic.json (%s): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of one scene of this workload%s, '
                                           'not measured in the bench run' % (tag, ' x %d scenes per launch' % spg if spg > 1 else '')) if t is not None else None
        json.dump(d, open(f, 'w'), indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == '__main__':
    main()
