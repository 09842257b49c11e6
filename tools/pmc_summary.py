"""Summarise rocprofv3 --pmc / --kernel-trace CSV output per kernel (averages per dispatch).
usage: python tools/pmc_summary.py <dir-with-*_counter_collection.csv or *_kernel_stats.csv> [...]"""
import collections
import csv
import glob
import os
import sys


def main():
    for d in sys.argv[1:]:
        for path in sorted(glob.glob(os.path.join(d, '**', '*_counter_collection.csv'), recursive=True)):
            agg = collections.defaultdict(lambda: collections.defaultdict(list))
            meta = {}
            for r in csv.DictReader(open(path)):
                k = r['Kernel_Name'].split('(')[0][-48:]
                agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
                meta[k] = (r['VGPR_Count'], r['SGPR_Count'], r['LDS_Block_Size'], r['Workgroup_Size'], r['Grid_Size'])
            print('## %s' % path)
            for k, dct in agg.items():
                print('%s  vgpr=%s sgpr=%s lds=%s wg=%s grid=%s' % ((k,) + meta[k]))
                for c, v in sorted(dct.items()):
                    print('    %-28s %16.0f  (n=%d)' % (c, sum(v) / len(v), len(v)))
        for path in sorted(glob.glob(os.path.join(d, '**', '*_kernel_stats.csv'), recursive=True)):
            print('## %s' % path)
            for r in csv.DictReader(open(path)):
                print('    %-52s calls=%-5s avg=%9.1f us  min=%9.1f  max=%9.1f  %5s%%' % (
                    r['Name'].split('(')[0][-52:], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3,
                    float(r['MaxNs']) / 1e3, r['Percentage']))


if __name__ == '__main__':
    main()
