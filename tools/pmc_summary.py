"""Sum rocprofv3 --pmc counters per kernel over one or more output directories.

    python tools/pmc_summary.py <dir> [<dir> ...]

Prints, per directory, kernel x counter sums (k_env<*> and k_point_cloud only) and, when
the counters are there, the derived lane utilisation SQ_THREAD_CYCLES_VALU / (64 x
SQ_ACTIVE_INST_VALU x 4) -- see /opt/skills/guides/MI355X_MICROARCH.md for the units.
"""
import csv
import glob
import os
import sys


def collect(d):
    tot = {}
    for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r['Kernel_Name']
                if 'k_env' not in k and 'k_point_cloud' not in k:
                    continue
                tot.setdefault(k[:40], {}).setdefault(r['Counter_Name'], 0.0)
                tot[k[:40]][r['Counter_Name']] += float(r['Counter_Value'])
    return tot


for d in sys.argv[1:]:
    print('#', d)
    tot = collect(d)
    for k in sorted(tot):
        c = tot[k]
        print('%-42s' % k + '  '.join('%s=%.4g' % (n, c[n]) for n in sorted(c)))
        if 'SQ_THREAD_CYCLES_VALU' in c and 'SQ_ACTIVE_INST_VALU' in c and c['SQ_ACTIVE_INST_VALU'] > 0:
            print('%-42s' % '' + 'thread-cycles per active VALU cycle = %.2f (64 = every lane busy)'
                  % (c['SQ_THREAD_CYCLES_VALU'] / c['SQ_ACTIVE_INST_VALU']))
        if 'SQ_THREAD_CYCLES_VALU' in c and 'SQ_INSTS_VALU' in c and c['SQ_INSTS_VALU'] > 0:
            print('%-42s' % '' + 'thread-cycles per VALU instruction = %.2f' % (c['SQ_THREAD_CYCLES_VALU'] / c['SQ_INSTS_VALU']))
