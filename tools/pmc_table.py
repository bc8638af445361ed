"""Summarise tools/pmc_phases.sh output: per-substep, per-wave counter values of k_env<MODE_SUB>."""
import csv, glob, os, sys
out = sys.argv[1]
stops = sys.argv[2:]
rows = {}
for st in stops:
    for tag in 'ab':
        for f in glob.glob(os.path.join(out, '%s%s' % (tag, st), '**', '*counter_collection.csv'), recursive=True):
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    if 'k_env' not in r.get('Kernel_Name', ''):
                        continue
                    if 'Li2E' not in r['Kernel_Name'] and 'MODE_SUB' not in r['Kernel_Name'] and '<2>' not in r['Kernel_Name']:
                        continue
                    key = (st, r['Counter_Name'])
                    rows.setdefault(key, []).append(float(r['Counter_Value']))
names = sorted({k[1] for k in rows})
print('per wave per substep (last dispatch of each run; 1024 waves x 200 substeps)')
print('%-22s' % 'counter' + ''.join('%12s' % ('stop' + s) for s in stops))
for n in names:
    line = '%-22s' % n
    for st in stops:
        v = rows.get((st, n))
        line += '%12.1f' % (v[-1] / (1024 * 200.0)) if v else '%12s' % '-'
    print(line)
