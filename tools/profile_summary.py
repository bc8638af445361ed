"""Summaries of tools/profile_bench.sh output, written under profiles/ (and profiles/traffic.json,
which bench.py reads for the offline-measured `traffic` and `issue_side` fields).

    python tools/profile_summary.py gpurun_out/<tag> <tag>

HBM bytes: FETCH_SIZE and WRITE_SIZE count KB per dispatch (MI355X_MICROARCH.md, rocprofv3 section);
on gfx950 FETCH_SIZE sees half of the bytes of a wide coalesced stream, so raw and doubled reads
bracket the truth -- the upper figure is reported.  SQ counters are summed over the chip by rocprofv3.
"""
import csv
import glob
import json
import os
import sqlite3
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(ROOT, 'profiles')


def collect(sub):
    """Counters of THE TIMED LAUNCH of this pass: the longest dispatch of an env kernel (the timed rollout is the
    register-rich k_env<4> for a plain launch and the run-time-dispatched k_env<-1> when it goes through the task queue,
    which also runs the reset and the short warm-up: summing per kernel name would mix them), under the key 'k_env'."""
    rows = {}
    for f in glob.glob(os.path.join(src, sub, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if 'k_env' not in r['Kernel_Name']:
                    continue
                d = rows.setdefault((f, r['Dispatch_Id']), {'name': r['Kernel_Name'], 'dur': int(r['End_Timestamp']) - int(r['Start_Timestamp']), 'c': {}})
                d['c'][r['Counter_Name']] = d['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    if not rows:
        return {}
    best = max(rows.values(), key=lambda d: d['dur'])
    TIMED_NAME[0] = best['name']
    return {('k_env', n): v for n, v in best['c'].items()}


TIMED_NAME = [None]


def kernel_stats():
    """Rows of the rocpd `top_kernels` view written by --stats (durations in us)."""
    rows = []
    for db in glob.glob(os.path.join(src, 'trace', '**', '*.db'), recursive=True):
        cur = sqlite3.connect(db).cursor()
        for name, calls, total, avg, pct in cur.execute('select name,total_calls,total_duration,average,percentage from top_kernels'):
            rows.append({'Name': name, 'Calls': calls, 'TotalDurationNs': total, 'AverageNs': avg, 'Percentage': pct})
    return rows


bench = json.loads(open(os.path.join(src, 'bench.json')).read().strip().splitlines()[-1])
ks = kernel_stats()
cmdfile = os.path.join(src, 'command.txt')
command = open(cmdfile).read().strip().replace(ROOT + '/', '') if os.path.exists(cmdfile) else 'python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs'
command = command.split('/')[-1] if command.startswith('python /') else command
headline = '--workload' not in command and '--over' not in command
rl = bench['roofline']
algo_bytes = rl.get('hbm_nominal', rl).get('algorithmic_bytes_per_env_substep', rl.get('algorithmic_bytes_per_env_substep'))
lines = ['# rocprofv3 --kernel-trace --stats -- python ' + command.replace('python ', '', 1) + '      [' + bench['config']['workload'] + ']',
         '%-60s %8s %14s %14s %8s' % ('kernel', 'calls', 'total_us', 'avg_us', 'pct')]
for r in ks[:12]:
    lines.append('%-60s %8d %14.0f %14.0f %8.3f' % (r['Name'][:60], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage']))
lines.append('# bench.py in the same session (HIP events around the k_env<MODE_ROLLOUT> launch): avg_kernel_ms %.1f, value %.0f env-steps/s'
             % (bench['roofline']['avg_kernel_ms'], bench['value']))
open(os.path.join(prof, tag + '_kernel_stats.txt'), 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))

c = {}
for sub in ('fetch', 'write', 'sqa', 'sqb'):
    c.update(collect(sub))
kname = TIMED_NAME[0]
get = lambda n: c.get(('k_env', n), 0.0)
env_substeps = bench['sim_steps_per_s'] * bench['ms_per_step'] * 1e-3 * bench['steps']
f_kb, w_kb = get('FETCH_SIZE'), get('WRITE_SIZE')
hbm = (2 * f_kb + w_kb) * 1024
out = ['# PMC passes (separate rocprofv3 runs), the longest env-kernel dispatch of each = the timed launch: %s' % kname,
       '# launch: %.4g env-substeps (bench.py of the same session)' % env_substeps,
       'FETCH_SIZE_KB %.0f  WRITE_SIZE_KB %.0f  -> HBM bytes (reads doubled, gfx950 correction) %.3e = %.1f B per env-substep (algorithmic: %d)'
       % (f_kb, w_kb, hbm, hbm / env_substeps, algo_bytes)]
names = ['SQ_WAVES', 'SQ_WAVE_CYCLES', 'SQ_BUSY_CYCLES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_FLAT',
         'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_INSTS_BRANCH', 'SQ_IFETCH', 'SQ_INSTS_SMEM']
for n in names:
    out.append('%-22s %16.0f   per env-substep %10.2f' % (n, get(n), get(n) / env_substeps))
wc = get('SQ_WAVE_CYCLES')
insts = get('SQ_INSTS_VALU') + get('SQ_INSTS_SALU') + get('SQ_INSTS_LDS') + get('SQ_INSTS_VMEM') + get('SQ_INSTS_BRANCH')
issue = None
if wc > 0:
    # SQ_WAVE_CYCLES counts quad-cycles (4 clocks) per resident wave (MI355X_MICROARCH.md cycle table)
    valu_per_cycle = get('SQ_INSTS_VALU') / (4.0 * wc)
    issue = {'valu_insts_per_env_substep': get('SQ_INSTS_VALU') / env_substeps,
             'all_insts_per_env_substep': insts / env_substeps,
             'valu_insts_per_wave_cycle': valu_per_cycle,
             'valu_issue_peak_per_simd_cycle': 0.5,
             'valu_issue_frac_of_peak_while_resident': valu_per_cycle / 0.5,
             'wait_frac': get('SQ_WAIT_ANY') / wc, 'issue_stall_frac': get('SQ_WAIT_INST_ANY') / wc,
             'active_frac': get('SQ_ACTIVE_INST_ANY') / wc,
             'source': tag + '_pmc.txt'}
    out.append('# derived: %.3f VALU instructions per wave clock (peak 0.5 per SIMD clock: a wave64 VALU op issues over 2 clocks) = %.1f %% of the VALU issue peak of an occupied SIMD;'
               % (valu_per_cycle, 100 * valu_per_cycle / 0.5))
    out.append('#          wave parked (s_waitcnt / barrier) %.1f %%, issue-stalled %.1f %%, issuing %.1f %% of its resident cycles'
               % (100 * get('SQ_WAIT_ANY') / wc, 100 * get('SQ_WAIT_INST_ANY') / wc, 100 * get('SQ_ACTIVE_INST_ANY') / wc))
open(os.path.join(prof, tag + '_pmc.txt'), 'w').write('\n'.join(out) + '\n')
print('\n'.join(out))
if headline:
  with open(os.path.join(prof, 'traffic.json'), 'w') as fh:
    json.dump({'hbm_bytes_per_env_substep': hbm / env_substeps, 'source': tag + '_pmc.txt', 'issue': issue}, fh, indent=1)
