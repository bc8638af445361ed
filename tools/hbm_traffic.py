"""HBM traffic per kernel from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE).

    python tools/hbm_traffic.py gpurun_out/<tag> profiles/<tag>_hbm_traffic.txt <env_substeps_of_the_rollout_launch> "<command>"

Counter units and the gfx950 correction follow /opt/skills/guides/MI355X_MICROARCH.md
(HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE count KB per dispatch; on gfx950
FETCH_SIZE sees 1/2 of the bytes of a WIDE (16 B/lane) coalesced stream, so the raw and
the doubled read figures bracket the truth for this dword-granular kernel.
Also rewrites profiles/traffic.json (bytes per env-substep of k_env<MODE_ROLLOUT>, upper figure).
"""
import csv
import glob
import json
import os
import sys

tag_dir, out, env_substeps = sys.argv[1], sys.argv[2], float(sys.argv[3])
cmd = sys.argv[4] if len(sys.argv) > 4 else ''


def collect(sub, counter):
    tot = {}
    for f in glob.glob(os.path.join(tag_dir, sub, '**', '*counter_collection.csv'), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r['Counter_Name'] != counter:
                    continue
                tot[r['Kernel_Name']] = tot.get(r['Kernel_Name'], 0.0) + float(r['Counter_Value'])
    return tot


fetch, write = collect('fetch', 'FETCH_SIZE'), collect('write', 'WRITE_SIZE')
lines = ['# rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) -- ' + cmd,
         '# units: KB per dispatch, summed over the dispatches of each kernel; HBM_GB = raw .. with reads doubled (gfx950 correction bracket)',
         '%-44s %16s %16s %20s' % ('kernel', 'FETCH_SIZE_KB', 'WRITE_SIZE_KB', 'HBM_GB(raw..2xF)')]
roll = None
for k in sorted(set(fetch) | set(write)):
    if 'k_env' not in k:
        continue
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    lo, hi = (f + w) * 1024 / 1e9, (2 * f + w) * 1024 / 1e9
    lines.append('%-44s %16.0f %16.0f %9.2f .. %6.2f' % (k[:44], f, w, lo, hi))
    if '<4>' in k:
        roll = hi * 1e9
if roll is not None:
    per = roll / env_substeps
    lines.append('# rollout launch: %.3e env-substeps -> %.0f HBM bytes per env-substep measured (upper figure) vs 3056 algorithmic' % (env_substeps, per))
    lines.append('# the write traffic is private-memory (scratch) spill traffic of the out-of-line heavy substep; the env state itself')
    lines.append('# moves 2 x sizeof(DevEnv) per env per launch')
    with open(os.path.join(os.path.dirname(out), 'traffic.json'), 'w') as fh:
        json.dump({'hbm_bytes_per_env_substep': per, 'source': os.path.basename(out)}, fh)
open(out, 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines))
