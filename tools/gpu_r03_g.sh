#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03_g; mkdir -p $O; cd $R
timeout 600 python -m pytest tests/test_limb_dynamics.py tests/test_kat_contact.py -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err
tail -6 $O/tests.log; tail -3 $O/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_g/bench.json').read().strip().splitlines()[-1])
print(d['value'], json.dumps(d['extra'].get('limb_dynamics'), indent=1)[:1500] if 'extra' in d else list(d))
PY
