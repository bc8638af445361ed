#!/bin/bash
# headline only, for each library variant under build/
cd $GRAFT_REPO_ROOT
for so in robovat_amd/librovat_hip.so build/librovat_*.so; do
  for rep in 1 2; do
    v=$(RV_LIB=$GRAFT_REPO_ROOT/$so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f  kernel %.1f ms' % (d['value'], d['roofline']['avg_kernel_ms']))")
    echo "$so: $v"
  done
done
