#!/bin/bash
# One parameterised GPU-box script (replaces the per-experiment tools/gpu_r03_*.sh of round 3).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/gpu.sh <tag> <stage> [<stage> ...]'
# Everything a stage writes goes to gpurun_out/<tag>/ (merged back into the build container).
# Stages:
#   tests          pytest -m gpu (whole suite) + smoke()
#   tests:<expr>   pytest -m gpu -k <expr>
#   bench          the driver's command: python bench.py --steps 20 --warmup 5
#   qstress / qstress50   tools/queue_stress.py 16 / 50 runs (task queues vs plain launch, every word)
#   c5             bench.py --workload config5 alone
#   headline       bench.py headline only (no CPU legs, no extra legs)
#   k50            bench.py --steps 50 headline only
#   profile        rocprofv3 kernel stats + PMC passes of the headline (tools/profile_bench.sh)
#   profile_c5 / _c3 / _c4   the same for configs 5 / 3 / 4
#   profile_nd     the same for the reference-semantics (no deactivation) launch
#   parts          per-part shader-clock table of the bench launch (RV_PROFILE build)
#   parts_nd       per-part table with PHYSICS.SLEEP_STEPS=0, 2 steps
#   parts_c3 / parts_c4 / parts_c5   per-part tables of configs 3 / 4 / 5
#   variants       headline + configs 5 / 3 / 4 for the product library and every build/librovat_*.so
#   sweep          tools/parity_sweep.py for both builds of the env kernel
#   poison         pytest -m gpu with every build/librovat_poison_*.so (LDS scratch starts as garbage)
#   lanes          SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU lane-utilisation pass (headline and no-deactivation)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
export TMPDIR=/tmp
for STAGE in "$@"; do
  T0=$(date +%s)
  case $STAGE in
    tests)
      timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/time.txt
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/time.txt
      tail -3 $O/tests.log; tail -1 $O/smoke.log ;;
    tests:*)
      timeout 1500 python -m pytest tests -m gpu -x -q -k "${STAGE#tests:}" > $O/tests_k.log 2>&1; echo "tests -k rc=$?" >> $O/time.txt
      tail -5 $O/tests_k.log ;;
    bench)
      timeout 1700 python bench.py --steps 20 --warmup 5 --legs-out $O/bench_legs.json > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/time.txt
      python tools/bench_digest.py $O/bench.json ;;
    qstress)
      timeout 1500 python tools/queue_stress.py 16 > $O/queue_stress.txt 2>&1; echo "qstress rc=$?" >> $O/time.txt; tail -4 $O/queue_stress.txt ;;
    qstress50)
      timeout 2400 python tools/queue_stress.py 50 > $O/queue_stress50.txt 2>&1; echo "qstress50 rc=$?" >> $O/time.txt; tail -4 $O/queue_stress50.txt ;;
    c5)
      timeout 600 python bench.py --workload config5 --steps 20 --warmup 5 --no-cpu-baseline > $O/c5.json 2> $O/c5.err; cut -c1-300 $O/c5.json ;;
    headline)
      timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/headline.json 2> $O/headline.err
      cut -c1-400 $O/headline.json ;;
    k50)
      timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extra-legs > $O/k50.json 2> $O/k50.err
      cut -c1-300 $O/k50.json ;;
    profile)    bash tools/profile_bench.sh $TAG/profile > $O/profile.log 2>&1; tail -2 $O/profile.log ;;
    profile_nd) bash tools/profile_bench.sh $TAG/profile_nd --steps 2 --warmup 1 --over PHYSICS.SLEEP_STEPS=0 > $O/profile_nd.log 2>&1; tail -2 $O/profile_nd.log ;;
    profile_c5) bash tools/profile_bench.sh $TAG/profile_c5 --workload config5 --steps 20 --warmup 5 > $O/profile_c5.log 2>&1; tail -2 $O/profile_c5.log ;;
    profile_c3) bash tools/profile_bench.sh $TAG/profile_c3 --workload config3 --steps 10 --warmup 2 > $O/profile_c3.log 2>&1; tail -2 $O/profile_c3.log ;;
    profile_c4) bash tools/profile_bench.sh $TAG/profile_c4 --workload config4 --steps 10 --warmup 2 > $O/profile_c4.log 2>&1; tail -2 $O/profile_c4.log ;;
    parts)      timeout 400 python tools/prof_rollout.py --warm 1 --warm-steps 5 --top 10 > $O/parts.txt 2>&1; head -40 $O/parts.txt ;;
    parts_nd)   timeout 600 python tools/prof_rollout.py --warm 0 --steps 2 --top 4 --over PHYSICS.SLEEP_STEPS=0 > $O/parts_nd.txt 2>&1; head -48 $O/parts_nd.txt ;;
    parts_c3)   timeout 600 python tools/prof_rollout.py --warm 0 --envs 4096 --steps 10 --top 6 --over TASK_NAME=crossing LAYOUT_ID=0 MOVABLE_NAME=CONCAVE MAX_STEPS=10 > $O/parts_c3.txt 2>&1; head -40 $O/parts_c3.txt ;;
    parts_c4)   timeout 600 python tools/prof_rollout.py --warm 0 --envs 2048 --steps 10 --top 6 --grasp > $O/parts_c4.txt 2>&1; head -40 $O/parts_c4.txt ;;
    parts_c5)   timeout 400 python tools/prof_rollout.py --warm 1 --envs 8192 --steps 10 --top 6 > $O/parts_c5.txt 2>&1; head -40 $O/parts_c5.txt ;;
    lanes)
      cd /tmp
      for V in head nd; do
        if [ $V = head ]; then A="--steps 20 --warmup 5"; else A="--steps 2 --warmup 1 --over PHYSICS.SLEEP_STEPS=0"; fi
        CMD="python $R/bench.py $A --no-cpu-baseline --no-extra-legs"
        rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $O/lanes_$V -o l --output-format csv -- $CMD > $O/lanes_$V.log 2>&1
      done
      cd $R; python tools/pmc_summary.py $O/lanes_head $O/lanes_nd > $O/lanes.txt 2>&1; cat $O/lanes.txt ;;
    icache)
      # instruction-cache passes (SQC_ICACHE_*, SQ_IFETCH, SQ_IFETCH_LEVEL): the env kernel is ~0.7 MB of code, the
      # instruction cache 64 KB per two CUs -- headline, no-deactivation (1024 and 8192 envs) and config 5
      cd /tmp
      for V in head nd nd8k c5; do
        case $V in
          head) A="--steps 20 --warmup 5" ;;
          nd)   A="--steps 2 --warmup 1 --over PHYSICS.SLEEP_STEPS=0" ;;
          nd8k) A="--steps 1 --warmup 1 --envs-per-gpu 8192 --over PHYSICS.SLEEP_STEPS=0" ;;
          c5)   A="--workload config5 --steps 10 --warmup 2" ;;
        esac
        CMD="python $R/bench.py $A --no-cpu-baseline --no-extra-legs"
        rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES -d $O/ic_$V -o i --output-format csv -- $CMD > $O/ic_$V.log 2>&1
        rocprofv3 --kernel-trace --pmc SQ_IFETCH_LEVEL SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS -d $O/il_$V -o i --output-format csv -- $CMD > $O/il_$V.log 2>&1
      done
      cd $R; python tools/pmc_summary.py $O/ic_head $O/ic_nd $O/ic_nd8k $O/ic_c5 $O/il_head $O/il_nd $O/il_nd8k $O/il_c5 > $O/icache.txt 2>&1; cat $O/icache.txt ;;
    sweep)
      # HIP == float oracle bit for bit over seeds / configs, for both builds of the env kernel
      (echo "# tools/parity_sweep.py 8 256 6, register-rich build (RV_ENV_OCC=1)"; RV_ENV_OCC=1 timeout 1500 python tools/parity_sweep.py 8 256 6;
       echo "# tools/parity_sweep.py 4 256 6, two-waves-per-SIMD build (RV_ENV_OCC=2)"; RV_ENV_OCC=2 timeout 1500 python tools/parity_sweep.py 4 256 6) > $O/parity_sweep.txt 2>&1
      grep -c "True joints True" $O/parity_sweep.txt; grep MISMATCH $O/parity_sweep.txt ;;
    poison)
      # GPU parity tests with the LDS scratch poisoned at kernel start (tools/build_poison.py built the libraries)
      for SO in $(ls build/librovat_poison_*.so 2>/dev/null); do
        echo "== $SO" | tee -a $O/poison.txt
        RV_LIB=$R/$SO timeout 900 python -m pytest tests -m gpu -q --deselect tests/test_gpu_api.py -k "not smoke" 2>&1 | grep "passed\|failed" | tail -3 | tee -a $O/poison.txt
      done ;;
    variants)
      # every library variant under build/ (and the product library): headline, config 5 / 3 / 4 single-launch rollouts
      for SO in robovat_amd/librovat_hip.so $(ls build/librovat_*.so 2>/dev/null); do
        for W in "--steps 20 --warmup 5" "--workload config5 --steps 10 --warmup 2" "--workload config3 --steps 10 --warmup 2" "--workload config4 --steps 10 --warmup 2"; do
          V=$(RV_LIB=$R/$SO timeout 600 python bench.py $W --no-cpu-baseline --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f env-steps/s  kernel %.1f ms' % (d['value'], d['roofline']['avg_kernel_ms']))")
          echo "$SO [$W]: $V" | tee -a $O/variants.txt
        done
      done ;;
    *) echo "unknown stage $STAGE" ;;
  esac
  echo "$STAGE: $(( $(date +%s) - T0 )) s" >> $O/time.txt
done
cat $O/time.txt
