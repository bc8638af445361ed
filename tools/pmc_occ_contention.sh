R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05_t; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for V in n1k n8k; do
  if [ $V = n1k ]; then A="--steps 2 --warmup 1 --over PHYSICS.SLEEP_STEPS=0"; else A="--steps 1 --warmup 1 --envs-per-gpu 8192 --over PHYSICS.SLEEP_STEPS=0"; fi
  CMD="python $R/bench.py $A --no-cpu-baseline --no-extra-legs"
  RV_ENV_OCC=2 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM -d $O/a_$V -o a --output-format csv -- $CMD > $O/a_$V.log 2>&1
  RV_ENV_OCC=2 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES -d $O/b_$V -o b --output-format csv -- $CMD > $O/b_$V.log 2>&1
done
cd $R; python tools/pmc_summary.py $O/a_n1k $O/a_n8k $O/b_n1k $O/b_n8k | tee $O/occ_contention.txt
