#!/bin/bash
# Round-5 profiling pass: for every driver-timed region of bench.py (c3 headline, c2, c4, c5, c5_moments, c3_model1, s3, s3h) ONE
# `rocprofv3 --kernel-trace --stats` run and five `--pmc` passes (counters only, one set per run) of the same bench command,
# summarised per kernel into gpurun_out/pmc_<name>_summary.txt; tools/collect_profiles.py copies them into profiles/r05/.
# PHF_NAMES selects the regions (default all), PHF_STEPS what runs (stats pmc).
set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
NAMES="${PHF_NAMES:-c3 c2 c4 c5 c5_moments c3_model1 s3 s3h}"
STEPS="${PHF_STEPS:-stats pmc}"
want() { [[ " $STEPS " == *" $1 "* ]]; }
args_of() { case $1 in
  c3) echo "--workload c3";; c2) echo "--workload c2";; c4) echo "--workload c4";; c5) echo "--workload c5";;
  c5_moments) echo "--workload c5 --moments";; c3_model1) echo "--workload c3 --model 1";; s3) echo "--workload s3";; s3h) echo "--workload s3h";; esac; }
step() { local name=$1 to=$2; shift 2
  timeout -k 10 "$to" "$@" > "$R/gpurun_out/$name.log" 2>&1; local rc=$?
  echo "$name rc=$rc $(grep -o '"ms_per_step": [0-9.]*' "$R/gpurun_out/$name.log" | head -1)"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out - stopping"; exit 1; fi; return $rc; }
cd /tmp && export TMPDIR=/tmp
for N in $NAMES; do
  A="$(args_of $N) --steps 3 --warmup 2 --no-cpu-baseline --no-other-workloads"
  if want stats; then rm -rf $R/gpurun_out/prof_$N; step rocprof_$N 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$N -- python $R/bench.py $A; fi
  if want pmc; then
    i=0
    for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
               "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_ACTIVE_INST_LDS" \
               "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
               "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
      i=$((i+1)); rm -rf $R/gpurun_out/pmc_${N}_$i
      step pmc_${N}_$i 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_${N}_$i -- python $R/bench.py $A
    done
    (cd $R && rm -f gpurun_out/pmc_${N}_summary.txt && python tools/pmc_summary.py gpurun_out/pmc_${N}_[0-9]*/ > gpurun_out/pmc_${N}_summary.txt 2>&1)
  fi
done
