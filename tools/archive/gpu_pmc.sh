#!/bin/bash
# PMC passes on the bench (counters only; no tracing flags besides kernel-trace)
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="${PHF_BENCH_ARGS:---steps 3 --warmup 2 --no-cpu-baseline}"
PROG="${PHF_PMC_PROG:-bench.py}"          # e.g. PHF_PMC_PROG=tools/bench_predictive.py PHF_BENCH_ARGS="--cpu-samples 200 --steps 3"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc$i -- python $R/$PROG $ARGS > $R/gpurun_out/pmc$i.log 2>&1
  rc=$?; echo "pmc pass $i rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
cd $R && python tools/pmc_summary.py gpurun_out/pmc*/ 2>&1 | tail -40
