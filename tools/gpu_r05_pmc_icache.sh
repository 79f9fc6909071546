#!/bin/bash
# PMC of the instruction cache and the wait states of a bench region ($1: c4 | s3h ...): the fused grid's many bodies against one kernel's.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
W=${1:-c4}
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_ANY"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmcic_${W}_$i
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmcic_${W}_$i -- python $R/bench.py --workload $W --steps 3 --warmup 2 --no-cpu-baseline --no-other-workloads > $R/gpurun_out/pmcic_${W}_$i.log 2>&1
  rc=$?; echo "pass $i rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
cd $R && python tools/pmc_summary.py gpurun_out/pmcic_${W}_*/ 2>&1 | grep -A18 "fused\|phf_hier3_advance$" | head -60
