#!/bin/bash
# PMC of the hierarchical kernel variants alone (Ne = 3 group, 1 024 chains per pair): one lane (512 registers) against two lanes
# (256 registers, two wavefronts per SIMD).  Counters only (--kernel-trace + --pmc), one pass per set.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/rocprof_counters.txt 2>&1
export PHF_DIAG_NE=${PHF_DIAG_NE:-3} PHF_DIAG_SHAPES=1024 PHF_DIAG_VARIANTS=${PHF_DIAG_VARIANTS:-}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmch_$i -- python $R/tools/diag_hier_lanes.py > $R/gpurun_out/pmch_$i.log 2>&1
  rc=$?; echo "pmc pass $i rc=$rc"; tail -n 2 $R/gpurun_out/pmch_$i.log | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
cd $R && python tools/pmc_summary.py gpurun_out/pmch_*/ > gpurun_out/pmc_hier_lanes_summary.txt 2>&1; cat gpurun_out/pmc_hier_lanes_summary.txt | cut -c1-120
