#!/bin/bash
# PMC of phf_hier3_advance (the hand-allocated gfx950 build of the Ne = 3 iteration) against the hipcc one-lane kernel, on the 147
# uniform pairs (tools/diag_isa_ne3.py; $1 = chains per pair, default 1024).  Counters only (--kernel-trace + --pmc), one pass per set.
# Produced profiles/r05/c4_isa_ne3_pmc*.txt
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=${1:-1024}
export PHF_DIAG_ONLY=${2:-isa}        # isa | hipcc: one kernel, launches of 2 000 iterations only (every dispatch the same)
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmci_$i -- python $R/tools/diag_isa_ne3.py $C > $R/gpurun_out/pmci_$i.log 2>&1
  rc=$?; echo "pmc pass $i rc=$rc"; tail -n 2 $R/gpurun_out/pmci_$i.log | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
cd $R && python tools/pmc_summary.py gpurun_out/pmci_*/ > gpurun_out/pmc_${PHF_DIAG_ONLY}_ne3_$C.txt 2>&1; cat gpurun_out/pmc_${PHF_DIAG_ONLY}_ne3_$C.txt | cut -c1-120
