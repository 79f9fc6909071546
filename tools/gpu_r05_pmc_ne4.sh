#!/bin/bash
# PMC of phf_hier4_advance_s4441 (the gfx950 build of the Ne = 4 iteration) at $1 chains per pair (tools/diag_isa_ne4.py): one wavefront per
# SIMD (1024) against two (4096).  Counters only (--kernel-trace + --pmc), one pass per set.  -> profiles/r05/c4_ne4_assembly_pmc.txt
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=${1:-4096}
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_VALU_FMA_F64" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc4_${C}_$i -- python $R/tools/diag_isa_ne4.py $C > $R/gpurun_out/pmc4_${C}_$i.log 2>&1
  rc=$?; echo "pmc pass $i rc=$rc"; tail -n 2 $R/gpurun_out/pmc4_${C}_$i.log | cut -c1-300
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
done
cd $R && python tools/pmc_summary.py gpurun_out/pmc4_${C}_*/ > gpurun_out/pmc_ne4_$C.txt 2>&1; cat gpurun_out/pmc_ne4_$C.txt | cut -c1-160
