#!/bin/bash
# per-kernel times and wait counters of the hierarchical kernels (PHF_HIER_LANES=1|2: lanes per chain), C4
set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for wps in 1 2; do
  export PHF_HIER_LANES=$wps
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/hprof_wps$wps -- python $R/bench.py --workload c4 --steps 3 --warmup 4 --no-cpu-baseline > $R/gpurun_out/hprof_wps$wps.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/hpmc_wps$wps -- python $R/bench.py --workload c4 --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/hpmc_wps$wps.log 2>&1 || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/hpmc2_wps$wps -- python $R/bench.py --workload c4 --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/hpmc2_wps$wps.log 2>&1 || exit 1
  echo "== wps $wps"; head -5 $R/gpurun_out/hprof_wps$wps/*/*kernel_stats.csv | cut -c1-160
  (cd $R && python tools/pmc_summary.py gpurun_out/hpmc_wps$wps/ gpurun_out/hpmc2_wps$wps/ | grep -A17 "advance_kernel<3")
done
