#!/bin/bash
# Round-2 GPU pass: smoke, gpu tests, default bench (c3, the metric's config) with cpu baseline, the other workloads,
# a 2-rank rehearsal of `bench.py --gpus 2` on the one GPU (gloo), rocprof kernel stats + PMC passes of the default bench.
# PHF_STEPS selects steps (space-separated names), default all.
set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
STEPS="${PHF_STEPS:-smoke pytest bench c2 c4 c5 two_ranks rocprof pmc}"
want() { [[ " $STEPS " == *" $1 "* ]]; }
step() { local name=$1 to=$2; shift 2
  echo "== $name"; timeout -k 10 "$to" "$@" > "$R/gpurun_out/$name.log" 2>&1; local rc=$?
  echo "$name rc=$rc"; tail -n 4 "$R/gpurun_out/$name.log" | cut -c1-1200
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out - stopping"; exit 1; fi; return $rc; }
if want smoke; then step smoke 300 python -c 'import __graft_entry__ as g; g.build(); g.smoke()' || exit 1; fi
if want pytest; then step pytest_gpu 1100 python -m pytest tests -m gpu -q --timeout 900 ${PHF_PYTEST_ARGS:-}; fi
if want bench; then step bench 500 python bench.py --gpus 1 --steps 20 --warmup 5; fi
if want c2; then step bench_c2 300 python bench.py --workload c2 --no-cpu-baseline; fi
if want c4; then step bench_c4 400 python bench.py --workload c4 --steps 5 --warmup 4 --no-cpu-baseline; fi
if want c5; then step bench_c5 400 python bench.py --workload c5 --steps 5 --warmup 8 --no-cpu-baseline; fi
if want two_ranks; then PHF_BENCH_BACKEND=gloo step bench_2rank 500 python bench.py --gpus 2 --steps 5 --warmup 3 --chains 2048 --no-cpu-baseline; fi
cd /tmp && export TMPDIR=/tmp
if want rocprof; then
  step rocprof 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python $R/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-other-workloads
  for W in ${PHF_ROCPROF_EXTRA:-}; do step rocprof_$W 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$W -- python $R/bench.py --workload $W --steps 3 --warmup 4 --no-cpu-baseline; done
fi
if want pmc; then
 for W in ${PHF_PMC_WORKLOAD:-c3}; do
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
             "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES SQ_INST_CYCLES_VMEM" \
             "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    step pmc_${W}_$i 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_${W}_$i -- python $R/bench.py --workload $W --steps 3 --warmup 2 --no-cpu-baseline --no-other-workloads
  done
  (cd $R && rm -f gpurun_out/pmc_${W}_summary.txt && python tools/pmc_summary.py gpurun_out/pmc_${W}_[0-9]*/ > gpurun_out/pmc_${W}_summary.txt 2>&1; tail -n 20 gpurun_out/pmc_${W}_summary.txt)
 done
fi
