#!/bin/bash
# Full GPU round: smoke, all gpu tests, benches for every workload, rocprof kernel-trace stats of the default bench.
set -u
mkdir -p gpurun_out
step() { local name=$1 to=$2; shift 2
  echo "== $name"; timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?
  echo "$name rc=$rc"; tail -n 4 "gpurun_out/$name.log" | cut -c1-900
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out - stopping"; exit 1; fi; return $rc; }
step smoke 300 python -c 'import __graft_entry__ as g; g.build(); g.smoke()' || exit 1
step pytest_gpu 1000 python -m pytest tests -m gpu -q --timeout 900
step bench 400 python bench.py
step bench_c3 300 python bench.py --workload c3 --steps 4 --warmup 4 --iters-per-step 1000 --no-cpu-baseline
step bench_c5 300 python bench.py --workload c5 --steps 3 --warmup 8 --iters-per-step 500 --no-cpu-baseline
step bench_c4 400 python bench.py --workload c4 --steps 3 --warmup 4 --iters-per-step 500 --no-cpu-baseline
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
step rocprof 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -- python $R/bench.py --steps 5 --no-cpu-baseline
step rocprof_c4 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c4 -- python $R/bench.py --workload c4 --steps 2 --warmup 4 --iters-per-step 500 --no-cpu-baseline
