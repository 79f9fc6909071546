#!/bin/bash
# hierarchical kernels: parity tests, then C4 with the one-lane and two-lanes-per-chain kernels (A/B on one box)
set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
step() { local name=$1 to=$2; shift 2
  echo "== $name"; timeout -k 10 "$to" "$@" > "$R/gpurun_out/$name.log" 2>&1; local rc=$?
  echo "$name rc=$rc"; tail -n ${PHF_TAIL:-4} "$R/gpurun_out/$name.log" | cut -c1-700
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out - stopping"; exit 1; fi; return $rc; }
step pytest_hier 600 python -m pytest tests/test_gpu_hierarchical.py -m gpu -q -x --timeout 500 -k "${PHF_TEST_FILTER:-bit_identical or golden}" || exit 1
PHF_HIER_LANES=1 step c4_wps1 300 python bench.py --workload c4 --steps 5 --warmup 4 --no-cpu-baseline
PHF_HIER_LANES=2 step c4_wps2 300 python bench.py --workload c4 --steps 5 --warmup 4 --no-cpu-baseline
step c4_auto 300 python bench.py --workload c4 --steps 5 --warmup 4 --no-cpu-baseline
