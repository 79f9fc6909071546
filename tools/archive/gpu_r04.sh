#!/bin/bash
# Round-4 GPU pass: the whole GPU suite, ONE default bench (the driver's command), the 2-rank rehearsal, the host-side profile of the
# command lines.  PHF_STEPS selects.
set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
STEPS="${PHF_STEPS:-pytest bench two_ranks cli}"
want() { [[ " $STEPS " == *" $1 "* ]]; }
step() { local name=$1 to=$2; shift 2
  echo "== $name"; timeout -k 10 "$to" "$@" > "$R/gpurun_out/$name.log" 2>&1; local rc=$?
  echo "$name rc=$rc"; tail -n ${PHF_TAIL:-6} "$R/gpurun_out/$name.log" | cut -c1-1800
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out - stopping"; exit 1; fi; return $rc; }
if want pytest; then PHF_TAIL=30 step pytest_gpu 1150 python -m pytest tests -m gpu -q --timeout 1000 -rs --durations=12 ${PHF_PYTEST_ARGS:-}; fi
if want bench; then step bench 500 python bench.py --gpus 1 --steps 20 --warmup 10; fi
if want two_ranks; then PHF_BENCH_BACKEND=gloo step bench_2rank 500 python bench.py --gpus 2 --steps 5 --warmup 3 --chains 2048 --iters-per-step 8000 --no-cpu-baseline; fi
if want cli; then
  PHF_TAIL=50 step cli_profile 600 python tools/profile_cli_host.py
  PHF_TAIL=50 step cli_profile_hier 600 python tools/profile_cli_host.py --hierarchical
fi
