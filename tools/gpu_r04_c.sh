#!/bin/bash
# Round-4 GPU pass C: the whole GPU suite on the Philox-7 build, then the Cholesky-pivot A/B on one box.
set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
STEPS="${PHF_STEPS:-pytest chol}"
want() { [[ " $STEPS " == *" $1 "* ]]; }
step() { local name=$1 to=$2; shift 2
  echo "== $name"; timeout -k 10 "$to" "$@" > "$R/gpurun_out/$name.log" 2>&1; local rc=$?
  echo "$name rc=$rc"; tail -n ${PHF_TAIL:-6} "$R/gpurun_out/$name.log" | cut -c1-1500
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out - stopping"; exit 1; fi; return $rc; }
if want pytest; then PHF_TAIL=25 step pytest_gpu 1100 python -m pytest tests -m gpu -q --timeout 900 -rs ${PHF_PYTEST_ARGS:-}; fi
if want chol; then PHF_AB_ROUNDS=2 step ab_chol 500 bash tools/ab_sl.sh tools/_build/exp/libexp_chol_split.so default; fi
