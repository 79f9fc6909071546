#!/bin/bash
# Round-4 GPU pass A: smoke, the device-numerics tests, ONE default bench, Philox 7 vs 10 on one box, hierarchical kernel variants.
set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
STEPS="${PHF_STEPS:-smoke quick bench philox hier c4}"
want() { [[ " $STEPS " == *" $1 "* ]]; }
step() { local name=$1 to=$2; shift 2
  echo "== $name"; timeout -k 10 "$to" "$@" > "$R/gpurun_out/$name.log" 2>&1; local rc=$?
  echo "$name rc=$rc"; tail -n ${PHF_TAIL:-6} "$R/gpurun_out/$name.log" | cut -c1-1500
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timed out - stopping"; exit 1; fi; return $rc; }
if want smoke; then step smoke 300 python -c 'import __graft_entry__ as g; g.build(); g.smoke()' || exit 1; fi
if want quick; then step quick 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 500 -k "philox or bit_identical or twin or golden" || exit 1; fi
if want bench; then step bench 500 python bench.py --gpus 1 --steps 20 --warmup 10; fi
if want philox; then PHF_AB_ROUNDS=2 step ab_philox 500 bash tools/ab_sl.sh tools/_build/exp/libexp_philox10.so default; fi
if want hier; then PHF_DIAG_NE=3,4 PHF_DIAG_SHAPES=1024 PHF_DIAG_VARIANTS="1:2,2:2" step hier_lanes 400 python tools/diag_hier_lanes.py; fi
if want c4; then
  for I in 500 2000; do step c4_$I 300 python bench.py --workload c4 --iters-per-step $I --steps 8 --warmup 4 --no-cpu-baseline; done
  step c4_p10 300 python tools/exp_run.py tools/_build/exp/libexp_philox10.so bench.py --workload c4 --iters-per-step 2000 --steps 8 --warmup 4 --no-cpu-baseline
fi
