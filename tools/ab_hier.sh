#!/bin/bash
# A/B of experimental builds of the library on ONE box: for every lib given (paths under tools/_build/exp/, or "default"), in
# turn and PHF_AB_ROUNDS times over, the one-lane / two-lane kernels per Ne group alone (tools/diag_hier_lanes.py) and C4.
#   bash tools/ab_hier.sh default tools/_build/exp/libexp_a.so ...
set -u
mkdir -p gpurun_out
R=${GRAFT_REPO_ROOT:-.}
for round in $(seq 1 ${PHF_AB_ROUNDS:-2}); do
  for L in "$@"; do
    echo "== round $round  $L"
    PHF_DIAG_NE=${PHF_DIAG_NE:-3,4} PHF_DIAG_SHAPES=${PHF_DIAG_SHAPES:-1024} timeout -k 10 200 python $R/tools/exp_run.py $L $R/tools/diag_hier_lanes.py 2>&1 | grep "chains/pair" | cut -c1-200
    timeout -k 10 200 python $R/tools/exp_run.py $L $R/bench.py --workload c4 --steps 10 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('   c4 ms_per_step %.3f  value %.4g' % (d['ms_per_step'], d['value']))"
  done
done
