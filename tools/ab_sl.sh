#!/bin/bash
# A/B of experimental builds of the single-level kernels on ONE box: bench.py (c3 by default) per lib, PHF_AB_ROUNDS rounds.
set -u
R=${GRAFT_REPO_ROOT:-.}
for round in $(seq 1 ${PHF_AB_ROUNDS:-2}); do
  for L in "$@"; do
    for W in ${PHF_AB_WORKLOADS:-c3}; do
      timeout -k 10 300 python $R/tools/exp_run.py $L $R/bench.py --workload $W --steps 10 --warmup 5 --no-cpu-baseline --no-other-workloads 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('round $round  %-44s $W ms_per_step %.3f  value %.4g' % ('$L', d['ms_per_step'], d['value']))"
    done
  done
done
