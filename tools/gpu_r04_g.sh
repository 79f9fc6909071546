#!/bin/bash
set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_start_points.py -m gpu -q -x -s 2>&1 | tail -15
PHF_AB_ROUNDS=2 PHF_DIAG_NE=3,4 bash tools/ab_hier.sh default tools/_build/exp/libexp_s_maxilp.so tools/_build/exp/libexp_s_memclause.so tools/_build/exp/libexp_s_trackers.so 2>&1 | tee gpurun_out/ab_hier_sched.log
