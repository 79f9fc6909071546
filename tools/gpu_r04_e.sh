#!/bin/bash
# Round-4 GPU pass E: hierarchical A/B on one box — shipped library against experimental builds (tools/_build/exp).
set -u
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
PHF_AB_ROUNDS=${PHF_AB_ROUNDS:-2} PHF_DIAG_NE=${PHF_DIAG_NE:-3,4} bash tools/ab_hier.sh ${PHF_LIBS:-default tools/_build/exp/libexp_both.so} 2>&1 | tee gpurun_out/ab_hier_e.log
