#!/bin/bash
# Runs on the GPU box via gpurun: smoke -> gpu tests -> bench -> rocprof kernel trace.
# A step that is killed by its timeout stops the script (no further GPU work after a hang).
set -u
mkdir -p gpurun_out
step() {  # name timeout cmd...
  local name=$1 to=$2; shift 2
  echo "== $name" | tee -a gpurun_out/steps.log
  timeout -k 10 "$to" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a gpurun_out/steps.log
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "step $name timed out - stopping"; exit 1; fi
  return $rc
}
: > gpurun_out/steps.log
step smoke 300 python -c 'import __graft_entry__ as g; g.build(); g.smoke()' || exit 1
step pytest_gpu 900 python -m pytest tests -m gpu -q -x --timeout 600
step bench 400 python bench.py
cd /tmp && export TMPDIR=/tmp
step_dir=$GRAFT_REPO_ROOT
cd "$step_dir"
step rocprof 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 5 --no-cpu-baseline
ls -R gpurun_out/prof | head -30
