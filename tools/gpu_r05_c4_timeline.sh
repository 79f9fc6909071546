#!/bin/bash
# When does each Ne group's kernel of a C4 step run?  rocprofv3 --kernel-trace of bench.py --workload c4 (start / end of every dispatch);
# tools/c4_timeline.py turns the trace into a per-stream timeline.  Produced profiles/r05/c4_timeline.txt
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c4tl -- python $R/bench.py --workload c4 --steps 6 --warmup 2 --no-cpu-baseline --no-other-workloads > $R/gpurun_out/c4tl.log 2>&1
echo rc=$?; tail -n 1 $R/gpurun_out/c4tl.log | cut -c1-200
cd $R && python tools/c4_timeline.py gpurun_out/c4tl > gpurun_out/c4_timeline.txt 2>&1; tail -n 60 gpurun_out/c4_timeline.txt
