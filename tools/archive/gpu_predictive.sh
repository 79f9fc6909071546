#!/bin/bash
# GPU loop for the posterior-predictive curves: tests, then the row's measurement (tools/bench_predictive.py) plain and under rocprofv3
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 600 -k "${PHF_TEST_FILTER:-predictive or cli}" > gpurun_out/pytest_pred.log 2>&1; rc=$?; tail -n 8 gpurun_out/pytest_pred.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python tools/bench_predictive.py > gpurun_out/bench_predictive.log 2>&1; rc=$?; tail -n 2 gpurun_out/bench_predictive.log | cut -c1-900
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 400 python tools/bench_predictive.py --chains 64 --rows 4000 --cpu-samples 2000 > gpurun_out/bench_predictive_c64.log 2>&1; rc=$?; tail -n 1 gpurun_out/bench_predictive_c64.log | cut -c1-900
if [ $rc -ne 0 ]; then exit 1; fi
rm -rf gpurun_out/prof_pred
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pred -- python tools/bench_predictive.py --cpu-samples 2000 > gpurun_out/prof_pred.log 2>&1; rc=$?
tail -n 1 gpurun_out/prof_pred.log | cut -c1-300
find gpurun_out/prof_pred -name "*kernel_stats.csv" | head -1 | xargs -r head -n 6
