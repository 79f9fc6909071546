#!/bin/bash
# quick GPU loop: smoke + bit-exactness tests + bench (no cpu baseline)
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -c 'import __graft_entry__ as g; g.build(); g.smoke()' > gpurun_out/smoke.log 2>&1; rc=$?; tail -n 3 gpurun_out/smoke.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 600 python -m pytest tests -m gpu -q -x --timeout 500 -k "${PHF_TEST_FILTER:-bit_identical or philox or log_target or shard}" > gpurun_out/pytest_gpu.log 2>&1; rc=$?; tail -n 8 gpurun_out/pytest_gpu.log
if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit 1; fi
timeout -k 10 300 python bench.py --no-cpu-baseline ${PHF_BENCH_ARGS:-} > gpurun_out/bench.log 2>&1; tail -n 2 gpurun_out/bench.log | cut -c1-600
