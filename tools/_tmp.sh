set -e
timeout -k 10 400 python -m pytest tests/test_gpu_isa.py -x -q 2>&1 | tail -3
for isa in 0 1 0 1; do PHF_SL_ISA=$isa timeout -k 10 250 python bench.py --workload c3 --steps 5 --warmup 3 --no-cpu-baseline --no-other-workloads 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sl_isa=$isa c3 ms_per_step %.3f  value %.4g' % (d['ms_per_step'], d['value']))"; done
