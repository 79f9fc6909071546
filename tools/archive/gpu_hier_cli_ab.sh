#!/bin/bash
# hierarchical command line in the throughput regime (1024 chains per pair): segments joined (PHF_JOIN_SEGMENTS=1) vs free-running streams
set -u
mkdir -p gpurun_out /tmp/phf_cli
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from pyhillfit_amd import doseresponse as dr
dr.setup("data/crumb_dataset.json"); dr.table.to_csv("/tmp/phf_cli/crumb_data.csv")
PY
for j in 1 0 1 0; do
  rm -rf /tmp/phf_cli/out_ab
  PHF_JOIN_SEGMENTS=$j timeout -k 10 600 python python/PyHillFit.py --data-file /tmp/phf_cli/crumb_data.csv -m 2 -a --hierarchical --num-chains ${PHF_AB_CHAINS:-1024} --iterations ${PHF_AB_ITERS:-60000} --segment ${PHF_AB_SEGMENT:-5000} --output-root /tmp/phf_cli/out_ab > gpurun_out/cli_ab_$j.log 2>&1
  echo "join=$j rc=$? $(grep '^timing' gpurun_out/cli_ab_$j.log)"
done
