#!/bin/bash
# End-to-end drop-in run at the reference's defaults (500 000 iterations, thinning 5, burn-in 1/4) on all 210 Crumb pairs.
set -u
mkdir -p gpurun_out /tmp/phf_cli
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from pyhillfit_amd import doseresponse as dr
dr.setup("data/crumb_dataset.json"); dr.table.to_csv("/tmp/phf_cli/crumb_data.csv")
PY
for mode in "-m 2" "-m 1" "-m 2 --hierarchical --predictive-cdfs"; do
  tag=$(echo $mode | tr -d ' -')
  SECONDS=0; timeout -k 10 900 python python/PyHillFit.py --data-file /tmp/phf_cli/crumb_data.csv $mode -a --num-chains 64 --output-root /tmp/phf_cli/output > gpurun_out/cli_$tag.log 2>&1
  echo "$mode rc=$? wall=${SECONDS}s"; grep "^timing" gpurun_out/cli_$tag.log
done
for m in 1 2; do
  SECONDS=0; timeout -k 10 600 python python/PyHillTemp.py --data-file /tmp/phf_cli/crumb_data.csv -m $m -d 0 -c 0 --num-chains 64 --output-root /tmp/phf_cli/output > gpurun_out/cli_temp_m$m.log 2>&1
  echo "PyHillTemp -m $m -d 0 -c 0 (41 rungs) rc=$? wall=${SECONDS}s"; grep "MCMC time" gpurun_out/cli_temp_m$m.log
done
SECONDS=0; timeout -k 10 300 python python/compute_bayes_factors.py --data-file /tmp/phf_cli/crumb_data.csv -d 0 -c 0 --output-root /tmp/phf_cli/output --bf-dir /tmp/phf_cli/BFs/ > gpurun_out/cli_bf.log 2>&1
echo "compute_bayes_factors rc=$? wall=${SECONDS}s"; tail -n 2 gpurun_out/cli_bf.log; cat /tmp/phf_cli/BFs/*B12.txt 2>/dev/null | head -5
SECONDS=0; timeout -k 10 900 python python/construct_hierarchical_cdfs.py --data-file /tmp/phf_cli/crumb_data.csv -a --num-cores 15 --output-root /tmp/phf_cli/output2 > gpurun_out/cli_cdfs_nofiles.log 2>&1
echo "construct_hierarchical_cdfs (no chain files: every pair reported and skipped) rc=$? wall=${SECONDS}s"
SECONDS=0; timeout -k 10 900 python python/construct_hierarchical_cdfs.py --data-file /tmp/phf_cli/crumb_data.csv -a --num-cores 15 --output-root /tmp/phf_cli/output > gpurun_out/cli_cdfs.log 2>&1
echo "construct_hierarchical_cdfs rc=$? wall=${SECONDS}s"
find /tmp/phf_cli/output -name "*cdf.txt" | wc -l
find /tmp/phf_cli/output -name "*chain*.txt" | wc -l
du -sh /tmp/phf_cli/output
python - <<'PY'
import json, glob, numpy as np
for pat, name in (("/tmp/phf_cli/output/crumb_data/single-level/*/*/model_2/temperature_1/chain/*_summary.json", "model 2"),
                  ("/tmp/phf_cli/output/crumb_data/single-level/*/*/model_1/temperature_1/chain/*_summary.json", "model 1"),
                  ("/tmp/phf_cli/output/crumb_data/hierarchical/*/*/*_expts/chain/*_summary.json", "hierarchical")):
    fs = glob.glob(pat)
    ss = [json.load(open(f)) for f in fs]
    if ss:
        print(name, len(ss), "pairs; sampler MH samples/s:", "%.3g" % ss[0]["mh_samples_per_second"], "mean acceptance %.3f" % np.mean([s["acceptance"] for s in ss]))
PY
