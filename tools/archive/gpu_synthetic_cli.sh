#!/bin/bash
# Drop-in run on the reference's synthetic data set (3 drugs, Ne = 5, 5, 50: the last one runs the generic hierarchical kernel)
set -u
mkdir -p gpurun_out /tmp/phf_syn
python - <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
from pyhillfit_amd import doseresponse as dr
dr.setup("data/synthetic_dataset.json"); dr.table.to_csv("/tmp/phf_syn/synthetic_data.csv")
print(dr.drugs, dr.channels)
PY
SECONDS=0; timeout -k 10 600 python python/PyHillFit.py --data-file /tmp/phf_syn/synthetic_data.csv -m 2 -a --num-chains 64 -i 20000 --output-root /tmp/phf_syn/output > gpurun_out/syn_m2.log 2>&1
echo "single-level rc=$? wall=${SECONDS}s"; grep "^timing" gpurun_out/syn_m2.log
SECONDS=0; timeout -k 10 900 python python/PyHillFit.py --data-file /tmp/phf_syn/synthetic_data.csv -m 2 -a --hierarchical --predictive-cdfs --num-chains 64 -i 20000 --output-root /tmp/phf_syn/output > gpurun_out/syn_hier.log 2>&1
echo "hierarchical rc=$? wall=${SECONDS}s"; grep "^timing" gpurun_out/syn_hier.log; tail -n 3 gpurun_out/syn_hier.log | cut -c1-200
find /tmp/phf_syn/output -name "*.txt" | wc -l
python - <<'PY'
import glob, json
for f in sorted(glob.glob("/tmp/phf_syn/output/synthetic_data/hierarchical/*/*/*_expts/chain/*_summary.json")):
    s = json.load(open(f)); print(s["drug"], s["num_expts"], "acceptance %.3f" % s["acceptance"], "alpha %.3f mu %.3f" % (s["pooled_mean"][0], s["pooled_mean"][2]))
PY
