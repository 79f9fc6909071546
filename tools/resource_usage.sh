#!/bin/bash
# register / scratch / occupancy report of the kernels of one source file: tools/resource_usage.sh phf_hierarchical.hip [filter]
f=${1:-phf_single_level.hip}; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-pass-failed \
  -Rpass-analysis=kernel-resource-usage -c -o /tmp/ru_$$.o "$(dirname "$0")/../pyhillfit_amd/csrc/$f" 2>&1 \
 | sed -n 's/.*remark: *\(.*\) \[-Rpass.*/\1/p' \
 | awk '/Function Name/{if(l)print l; l=$3} /VGPRs:|AGPRs:|ScratchSize|Occupancy|SGPRs:|LDS Size/{l=l" "$0} END{print l}' \
 | sed 's/  */ /g' | grep -E "$pat" | while read -r line; do n=$(echo "$line" | cut -d' ' -f1 | c++filt | cut -c1-70); echo "$n |$(echo "$line" | cut -d' ' -f2-)"; done
rm -f /tmp/ru_$$.o
