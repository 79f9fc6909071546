#!/bin/bash
# tools/build_exp_sl.sh NAME SRC.hip [extra hipcc flags]: an experimental libexp_NAME.so (timing / A-B only) from a copy of
# pyhillfit_amd/csrc/phf_single_level.hip, linked with the current objects of the other translation units.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; shift 2
out=$R/tools/_build/exp
mkdir -p $out
sed -e "s#\"../../include/pyhillfit_amd.h\"#\"$R/include/pyhillfit_amd.h\"#" $src > $out/tmp_$name.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function -Wno-pass-failed \
  -I$R/pyhillfit_amd/csrc "$@" -c -o $out/exp_$name.o $out/tmp_$name.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libexp_$name.so $R/pyhillfit_amd/lib/obj/phf_capi.o \
  $out/exp_$name.o $R/pyhillfit_amd/lib/obj/phf_hierarchical.o $R/pyhillfit_amd/lib/obj/phf_predictive.o $R/pyhillfit_amd/lib/obj/phf_hier3_isa.o
rm -f $out/tmp_$name.hip
echo built $out/libexp_$name.so
