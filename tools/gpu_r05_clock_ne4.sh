cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for C in 1024 4096; do
  rm -rf $R/gpurun_out/grbm_$C
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/grbm_$C -- python $R/tools/diag_isa_ne4.py $C > $R/gpurun_out/grbm_$C.log 2>&1
  echo "C=$C rc=$?"
done
