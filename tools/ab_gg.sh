cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in "$@"; do
  echo "== $L" >> gpurun_out/ab_gg.txt
  NMRGNN_HIP_LIB=$PWD/$L python tools/f256_ab.py 2>&1 | grep -E "step median|mp_gg|inference|mp_aggregate|mp_dw |mp_update|scatter" >> gpurun_out/ab_gg.txt
done
cat gpurun_out/ab_gg.txt
