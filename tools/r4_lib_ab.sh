#!/bin/bash
# A/B of two builds of the library on one box: kernel statistics of the serialised report step, alternating.
#   bash tools/r4_lib_ab.sh <outdir> <other.so> [config] [grep pattern]
R=$GRAFT_REPO_ROOT; O=${1:-gpurun_out/r4_ab}; other=$2; cfg=${3:-report}; pat=${4:-smooth_}
for rep in 1 2; do
  echo "== this build, rep $rep"; bash $R/tools/r4_stats.sh $O/new$rep $cfg 2>/dev/null | grep -i "$pat"
  echo "== $other, rep $rep"; IRA_TUNING=1 IRA_LIBRARY=$R/$other bash $R/tools/r4_stats.sh $O/old$rep $cfg 2>/dev/null | grep -i "$pat"
done
