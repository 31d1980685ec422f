#!/bin/bash
# Everything profiles/ needs for the committed build, in two or three GPU calls (one runner for every round; rounds 2-4 had
# r2_/r3_/r4_final_runs.sh):
#   bash tools/final_runs.sh <tag> bench [configs]   -> the bench lines (report, literal, 2, 3, 4, 5) under gpurun_out/<tag>_final
#   bash tools/final_runs.sh <tag> prof  [configs]   -> kernel stats + FETCH / WRITE / RDREQ passes (tools/profile_config.sh)
# then copy gpurun_out/<tag>_final/bench_cfg*.json and the prof_* tables to profiles/<tag>_*.
R=$GRAFT_REPO_ROOT; tag=${1:-r05}; what=${2:-bench}; shift 2
O=$R/gpurun_out/${tag}_final; mkdir -p $O
if [ "$what" = bench ]; then
  for c in ${@:-report literal 2 3 4 5}; do
    timeout -k 10 500 python3 $R/bench.py --config $c > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err || echo "bench $c failed" >> $O/fail.log
    grep "timed region" $O/bench_cfg$c.err
  done
else
  for c in ${@:-report literal 2 3 4 5}; do
    bash $R/tools/profile_config.sh $c gpurun_out/${tag}_final/prof_$c > $O/prof_$c.log 2>&1 || echo "profile $c failed" >> $O/fail.log
    ls $O/prof_$c | head -3
  done
fi
ls $O
