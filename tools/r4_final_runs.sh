#!/bin/bash
# Everything profiles/ needs for the committed build, in one or two GPU calls:
#   bash tools/r4_final_runs.sh bench    -> the six bench lines (report, literal, 2, 3, 4, 5)
#   bash tools/r4_final_runs.sh prof     -> kernel stats + FETCH/WRITE/RDREQ passes of report, literal, 2, 3, 4, 5
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4_final; mkdir -p $O
what=${1:-bench}
if [ "$what" = bench ]; then
  for c in report literal 2 3 4 5; do
    timeout -k 10 400 python3 $R/bench.py --config $c > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err || echo "bench $c failed" >> $O/fail.log
    grep "timed region" $O/bench_cfg$c.err
  done
else
  shift
  for c in ${@:-report literal 2 3 4 5}; do
    bash $R/tools/profile_config.sh $c gpurun_out/r4_final/prof_$c > $O/prof_$c.log 2>&1 || echo "profile $c failed" >> $O/fail.log
    ls $O/prof_$c | head -3
  done
fi
ls $O
