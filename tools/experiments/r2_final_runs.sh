#!/bin/bash
# Everything profiles/ needs for the committed build, in one GPU call: the five bench lines, then the profile passes
# (kernel stats + FETCH/WRITE/RDREQ counters) of the four single-GPU configurations.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2_final; mkdir -p $O
for c in report 2 3 4 5; do
  timeout -k 10 400 python3 $R/bench.py --config $c > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err || echo "bench $c failed" >> $O/fail.log
  grep "timed region" $O/bench_cfg$c.err
done
for c in report 2 3 4; do
  bash $R/tools/profile_config.sh $c gpurun_out/r2_final/prof_$c > $O/prof_$c.log 2>&1 || echo "profile $c failed" >> $O/fail.log
done
ls $O
