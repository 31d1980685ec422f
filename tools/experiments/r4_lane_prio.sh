#!/bin/bash
# HIP stream priorities of the two lanes (IRA_LANE_PRIO, lower = served first): headline bench, alternating.
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r4_prio}; mkdir -p $O
for rep in 1 2; do
  for pr in "0,0" "-1,0" "0,-1"; do
    tag=$(echo $pr | tr ',-' '_m')
    IRA_LANE_PRIO="$pr" timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --variants value --literal-steps 0 > $O/p${tag}_$rep.json 2> $O/p${tag}_$rep.err || echo failed
    python3 -c "
import json; d=json.load(open('$O/p${tag}_$rep.json')); print('prio $pr rep $rep:', round(d['value'],1), 'IRs/s', round(d['ms_per_step'],2), 'ms')"
  done
done
