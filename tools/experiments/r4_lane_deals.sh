#!/bin/bash
# Which report blocks share a lane (HIP stream): headline bench per deal, alternating (IRA_LANE_DEAL: 0 bands, 1 spectrum,
# 2 zplane, 3 decay, 4 modal, 5 stft).  bash tools/experiments/r4_lane_deals.sh <outdir> [reps]
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r4_deals}; mkdir -p $O; reps=${2:-2}
for rep in $(seq 1 $reps); do
  for deal in "0,1,3|4,5,2" "0,5,2,3|4,1" "0,2,3|4,5,1" "1,5,3|0,4,2" "1,2,3,5|0,4" "0,3|1,2|4,5"; do
    n=$(echo "$deal" | tr -cd '|' | wc -c); n=$((n + 1))
    tag=$(echo "$deal" | tr ',|' '_-')
    IRA_STREAMS=$n IRA_LANE_DEAL="$deal" timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --variants value --literal-steps 0 > $O/d${tag}_$rep.json 2> $O/d${tag}_$rep.err || echo "failed $deal"
    python3 -c "
import json; d=json.load(open('$O/d${tag}_$rep.json')); print('deal $deal rep $rep:', round(d['value'],1), 'IRs/s', round(d['ms_per_step'],2), 'ms')"
  done
done
