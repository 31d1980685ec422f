#!/bin/bash
# A/B of how the report blocks are dealt onto HIP streams (0 bands, 1 spectrum, 2 zplane, 3 decay, 4 modal, 5 stft):
#   bash tools/experiments/lane_ab.sh [repeats]      each deal `repeats` times, alternating, in one GPU call
run() { IRA_STREAMS=$1 IRA_LANE_DEAL="$2" timeout -k 10 300 python3 bench.py --no-cpu-baseline --variants value --literal-steps 0 --roofline-steps 1 > /tmp/o.json 2> /tmp/e.txt; python3 -c "import json,sys; d=json.loads(open('/tmp/o.json').read().strip().splitlines()[-1]); print('streams $1 deal $2', round(d['value']), round(d['ms_per_step'],2))" || tail -3 /tmp/e.txt; }
for i in $(seq 1 ${1:-3}); do run 2 "0,1,3|4,5,2"; run 2 "0,1|4,3,5,2"; run 2 "0,3,4|1,5,2"; run 2 "1,3,4|0,5,2"; run 3 "0,3|1|4,5,2"; run 1 ""; done
