#!/bin/bash
# Config 5 on one GPU: reader-thread prefetch depth x interpreter switch interval, alternating (bench.py --config 5).
R=$GRAFT_REPO_ROOT
for rep in 1 2; do
  for arm in "IRA_BUNDLE_PREFETCH=1 IRA_BUNDLE_SWITCH_US=0" "IRA_BUNDLE_PREFETCH=2 IRA_BUNDLE_SWITCH_US=0" "IRA_BUNDLE_PREFETCH=1 IRA_BUNDLE_SWITCH_US=500" "IRA_BUNDLE_PREFETCH=2 IRA_BUNDLE_SWITCH_US=500" "IRA_BUNDLE_PREFETCH=2 IRA_BUNDLE_SWITCH_US=100" "IRA_BUNDLE_PREFETCH=3 IRA_BUNDLE_SWITCH_US=500"; do
    env $arm IRA_BUNDLE_TIMING=1 timeout -k 10 300 python3 $R/bench.py --config 5 --no-cpu-baseline > /tmp/b5.json 2> /tmp/b5.err || echo failed
    python3 -c "
import json; d=json.load(open('/tmp/b5.json')); print('$arm:', round(d['value']), d['unit'], round(d['ms_per_step'],2), 'ms/step')"
    grep "host ms per group" /tmp/b5.err | tail -1 | cut -c1-260
  done
done
