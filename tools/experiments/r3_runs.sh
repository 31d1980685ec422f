#!/bin/bash
# Round 3: one GPU call = GPU tests, then the bench lines and profile passes asked for.
#     bash tools/experiments/r3_runs.sh <tag> [test] [bench:<cfg>...] [prof:<cfg>...]
R=$GRAFT_REPO_ROOT; tag=$1; shift; O=$R/gpurun_out/$tag; mkdir -p $O
for a in "$@"; do
  case $a in
    test)   timeout -k 10 900 python3 -m pytest $R/tests -m gpu -x -q > $O/gputest.log 2>&1; rc=$?; tail -3 $O/gputest.log; [ $rc -ne 0 ] && exit $rc ;;
    bench:*) c=${a#bench:}; timeout -k 10 400 python3 $R/bench.py --config $c > $O/bench_cfg$c.json 2> $O/bench_cfg$c.err || { echo "bench $c failed"; tail -5 $O/bench_cfg$c.err; exit 1; }
             python3 - $O/bench_cfg$c.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"].get("workload"), d["value"], d["unit"], "ms/step", d["ms_per_step"], "roofline", d.get("roofline", {}).get("frac"))
print({k: round(v, 3) for k, v in sorted(d.get("device_ms_per_step_by_call", {}).items(), key=lambda kv: -kv[1])[:14]})
PY
             ;;
    prof:*) c=${a#prof:}; bash $R/tools/profile_config.sh $c gpurun_out/$tag/prof_$c > $O/prof_$c.log 2>&1 || echo "profile $c failed" ;;
  esac
done
ls $O
