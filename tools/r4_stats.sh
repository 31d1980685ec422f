#!/bin/bash
# Kernel-trace statistics of the serialised report step (one stream) at HEAD: bash tools/r4_stats.sh <outdir> [config] [extra bench args]
R=$GRAFT_REPO_ROOT; out=${1:-gpurun_out/r4_stats}; cfg=${2:-report}; shift 2
mkdir -p "$R/$out"
cd /tmp && export TMPDIR=/tmp
export IRA_STREAMS=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/stats" -- python3 "$R/bench.py" --config $cfg --steps 8 --warmup 1 --host-batches 4 --variants value --no-cpu-baseline --literal-steps 0 --roofline-steps 1 --upload copy "$@" > "$R/$out/stats.log" 2>&1 || echo "stats pass failed" >> "$R/$out/fail.log"
f=$(find "$R/$out/stats" -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:45]:
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{nm[:64]:64s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e6:8.4f} ms total {float(r['TotalDurationNs'])/1e6:8.3f} {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
cp "$f" "$R/$out/kernel_stats.csv"
