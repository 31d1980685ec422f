#!/bin/bash
# One call = everything profiles/ needs for a build: kernel-trace stats, the two HBM-traffic counter passes (each
# --pmc group in its own run, kernel-trace only, program directly after --), and their per-call reduction.
# Usage (on the GPU box, from the repo root):  bash tools/profile_round.sh gpurun_out/prof_tag [batch]
out=$1; batch=${2:-64}
R=$GRAFT_REPO_ROOT
mkdir -p "$R/$out"
cd /tmp && export TMPDIR=/tmp
# Per-kernel durations and counters are taken with the report blocks on ONE stream (kernels one at a time, dispatch order
# = call order, which tools/traffic_from_pmc.py relies on); a second stats pass records the default multi-stream run.
export IRA_STREAMS=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/stats" -- python3 "$R/bench.py" --steps 3 --warmup 1 --batch $batch --no-cpu-baseline --literal-steps 0 > "$R/$out/stats.log" 2>&1 || echo "stats pass failed" >> "$R/$out/fail.log"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/$out/$c" -- python3 "$R/bench.py" --steps 2 --warmup 1 --batch $batch --no-cpu-baseline --literal-steps 0 > "$R/$out/$c.log" 2>&1 || echo "$c pass failed" >> "$R/$out/fail.log"
done
python3 "$R/tools/traffic_from_pmc.py" "$R/$out" $batch
unset IRA_STREAMS
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/stats_lanes" -- python3 "$R/bench.py" --steps 3 --warmup 1 --batch $batch --no-cpu-baseline --literal-steps 0 --roofline-steps 1 > "$R/$out/stats_lanes.log" 2>&1 || echo "lanes stats pass failed" >> "$R/$out/fail.log"
