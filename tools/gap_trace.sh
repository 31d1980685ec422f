#!/bin/bash
# GPU idle time between operations of the steady-state step: kernel trace + memory-copy trace (no counters), then
# tools/gap_analysis.py.  Usage (GPU box):  bash tools/gap_trace.sh gpurun_out/gaps
out=$1
R=$GRAFT_REPO_ROOT
mkdir -p "$R/$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$R/$out/trace" -- python3 "$R/bench.py" --steps 6 --warmup 2 --no-cpu-baseline --literal-steps 0 > "$R/$out/trace.log" 2>&1 || echo "trace failed" >> "$R/$out/fail.log"
python3 "$R/tools/gap_analysis.py" "$R/$out" > "$R/$out/gaps.txt" 2>&1
tail -40 "$R/$out/gaps.txt"
