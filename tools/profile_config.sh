#!/bin/bash
# Profile of one bench configuration (on the GPU box, from the repo root):
#     bash tools/profile_config.sh <config: report|2|3|4> [outdir]
# Passes (each its own run; --pmc never combined with other trace domains; the program directly after --):
#   stats        rocprofv3 --kernel-trace --stats             report blocks on ONE stream (kernels one at a time)
#   FETCH_SIZE   WRITE_SIZE   TCC_EA0_RDREQ_sum+TCC_EA0_RDREQ_32B_sum   counters per dispatch
# then tools/traffic_profile.py reduces them to per-kernel and per-ABI-call tables (profiles/rNN_*).
cfg=${1:-report}; out=${2:-gpurun_out/prof_r2_$cfg}
R=$GRAFT_REPO_ROOT
mkdir -p "$R/$out"
cd /tmp && export TMPDIR=/tmp
export IRA_STREAMS=1
B=${3:-}; ARGS="--config $cfg ${B:+--batch $B} --steps 2 --warmup 1 --host-batches 2 --variants value --no-cpu-baseline --literal-steps 0 --roofline-steps 1 --upload copy"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$out/stats" -- python3 "$R/bench.py" $ARGS > "$R/$out/stats.log" 2>&1 || echo "stats pass failed" >> "$R/$out/fail.log"
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  tag=$(echo $c | cut -d' ' -f1)
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$R/$out/$tag" -- python3 "$R/bench.py" $ARGS > "$R/$out/$tag.log" 2>&1 || echo "$tag pass failed" >> "$R/$out/fail.log"
done
python3 "$R/tools/traffic_profile.py" "$R/$out" $cfg
