#!/bin/bash
# A/B of the float32 upload on ONE box: pieces per batch x copy-stream priority, alternating, headline +
# resident-input rate from the same process.   bash tools/experiments/r5_upload_ab.sh <outdir> [reps]
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r5_upload}; reps=${2:-2}; mkdir -p $O
for rep in $(seq $reps); do
  for arm in "2" "4" "8" "1" "2 --upload-priority" "4 --upload-priority" "2 --upload pull"; do
    tag=$(echo $arm | tr -d ' -'); f=$O/s${tag}_$rep
    timeout -k 10 300 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --variants resident --literal-steps 0 --roofline-steps 1 --upload-streams $arm > $f.json 2> $f.err || echo "failed $arm $rep"
    python3 - $f.json "$arm" $rep <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
u = d.get("upload_detail") or {}
print(f"streams {sys.argv[2]:22s} rep {sys.argv[3]}: value {d['value']:8.0f}  resident {d['value_resident']:8.0f}  ratio {d['value']/d['value_resident']:.3f}  "
      f"step {d['ms_per_step']:.2f} ms  upload {d.get('upload_ms_per_step') or 0:.2f} ms  under compute {d.get('h2d_under_compute_GBps') or 0:.1f} GB/s  alone {d.get('h2d_alone_GBps') or 0:.1f} GB/s  bound {d.get('bound')}")
PY
  done
done
