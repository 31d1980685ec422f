#!/bin/bash
# Narrow-band kernel: what its time is made of (timing-only ablations of the tuning build; results are wrong by construction).
# bash tools/experiments/r4_sparse_ablate.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r4_sparse_abl}; mkdir -p $O
export IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so IRA_STREAMS=1
cd /tmp && export TMPDIR=/tmp
for a in 0 256 512 768 1024 2048 3072 3840; do
  export IRA_SMOOTH_ABLATE=$a
  rm -rf $O/st$a
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/st$a -- python3 $R/bench.py --config 3 --steps 4 --warmup 1 --host-batches 2 --variants value --no-cpu-baseline --literal-steps 0 --roofline-steps 1 --upload copy > $O/a$a.log 2>&1 || { echo "ablate $a failed" >> $O/summary.txt; break; }
  f=$(find $O/st$a -name '*kernel_stats.csv' | head -1)
  python3 - $f $a >> $O/summary.txt <<'PY'
import csv, sys
rows = {r["Name"]: r for r in csv.DictReader(open(sys.argv[1]))}
def avg(sub):
    for k, r in rows.items():
        if sub in k: return float(r["AverageNs"]) / 1e6
    return float("nan")
print(f"ablate {int(sys.argv[2]):5d}: sparse rows {avg('smooth_rows_sparse'):7.3f}  regular rows {avg('smooth_rows_kernel<1>'):7.3f}  cols {avg('smooth_cols_kernel<1, false>'):7.3f}  compact {avg('band_compact'):6.3f} ms")
PY
done
cat $O/summary.txt
