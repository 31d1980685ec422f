#!/bin/bash
# A/B of two builds of the library on ONE box: kernel statistics of one report block (tools/block_probe.py under rocprofv3
# --kernel-trace --stats), alternating.  bash tools/lib_ab.sh <block> <other.so (repo-relative)> [batch] [reps] [grep pattern]
R=$GRAFT_REPO_ROOT; blk=$1; other=$2; batch=${3:-256}; reps=${4:-2}; pat=${5:-kernel}
cd /tmp && export TMPDIR=/tmp
one() {  # label, env...
  local d=$R/gpurun_out/lib_ab/$1; shift; rm -rf $d; mkdir -p $d
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/block_probe.py --block $blk --batch $batch --iters 6 > $d.log 2>&1 || echo "failed"
  python3 - "$(find $d -name '*kernel_stats.csv' | head -1)" "$pat" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:8]:
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    if sys.argv[2] in nm: print(f"    {nm[:70]:70s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e6:8.4f} ms")
PY
}
for rep in $(seq $reps); do
  echo "== this build, rep $rep"; one new$rep X=0
  echo "== $other, rep $rep"; one old$rep IRA_TUNING=1 IRA_LIBRARY=$R/$other
done
