#!/bin/bash
# Kernel durations + SQ counters of one report block, with the derived utilisation figures.
#   bash tools/r2_block_profile.sh <block> [outdir]        (block as in tools/block_probe.py)
blk=$1; out=${2:-gpurun_out/r2_blockprof_$blk}
R=$GRAFT_REPO_ROOT; mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -- python3 $R/tools/block_probe.py --block $blk --iters 4 > $R/$out/stats.log 2>&1 || echo "stats failed" >> $R/$out/fail.log
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$out/p$i -- python3 $R/tools/block_probe.py --block $blk --iters 2 > $R/$out/p$i.log 2>&1 || echo "pmc $i failed" >> $R/$out/fail.log
done
python3 - $R/$out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = {}
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Name"][:60]] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]), float(r["Percentage"]))
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
with open(out + "/summary.txt", "w") as fh:
    for k, (us, calls, pct) in sorted(dur.items(), key=lambda kv: -kv[1][2])[:12]:
        d = {c: v / cnt[(k, c)] for c, v in acc.get(k, {}).items()}
        line = f"{k:60s} avg {us:9.1f} us  calls {calls:4d}  {pct:5.1f} %"
        if d.get("GRBM_GUI_ACTIVE"):
            simd_cycles = d["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0          # GRBM_GUI_ACTIVE is summed over the 8 XCDs
            line += (f" | VALU busy {4 * d.get('SQ_ACTIVE_INST_VALU', 0) / simd_cycles:5.1%}  LDS busy {4 * d.get('SQ_ACTIVE_INST_LDS', 0) / simd_cycles:5.1%}"
                     f"  VALU insts/wave {d.get('SQ_INSTS_VALU', 0) / max(d.get('SQ_WAVES', 1), 1):7.0f}  waves {d.get('SQ_WAVES', 0):9.0f}"
                     f"  wait {d.get('SQ_WAIT_ANY', 0) / max(d.get('SQ_WAIT_ANY', 0) + d.get('SQ_ACTIVE_INST_ANY', 1), 1):5.1%}"
                     f"  bank-conflict {d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_LDS_IDX_ACTIVE', 1), 1):5.1%}")
        fh.write(line + "\n")
print(open(out + "/summary.txt").read())
PY
