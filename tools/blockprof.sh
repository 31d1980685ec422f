#!/bin/bash
# Kernel durations + SQ counters of ONE report block (tools/block_probe.py), with derived utilisation figures.  One runner for
# what tools/blockprof.sh, r2_pmc_block.sh and r3_block_counters.sh did; environment assignments select the build/knobs.
#   bash tools/blockprof.sh <block> <outdir> [--batch N] [VAR=value ...]      e.g.
#   bash tools/blockprof.sh spectrum gpurun_out/r5_k3g/glds1 --batch 256 IRA_TUNING=1 IRA_LIBRARY=$PWD/audio_analysis_amd/csrc/libira_tuning.so IRA_FFT_GLDS=1
# Passes: rocprofv3 --kernel-trace --stats, then three --pmc groups, each its own run (never combined with other trace domains).
blk=$1; out=$2; shift 2
batch=64; if [ "$1" = "--batch" ]; then batch=$2; shift 2; fi
R=$GRAFT_REPO_ROOT; mkdir -p $R/$out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$out/stats -- python3 $R/tools/block_probe.py --block $blk --batch $batch --iters 4 > $R/$out/stats.log 2>&1 || echo "stats failed" >> $R/$out/fail.log
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$out/p$i -- python3 $R/tools/block_probe.py --block $blk --batch $batch --iters 2 > $R/$out/p$i.log 2>&1 || echo "pmc $i failed" >> $R/$out/fail.log
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
            wc = max(d.get("SQ_WAVE_CYCLES", 0), 1)
            line += (f" | VALU busy {4 * d.get('SQ_ACTIVE_INST_VALU', 0) / simd_cycles:5.1%}  LDS busy {4 * d.get('SQ_ACTIVE_INST_LDS', 0) / simd_cycles:5.1%}"
                     f"  VALU insts/wave {d.get('SQ_INSTS_VALU', 0) / max(d.get('SQ_WAVES', 1), 1):7.0f}  waves {d.get('SQ_WAVES', 0):9.0f}"
                     f"  wave-cycles waiting (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) {d.get('SQ_WAIT_INST_ANY', 0) / wc:5.1%}"
                     f"  resident waves per SIMD {4 * d.get('SQ_WAVE_CYCLES', 0) / simd_cycles:4.2f}"
                     f"  bank-conflict {d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d.get('SQ_LDS_IDX_ACTIVE', 1), 1):5.1%}")
        fh.write(line + "\n")
print(open(out + "/summary.txt").read())
PY
