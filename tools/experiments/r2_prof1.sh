#!/bin/bash
# round-2 profiling pass 1: kernel stats of single blocks + SQ counters of the float64 modal STFT
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r2_prof1; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for blk in decay bands3rd modal; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_$blk -- python3 $R/tools/block_probe.py --block $blk > $out/stats_$blk.log 2>&1 || echo "stats $blk failed" >> $out/fail.log
done
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_modal_$i -- python3 $R/tools/block_probe.py --block modal --iters 2 > $out/pmc_modal_$i.log 2>&1 || echo "pmc modal $i failed" >> $out/fail.log
done
python3 - $out <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
with open(out + "/summary.txt", "w") as fh:
    for f in sorted(glob.glob(out + "/stats_*/**/*kernel_stats.csv", recursive=True)):
        fh.write("== " + f.split("/r2_prof1/")[1].split("/")[0] + "\n")
        for r in list(csv.DictReader(open(f)))[:14]:
            fh.write(f"  {r['Name'][:70]:70s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:10.1f} total_ms {float(r['TotalDurationNs'])/1e6:9.3f} pct {r['Percentage']}\n")
    for f in sorted(glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:50]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            if "stft" in k:
                fh.write(f"{f.split('/r2_prof1/')[1].split('/')[0]} {k} " + str({c: v / cnt[(k, c)] for c, v in d.items()}) + "\n")
print(open(out + "/summary.txt").read())
PY
