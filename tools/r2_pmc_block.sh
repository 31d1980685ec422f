#!/bin/bash
# SQ counters of one report block: bash tools/r2_pmc_block.sh <block> <kernel substring> [outdir]
blk=$1; pat=$2; out=${3:-gpurun_out/r2_pmc_$blk}
R=$GRAFT_REPO_ROOT; mkdir -p $R/$out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/$out/p$i -- python3 $R/tools/block_probe.py --block $blk --iters 2 > $R/$out/p$i.log 2>&1 || echo "pmc $i failed" >> $R/$out/fail.log
done
python3 - $R/$out "$pat" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
with open(out + "/summary.txt", "w") as fh:
    for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            if pat in k:
                fh.write(f"{k} " + str({c: round(v / cnt[(k, c)], 1) for c, v in d.items()}) + "\n")
print(open(out + "/summary.txt").read())
PY
