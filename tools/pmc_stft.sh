#!/bin/bash
# Counter passes over the STFT probe (one --pmc group per run; kernel-trace only).
# Usage: [PMC_GROUPS="A B;C D"] tools/pmc_stft.sh <outdir> [probe args]   (environment, e.g. IRA_STFT3_ABLATE, reaches the probe)
out=$1; shift
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > "$R/$out/counters.txt" 2>&1
i=0
DEFAULT_GROUPS="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU;SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS;SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE;TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum;TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
IFS=';' read -ra GROUPS_ARR <<< "${PMC_GROUPS:-$DEFAULT_GROUPS}"
for grp in "${GROUPS_ARR[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d "$R/$out/p$i" -- python3 "$R/tools/${PMC_PROBE:-stft_probe.py}" ${PMC_PROBE_ARGS:---iters 3} "$@" > "$R/$out/p$i.log" 2>&1 || echo "pass $i failed" >> "$R/$out/fail.log"
done
python3 - "$R/$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[(k, r["Counter_Name"])] += 1
    for k, d in acc.items():
        if any(t in k for t in ("stft", "cols_", "rows_", "edc", "curve", "gram")):
            print(f.split("/")[-3] if "/p" in f else f, k, {c: v / cnt[(k, c)] for c, v in d.items()})
PY
