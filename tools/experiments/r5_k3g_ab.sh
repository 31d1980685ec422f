#!/bin/bash
# A/B on one box: Bluestein K3 by LDS-DMA (cols_inv_glds_kernel, IRA_FFT_GLDS=1) against the register-staged kernel
# (IRA_FFT_GLDS=0), tuning build, kernel-trace statistics of tools/fft_probe.py (256 fr/filter spectra of ~10 s).
#   bash tools/experiments/r5_k3g_ab.sh <outdir> [extra env assignments for the glds arm, e.g. IRA_FFT_GLDS_WG=2]
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r5_k3g}; shift; mkdir -p $O
export IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
  for arm in 0 1; do
    d=$O/glds${arm}_$rep
    env IRA_FFT_GLDS=$arm "$@" timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/fft_probe.py 256 > $d.log 2>&1 || echo "arm $arm rep $rep failed" >> $O/fail.log
    f=$(find $d -name '*kernel_stats.csv' | head -1)
    echo "== glds=$arm rep $rep $*"; grep rfft_any $d.log
    python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:6]:
    nm = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"  {nm[:60]:60s} calls {int(r['Calls']):4d} avg {float(r['AverageNs'])/1e6:8.4f} ms")
PY
  done
done
