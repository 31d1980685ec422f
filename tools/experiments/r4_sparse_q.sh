#!/bin/bash
# Narrow-band path of ira_band_irfft_smooth: device time of the call in config 3 (26 third-octave bands, 256 x 10 s) as a
# function of the largest number of terms per cluster a job may have and still skip pass 1 (IRA_SPARSE_Q, tuning build; 0 = off).
# bash tools/experiments/r4_sparse_q.sh <outdir> [q ...]
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r4_sparse_q}; mkdir -p $O; shift
export IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so
for q in ${@:-0 1 2 4 6 9 12 14}; do
  export IRA_SPARSE_Q=$q
  timeout -k 10 200 python3 $R/bench.py --config 3 --steps 6 --warmup 2 --no-cpu-baseline --variants value > $O/q$q.json 2> $O/q$q.err || { echo "q $q failed" >> $O/summary.txt; break; }
  python3 - $O/q$q.json $q >> $O/summary.txt <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
c = d["device_ms_per_step_by_call"]
print(f"sparse_q {sys.argv[2]:>2}: ira_band_irfft_smooth {c['ira_band_irfft_smooth']:7.3f} ms per step   value {d['value']:8.1f} IRs/s")
PY
done
cat $O/summary.txt
