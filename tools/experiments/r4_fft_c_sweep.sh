#!/bin/bash
# Bluestein passes: tile widths of the forward (IRA_FFT_C) and inverse (IRA_FFT_C3) column passes chosen separately, rows per
# workgroup of the row pass (IRA_FFT_R; 0 = the plan's choice) -- tuning build.
#   bash tools/experiments/r4_fft_c_sweep.sh [batch] [list of C:C3:R ...]
export IRA_TUNING=1 IRA_LIBRARY=$GRAFT_REPO_ROOT/audio_analysis_amd/csrc/libira_tuning.so
B=${1:-256}; shift
LIST=${@:-2:2:0 4:4:0 2:2:0 4:2:0 4:4:0 4:4:2 4:4:8 2:2:2 2:2:8 2:2:0 4:4:0}
timeout -k 10 100 python3 tools/fft_probe.py $B > /dev/null 2>&1     # warm the box up
for cc in $LIST; do
  IFS=: read c c3 r <<< "$cc"
  echo -n "C $c C3 $c3 R $r: "; IRA_FFT_C=$c IRA_FFT_C3=$c3 IRA_FFT_R=$r timeout -k 10 100 python3 tools/fft_probe.py $B 2>&1 | grep rfft_any
done
