#!/bin/bash
# Columns per workgroup of the smooth-FFT passes (IRA_SMOOTH_C1 / C2, tuning build): block times.
export IRA_TUNING=1 IRA_LIBRARY=audio_analysis_amd/csrc/libira_tuning.so
out=gpurun_out/smooth_c_sweep.txt; : > $out
for c in "2 2" "3 2" "5 2" "6 2" "2 4" "2 5" "3 4" "5 4" "5 5" "6 5" "1 1"; do
  set -- $c
  echo "== C1=$1 C2=$2" >> $out
  IRA_SMOOTH_C1=$1 IRA_SMOOTH_C2=$2 timeout -k 10 120 python3 tools/block_probe.py --block bands --batch 64 --iters 5 2>&1 | grep block= >> $out
done
cat $out
