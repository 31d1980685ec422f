#!/bin/bash
# Bluestein passes with radix-8 sub-FFT passes (fewer registers) AND smaller tiles (less LDS): more workgroups per CU?
# The tuning library must have been built with FL_LR = 3 and without the waves-per-SIMD attribute.
export IRA_TUNING=1 IRA_LIBRARY=audio_analysis_amd/csrc/libira_tuning.so
out=gpurun_out/bluestein_occ.txt; : > $out
for cr in "0 0" "2 2" "1 1" "2 1" "1 2" "4 2"; do
  set -- $cr
  echo "== IRA_FFT_C=$1 IRA_FFT_R=$2" >> $out
  IRA_FFT_C=$1 IRA_FFT_R=$2 timeout -k 10 120 python3 tools/block_probe.py --block spectrum --batch 64 --iters 5 2>&1 | grep block= >> $out
done
cat $out
