#!/bin/bash
# Start-up stagger of the FFT pass kernels (ira::stagger_start): block times for a few delays.  Tuning build only.
set -e
export IRA_TUNING=1 IRA_LIBRARY=audio_analysis_amd/csrc/libira_tuning.so
mkdir -p gpurun_out
out=gpurun_out/stagger_sweep.txt
: > $out
for cyc in 0 15000 30000 60000 120000 240000; do
  for blk in bands spectrum; do
    echo "== IRA_STAGGER_CYC=$cyc block=$blk" >> $out
    IRA_STAGGER_CYC=$cyc timeout -k 10 120 python3 tools/block_probe.py --block $blk --batch ${BATCH:-64} --iters 5 >> $out 2>&1
  done
done
tail -30 $out
