#!/bin/bash
# float32 STFT, persistent kernel: time and L2->fabric fetch traffic as a function of the waves' re-sync interval
# (IRA_STFT6_RESYNC, tuning build).  bash tools/experiments/r4_stft_resync.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r4_resync}; mkdir -p $O
export IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so
for rs in 0 1 2 4 8 16; do
  export IRA_STFT6_RESYNC=$rs
  echo "== resync $rs" >> $O/summary.txt
  python3 $R/tools/stft_probe.py --batch 256 --tf --iters 10 2>/dev/null | tail -1 >> $O/summary.txt
  PMC_GROUPS="FETCH_SIZE;WRITE_SIZE" PMC_PROBE_ARGS="--batch 256 --tf --iters 3" bash $R/tools/pmc_stft.sh ${1:-gpurun_out/r4_resync}/pmc_$rs 2>/dev/null | grep stft6 >> $O/summary.txt
done
cat $O/summary.txt
