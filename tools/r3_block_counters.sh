#!/bin/bash
# Kernel durations + SQ counters per report block at HEAD (tools/r2_block_profile.sh for every block), one summary file.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3_blocks; mkdir -p $O
for b in bands spectrum modal stft zplane decay; do
  bash $R/tools/r2_block_profile.sh $b gpurun_out/r3_blocks/$b > $O/$b.log 2>&1 || echo "$b failed" >> $O/fail.log
  echo "== block $b (64 x 10 s)" >> $O/summary_all.txt
  cat $O/$b/summary.txt >> $O/summary_all.txt
done
cat $O/summary_all.txt | cut -c1-200
