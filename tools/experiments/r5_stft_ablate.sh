#!/bin/bash
# float32 STFT (stft6_kernel), two tuning builds alternating on one box: whole kernel and with memory ablated
# (IRA_STFT6_ABLATE: 1 no sample loads, 4 no stores, 7 no loads / window reads / stores; +16 polynomial logarithm).
#   bash tools/experiments/r5_stft_ablate.sh <other tuning .so> ["ablate values of this build"] ["ablate values of the other"]
R=$GRAFT_REPO_ROOT; other=$1; mine=${2:-"0 16 7 23"}; theirs=${3:-"0 7"}
for rep in 1 2 3; do
  for ab in $mine; do echo -n "this build  ablate $ab: "; IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so IRA_STFT6_ABLATE=$ab python3 $R/tools/stft_probe.py --tf --batch 256 --iters 30 2>/dev/null | tail -1; done
  for ab in $theirs; do echo -n "$(basename $other) ablate $ab: "; IRA_TUNING=1 IRA_LIBRARY=$R/$other IRA_STFT6_ABLATE=$ab python3 $R/tools/stft_probe.py --tf --batch 256 --iters 30 2>/dev/null | tail -1; done
done
