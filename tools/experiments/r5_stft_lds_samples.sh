#!/bin/bash
# float32 STFT: what would a tile's sample span staged once in LDS buy?  Timing-only emulation (tuning build, IRA_STFT6_ABLATE=64:
# the frame's 32 global sample loads become 8-byte LDS reads, plus three 16-byte global loads + LDS writes per frame-wave as the
# wave's share of the staging), at sixteen and at twelve waves per CU (IRA_STFT6_VARIANT=2: the LDS the span would need).
R=$GRAFT_REPO_ROOT
export IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so
for rep in 1 2 3; do
  for arm in "IRA_STFT6_ABLATE=0" "IRA_STFT6_ABLATE=64" "IRA_STFT6_ABLATE=68" "IRA_STFT6_VARIANT=2 IRA_STFT6_ABLATE=0" "IRA_STFT6_VARIANT=2 IRA_STFT6_ABLATE=64"; do
    echo -n "$arm: "; env $arm python3 $R/tools/stft_probe.py --tf --batch 256 --iters 30 2>/dev/null | tail -1
  done
done
