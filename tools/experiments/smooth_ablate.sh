#!/bin/bash
# Timing-only ablations of the smooth-FFT passes (tuning build; results are wrong by construction): IRA_SMOOTH_ABLATE bits, see SmoothPlan.
export IRA_TUNING=1 IRA_LIBRARY=$GRAFT_REPO_ROOT/audio_analysis_amd/csrc/libira_tuning.so
for a in ${@:-0 1 7 8 16 24 32 56 63}; do echo "ablate $a"; IRA_SMOOTH_ABLATE=$a timeout -k 10 100 python3 tools/smooth_probe.py 2>&1 | grep split; done
