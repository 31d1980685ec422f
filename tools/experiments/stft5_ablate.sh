#!/bin/bash
# Timing-only ablations of the float64 modal STFT (tuning build): IRA_STFT5_ABLATE 1 no window loads, 2 no sample loads,
# 4 no dB -> float32 -> linear conversion.
export IRA_TUNING=1 IRA_LIBRARY=$GRAFT_REPO_ROOT/audio_analysis_amd/csrc/libira_tuning.so
for a in ${@:-0 1 2 3 4 7}; do echo -n "ablate $a: "; IRA_STFT5_ABLATE=$a timeout -k 10 100 python3 tools/block_probe.py --block modal 2>&1 | grep "block="; done
