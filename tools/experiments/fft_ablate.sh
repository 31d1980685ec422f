#!/bin/bash
# Timing-only ablations of the Bluestein passes (tuning build; results are wrong by construction): IRA_FFT_ABLATE bits, see Geom
# in ira_fftlong.hip: 1 K1 plain input, 2 K1 no FFT, 4 K1 no store, 8 K2 no FFTs, 16 K2 no filter multiply, 32 K3 no FFT,
# 64 K3 plain epilogue, 128 K2 no store, 256 K3 no load.
export IRA_TUNING=1 IRA_LIBRARY=$GRAFT_REPO_ROOT/audio_analysis_amd/csrc/libira_tuning.so
for a in ${@:-0 1 2 3 8 16 24 32 64 96 42 107 123}; do echo -n "ablate $a: "; IRA_FFT_ABLATE=$a timeout -k 10 100 python3 tools/fft_probe.py 64 2>&1 | grep rfft_any; done
