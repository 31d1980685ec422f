#!/bin/bash
# Timing-only ablations of the LDS-DMA K3 (tuning build): IRA_FFT_ABLATE bits 32 no FFT, 64 plain epilogue, 256 no load,
# 512 no store; IRA_FFT_GLDS_WG = resident workgroups per CU.  Prints ms per rfft_any call of 256 spectra (K1 + K2 + K3).
R=$GRAFT_REPO_ROOT
export IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so
for arm in "IRA_FFT_GLDS=0" "IRA_FFT_GLDS=0 IRA_FFT_ABLATE=32" "IRA_FFT_GLDS=0 IRA_FFT_ABLATE=256" "IRA_FFT_GLDS=0 IRA_FFT_ABLATE=352" \
           "IRA_FFT_GLDS=1" "IRA_FFT_GLDS=1 IRA_FFT_GLDS_WG=1" "IRA_FFT_GLDS=1 IRA_FFT_GLDS_WG=2" \
           "IRA_FFT_GLDS=1 IRA_FFT_ABLATE=32" "IRA_FFT_GLDS=1 IRA_FFT_ABLATE=64" "IRA_FFT_GLDS=1 IRA_FFT_ABLATE=96" "IRA_FFT_GLDS=1 IRA_FFT_ABLATE=256" \
           "IRA_FFT_GLDS=1 IRA_FFT_ABLATE=512" "IRA_FFT_GLDS=1 IRA_FFT_ABLATE=768" "IRA_FFT_GLDS=1 IRA_FFT_ABLATE=864"; do
  echo -n "$arm: "; env $arm timeout -k 10 100 python3 $R/tools/fft_probe.py 256 2>&1 | grep rfft_any
done
