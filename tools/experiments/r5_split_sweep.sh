R=$GRAFT_REPO_ROOT
export IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so IRA_FFT_GLDS=0
for rep in 1 2; do
for arm in "X=0" "IRA_FFT_SPLIT=9" "IRA_FFT_SPLIT3=7" "IRA_FFT_SPLIT=9 IRA_FFT_SPLIT3=7" "IRA_FFT_SPLIT=9 IRA_FFT_SPLIT3=7 IRA_FFT_R=4" "IRA_FFT_SPLIT=9 IRA_FFT_C=8 IRA_FFT_SPLIT3=7 IRA_FFT_C3=8" "IRA_FFT_SPLIT=8"; do
  echo -n "$arm: "; env $arm timeout -k 10 100 python3 $R/tools/fft_probe.py 256 2>&1 | grep rfft_any
done; done
