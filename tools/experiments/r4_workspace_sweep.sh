#!/bin/bash
# Do the Bluestein passes gain when the work arrays of the jobs in flight fit the 256 MB Infinity Cache?  IRA_FFT_CHUNK (tuning
# build) runs the three passes over sub-ranges of that many jobs inside one library call (no host work per sub-range).
#   bash tools/experiments/r4_workspace_sweep.sh [batch]
export IRA_TUNING=1 IRA_LIBRARY=$GRAFT_REPO_ROOT/audio_analysis_amd/csrc/libira_tuning.so
B=${1:-256}
timeout -k 10 100 python3 tools/fft_probe.py $B > /dev/null 2>&1
for ch in 0 64 32 16 8 4 0 16; do
  echo -n "jobs per sub-range $ch: "; IRA_FFT_CHUNK=$ch timeout -k 10 100 python3 tools/fft_probe.py $B 2>&1 | grep rfft_any
done
