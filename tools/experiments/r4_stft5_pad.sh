#!/bin/bash
# Does the two-lane pipeline gain when the float64 STFT leaves LDS for the other lane's kernels?  IRA_STFT5_LDS_PAD (tuning
# build) adds unused dynamic LDS to every stft5 workgroup: 0 -> four workgroups per CU, 8192 -> three, 30000 -> two.
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r4_stft5_pad}; mkdir -p $O
export IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so
for rep in 1 2; do
  for pad in 0 8192 30000; do
    IRA_STFT5_LDS_PAD=$pad timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --variants value --literal-steps 0 > $O/p${pad}_$rep.json 2> $O/p${pad}_$rep.err || echo "failed $pad"
    python3 -c "
import json; d=json.load(open('$O/p${pad}_$rep.json')); print('pad $pad rep $rep:', round(d['value'],1), 'IRs/s', round(d['ms_per_step'],2), 'ms; stft_logbin alone', round(d['device_ms_per_step_by_call']['ira_stft_logbin[f64,n8192]'],3))"
  done
done
