#!/bin/bash
# Headline bench (two lanes, H2D included) with the Bluestein column passes at 2 and at 4 columns per workgroup, alternating
# (tuning build: IRA_FFT_C sets both passes).  bash tools/experiments/r4_fft_c_bench.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r4_fft_c}; mkdir -p $O
export IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so
for rep in 1 2 3; do
  for c in 2 4; do
    IRA_FFT_C=$c timeout -k 10 200 python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --variants value --literal-steps 0 > $O/c${c}_$rep.json 2> $O/c${c}_$rep.err || echo "failed $c $rep"
    python3 -c "
import json,sys; d=json.load(open('$O/c${c}_$rep.json')); print('C', $c, 'rep', $rep, round(d['value'],1), 'IRs/s', round(d['ms_per_step'],2), 'ms; rfft_any', round(d['device_ms_per_step_by_call']['ira_rfft_any'],3))"
  done
done
