#!/bin/bash
# On the GPU box, from the repo root: build + count + run the dft16 issue-rate microbenchmark, then the real kernel
# (frame-major float32 STFT) whole and with its memory ablated (tuning build) for the same wave-instr / CU-cycle figure.
#     bash tools/micro/run_dft16.sh [outfile]
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=${1:-$R/gpurun_out/dft16_rate.txt}
mkdir -p "$(dirname "$out")"
FLAGS="--offload-arch=gfx950 -O3 -fno-slp-vectorize -I $R/audio_analysis_amd/csrc"
hipcc $FLAGS --cuda-device-only -S $R/tools/micro/dft16_rate.hip -o /tmp/dft16_rate.s 2> /dev/null
python3 $R/tools/micro/dft16_count.py /tmp/dft16_rate.s > /tmp/dft16_counts.txt 2> /tmp/dft16_mix.txt
hipcc $FLAGS $R/tools/micro/dft16_rate.hip -o /tmp/dft16_rate 2> /dev/null
{
  echo "== VALU instructions per loop iteration (device assembly)"; cat /tmp/dft16_mix.txt
  echo "== issue rates"; /tmp/dft16_rate /tmp/dft16_counts.txt
  echo "== the kernel itself: ira_stft_mag_db_tf, 64 x 10 s (59 392 frames; 2083 VALU instructions per frame-wave measured, SQ_INSTS_VALU)"
  python3 $R/tools/stft_probe.py --tf --iters 10
  if [ -f $R/audio_analysis_amd/csrc/libira_tuning.so ]; then
    for ab in 1 3 7 4; do
      echo "-- IRA_STFT3_ABLATE=$ab (1 no sample loads, 2 no window reads, 4 one store per workgroup)"
      IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so IRA_STFT3_ABLATE=$ab python3 $R/tools/stft_probe.py --tf --iters 10
    done
  fi
} > "$out" 2>&1
cat "$out"
