#!/bin/bash
# Long-FFT chunking experiment: jobs per launch sized so that a chunk's working set stays in the 256 MB MALL.
# usage (GPU box): bash tools/experiments/sweep_workspace.sh  -> gpurun_out/ws_sweep.txt
mkdir -p gpurun_out
: > gpurun_out/ws_sweep.txt
for mb in 49152 1024 512 256 192 128 96 64; do
  IRA_WORKSPACE_MB=$mb python bench.py --steps 10 --warmup 2 --no-cpu-baseline --literal-steps 0 > gpurun_out/ws_$mb.json 2> gpurun_out/ws_$mb.err || exit 1
  python - "$mb" <<'PY' >> gpurun_out/ws_sweep.txt
import json, sys
mb = sys.argv[1]
d = json.load(open(f"gpurun_out/ws_{mb}.json"))
c = d["device_ms_per_step_by_call"]
print(mb, "MB:", round(d["value"]), "IRs/s", round(d["ms_per_step"], 2), "ms/step | band_irfft_smooth", round(c.get("ira_band_irfft_smooth", 0), 3),
      "rfft_any", round(c.get("ira_rfft_any", 0), 3), "rfft_smooth", round(c.get("ira_rfft_smooth", 0), 3), "| device", round(d["device_ms_per_step"], 2))
PY
done
cat gpurun_out/ws_sweep.txt
