#!/usr/bin/env python3
"""Times the RT60 filter bank transforms on the bench batch (forward paired rfft + 96 band inverses) for tuning."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audio_analysis_amd.engine import Engine
from audio_analysis_amd.analyse import rt60bands as rb
from audio_analysis_amd.synth import synth_ir
B = 64; n = 480000
eng = Engine("cuda:0")
host = np.stack([synth_ir(i, 0, n) for i in range(B)])
batch = eng.wrap(eng.to_dev(host.reshape(-1)), np.arange(B, dtype=np.int64) * n, np.full(B, n, np.int64))
eng.peaks(batch)
st = rb.Rt60BandsAnalysisSettings()
for _ in range(2):
    rb.rt60_bands_device(eng, batch, 48000, st)
torch.cuda.synchronize(); eng.events = []
for _ in range(3):
    rb.rt60_bands_device(eng, batch, 48000, st)
ev = eng.collect_events(); eng.events = None
tot = {k: sum(v) / 3 for k, v in ev.items()}
print("split", eng.smooth_split(n), " ".join(f"{k} {v:.3f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1]) if "fft" in k))
