#!/usr/bin/env python3
"""Times the arbitrary-length float64 rFFT path alone (ira_rfft_any; chirp filters cached)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audio_analysis_amd.engine import Engine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine("cuda:0"); n = 480000
rng = np.random.default_rng(0)
x = (rng.standard_normal((B, n)) * np.exp(-np.arange(n) / 48000.0)).astype(np.float32)
b = eng.wrap(eng.to_dev(x.reshape(-1)), np.arange(B, dtype=np.int64) * n, np.full(B, n, np.int64))
lens = (n - 240 - (np.arange(B) % 512)).astype(np.int64)
for _ in range(2):
    eng.rfft_any(b.x, b.off + 240, lens, True)
torch.cuda.synchronize(); eng.events = []
t0 = time.perf_counter()
for _ in range(5):
    eng.rfft_any(b.x, b.off + 240, lens, True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"rfft_any B={B} L~{int(lens.mean())}: {dt*1e3:.3f} ms per call, {dt*1e6/B:.1f} us per transform")
