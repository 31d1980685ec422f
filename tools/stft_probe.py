#!/usr/bin/env python3
"""Times the spectrogram STFT kernel alone (B channels x S seconds) -- used under rocprofv3 --pmc."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audio_analysis_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--nfft", type=int, default=4096); ap.add_argument("--precision", type=int, default=32)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--tf", action="store_true", help="frame-major (T, F) output: what the metrics pipeline asks for")
a = ap.parse_args()
eng = Engine("cuda:0")
n = int(a.seconds * 48000)
rng = np.random.default_rng(0)
x = (rng.standard_normal((a.batch, n)) * np.exp(-np.arange(n) / 48000.0)).astype(np.float32)
b = eng.wrap(eng.to_dev(x.reshape(-1)), np.arange(a.batch, dtype=np.int64) * n, np.full(a.batch, n, np.int64))
starts = np.full(a.batch, 300, dtype=np.int64)
nfr = np.full(a.batch, 1 + (n - 300 - a.nfft) // 512, dtype=np.int32)
for _ in range(2):
    eng.stft_mag_db(b.x, b.off + starts, nfr, a.nfft, 512, True, -120.0, a.precision, frame_major=a.tf)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    out, off, cols = eng.stft_mag_db(b.x, b.off + starts, nfr, a.nfft, 512, True, -120.0, a.precision, frame_major=a.tf)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
L = n - 300
bytes_ = a.batch * (4.0 * L + 4.0 * (a.nfft // 2 + 1) * nfr[0])
if os.environ.get("IRA_STFT2_ABLATE") == "256":
    print("stamps [step1, step2, step3, post, copy/keep, barrier, tile+store] cycles:", out[:7].cpu().numpy().tolist(), " step2(h=1) [16 LDS reads, dft16, twiddle+16 LDS writes]:", out[8:11].cpu().numpy().tolist())
print(f"variant={'generic' if os.environ.get('IRA_STFT_GENERIC') else 'v2'} f{a.precision} nfft={a.nfft} "
      f"B={a.batch}: {dt*1e3:.3f} ms/launch, {bytes_/dt/1e9:.1f} GB/s algorithmic, "
      f"{dt*1e9/(a.batch*nfr[0]):.1f} ns/frame")
