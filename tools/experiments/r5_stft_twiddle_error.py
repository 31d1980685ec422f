#!/usr/bin/env python3
"""Is the float32 spectrogram's worst-bin error set by the twiddle product tree or by the transform?  Error of the float32 STFT
against the float64 STFT of the same samples on the bins SURVEY 8d names (float64 value > floor + 20 dB), product kernel vs the
tuning build's exact-twiddle variant (IRA_STFT6_ABLATE=128: every twiddle read from the float32 table).
    IRA_TUNING=1 IRA_LIBRARY=.../libira_tuning.so [IRA_STFT6_ABLATE=128] python3 tools/experiments/r5_stft_twiddle_error.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from audio_analysis_amd.engine import Engine
from audio_analysis_amd.synth import synth_ir
eng = Engine("cuda:0")
n, B = 480000, 16
b = eng.upload([synth_ir(i, 0, n) for i in range(B)])
start = np.array([240 + (i % 512) for i in range(B)], dtype=np.int64)
nfr = (1 + (n - start - 4096) // 512).astype(np.int32)
a, a_off, cols = eng.stft_mag_db(b.x, b.off + start, nfr, 4096, 512, True, -120.0, 32, frame_major=True)
r, r_off, _ = eng.stft_mag_db(b.x, b.off + start, nfr, 4096, 512, True, -120.0, 64)
eng.sync()
worst, n3, n4, tot, sq = 0.0, 0, 0, 0, 0.0
for i in range(B):
    T = int(cols[i])
    ai = a[int(a_off[i]) : int(a_off[i]) + 2049 * T].view(T, 2049).t().double()
    ri = r[int(r_off[i]) : int(r_off[i]) + 2049 * T].view(2049, T).double()
    m = ri > -100.0
    e = (ai - ri).abs()[m]
    worst = max(worst, float(e.max())); tot += e.numel(); n3 += int((e <= 1e-3).sum()); sq += float((e * e).sum())
    n4 += int((e <= 1e-4).sum())
print(f"ablate={os.environ.get('IRA_STFT6_ABLATE', '0')}: {tot} bins, max |err| {worst:.3e} dB, rms {np.sqrt(sq / tot):.3e} dB, within 1e-3 dB {n3 / tot:.7f}, within 1e-4 dB {n4 / tot:.5f}")
