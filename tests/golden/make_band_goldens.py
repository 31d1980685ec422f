#!/usr/bin/env python3
"""
Band-signal and mask goldens (a8 / a9 of SURVEY.md section 8a) from the REFERENCE implementation, run in the build container:

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 tests/golden/make_band_goldens.py

For two inputs of goldens.npz -- xd (n = 23257 = 13 * 1789: Bluestein) and xb (n = 48000: the direct smooth transform) --
and one low, one mid and one high band of each band mode's kind (low-pass <= 250 Hz, band-pass 500-2000 Hz, high-pass
>= 4000 Hz, plus a third-octave band-pass at 1 kHz) it stores what the reference's own functions return:
  mask/<input>/<band>   _make_*_mask on rfftfreq(n).astype(float32)          (analyse/rt60bands.py:127-167)
  y/<input>/<band>      _apply_fft_mask(x, mask) = irfft(rfft(x) * mask, n)   (analyse/rt60bands.py:170-175), float32
Only inputs' names and expected outputs are stored (tests/golden/band_signals.npz); no reference source text.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path[:] = [p for p in sys.path if p.rstrip("/") != "/root/reference"]
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

import analyse.rt60bands as rbands  # noqa: E402

SR = 48000
NYQ = 24000.0
TW = 1.0 / 6.0


def main():
    g = np.load(HERE / "goldens.npz")
    out = {}
    step = 2.0 ** (1.0 / 3.0)
    bands = {
        "lp250": lambda f: rbands._make_lowpass_mask(f, 250.0, TW, NYQ),
        "bp500_2000": lambda f: rbands._make_bandpass_mask(f, 500.0, 2000.0, TW, NYQ),
        "hp4000": lambda f: rbands._make_highpass_mask(f, 4000.0, TW, NYQ),
        # the third-octave band centred on 1 kHz: edges fc * 2^(-+1/6) as _build_fractional_octave_band_definitions forms them
        "third1k": lambda f: rbands._make_bandpass_mask(f, 1000.0 / (step ** 0.5), 1000.0 * (step ** 0.5), TW, NYQ),
    }
    for tag in ("xd", "xb"):
        x = g[f"in/{tag}"]
        n = int(x.size)
        freqs = np.fft.rfftfreq(n, d=1.0 / float(SR)).astype(np.float32)
        for name, make in bands.items():
            m = make(freqs)
            y = rbands._apply_fft_mask(x, m)
            assert m.dtype == np.float32 and y.dtype == np.float32 and y.size == n
            out[f"mask/{tag}/{name}"] = m
            out[f"y/{tag}/{name}"] = y
    out["third1k_edges"] = np.array([1000.0 / (step ** 0.5), 1000.0 * (step ** 0.5)])
    np.savez_compressed(HERE / "band_signals.npz", **out)
    print("wrote", HERE / "band_signals.npz", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
