#!/usr/bin/env python3
"""
Generate tests/golden/ar_illcond.npz: the REFERENCE's covariance-method AR fit (analyse/zplane.py:83-120, numpy.linalg.lstsq
= SVD) and pole finder (:145-158) on the ill-conditioned inputs SURVEY.md section 7 (hard part 1) names -- a float32 impulse
response low-passed at 500 Hz, orders 64 and 256 (cond(A^T A) >= 1e13) -- plus a milder 2 kHz case.  Run in the build container:

    python3 tests/golden/make_ar_illcond_goldens.py

Committed output (data only): per case the float32 input segment, the order, the reference's coefficient vector and poles,
the singular-value extremes of the design matrix and the rank lstsq reports.
"""
from __future__ import annotations

import os
import sys
from pathlib import Path

import numpy as np
from scipy.signal import butter, lfilter

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path[:] = [p for p in sys.path if p.rstrip("/") != "/root/reference"]
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

from audio_analysis_amd.synth import synth_ir  # noqa: E402

import analyse.zplane as rz  # noqa: E402  (the reference's)


def lowpassed(seed: int, cutoff_hz: float, n: int) -> np.ndarray:
    """A synthetic decaying-noise IR (SURVEY.md 8d generator) through a 4th-order Butterworth low-pass, float32, peak 1,
    starting at its largest sample (the zplane prologue trims there and divides by the peak, zplane.py:195-213)."""
    x = synth_ir(seed, 0, n + 2000, rt60_seconds=0.5).astype(np.float64)
    b, a = butter(4, cutoff_hz / 24000.0)
    y = lfilter(b, a, x)
    k = int(np.argmax(np.abs(y)))
    seg = y[k : k + n]
    return (seg / np.max(np.abs(seg))).astype(np.float32)


def main() -> None:
    out = {}
    meta = []
    for tag, seed, cutoff, n, order in (("lp500_p64", 31, 500.0, 24000, 64), ("lp500_p256", 32, 500.0, 24000, 256),
                                        ("lp2k_p64", 33, 2000.0, 24000, 64)):
        x = lowpassed(seed, cutoff, n)
        seg = x.astype(np.float64)
        a = rz._fit_ar_least_squares(seg, order, 0.0)
        poles = rz._roots_from_poly_descending(a)
        rows = np.arange(order, seg.size)
        A = np.stack([seg[rows - k] for k in range(1, order + 1)], axis=1)
        sv = np.linalg.svd(A, compute_uv=False)
        rank = int(np.linalg.lstsq(A, -seg[rows], rcond=None)[2])
        out[f"{tag}/x"] = x
        out[f"{tag}/coeffs"] = a
        out[f"{tag}/poles"] = poles
        out[f"{tag}/sv_max_min"] = np.array([sv[0], sv[-1]])
        out[f"{tag}/order_rank"] = np.array([order, rank])
        meta.append((tag, order, rank, float(sv[0] / sv[-1]), float(np.max(np.abs(poles)))))
    np.savez_compressed(HERE / "ar_illcond.npz", **out)
    for m in meta:
        print("%s: order %d, lstsq rank %d, cond(A) %.3g (cond(G) %.3g), max |pole| %.6f" % (m[0], m[1], m[2], m[3], m[3] ** 2, m[4]))


if __name__ == "__main__":
    main()
