#!/usr/bin/env python3
"""
Golden vectors for the deconvolve row, made by importing the REFERENCE
(/root/reference/analyse/deconvolve.py) in the build container:

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_deconvolve_goldens.py

Committed output: tests/golden/deconvolve.npz -- inputs (PCM16 recordings, float32 sweeps) and the reference's
float32 impulse responses for several settings; numpy/scipy versions inside.  Data only.
"""
from __future__ import annotations

import os
import sys
import tempfile
from pathlib import Path

import numpy as np
import scipy
from scipy.io import wavfile

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path[:] = [p for p in sys.path if p.rstrip("/") != "/root/reference"]
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

from audio_analysis_amd.synth import synth_ir  # noqa: E402

import analyse.deconvolve as rdec  # noqa: E402  (the reference's)

SR = 48000


def log_sweep(n: int, f0: float, f1: float, lead: int, tail: int) -> np.ndarray:
    t = np.arange(n, dtype=np.float64) / SR
    dur = n / SR
    k = np.log(f1 / f0)
    ph = 2.0 * np.pi * f0 * dur / k * (np.exp(t / dur * k) - 1.0)
    s = 0.8 * np.sin(ph)
    fade = min(256, n // 8)
    s[:fade] *= np.linspace(0.0, 1.0, fade)
    s[-fade:] *= np.linspace(1.0, 0.0, fade)
    return np.concatenate([np.zeros(lead), s, np.zeros(tail)]).astype(np.float32)


def main():
    out = {"numpy": np.array(np.__version__), "scipy": np.array(scipy.__version__)}
    rng = np.random.default_rng(2024)
    sweep = log_sweep(9000, 40.0, 20000.0, 300, 700)                      # 10000 samples
    irs = [synth_ir(90 + c, c, 6000, rt60_seconds=0.05, pre_delay=37 + 11 * c) for c in (0, 1)]
    rec = np.stack([np.convolve(sweep.astype(np.float64), h.astype(np.float64))[:15000] for h in irs], axis=1)
    rec = rec / np.max(np.abs(rec)) * 0.9 + 1e-4 * rng.standard_normal(rec.shape) + 0.003       # noise + a DC offset
    rec16 = np.clip(np.round(rec * 32767.0), -32768, 32767).astype(np.int16)
    out["stereo/recorded_pcm16"] = rec16
    out["stereo/sweep"] = sweep
    cases = {
        "default": rdec.DeconvolveSettings(),
        "raw": rdec.DeconvolveSettings(normalise_peak=False, remove_dc=False),
        "full": rdec.DeconvolveSettings(output_length_mode="full_fft", regularization_relative=1e-6, target_peak=0.5),
    }
    for name, s in cases.items():
        out[f"stereo/{name}"] = rdec.deconvolve_impulse_response(rec16, sweep, SR, s)
    # mono float recording SHORTER than the sweep (n_fft follows the sweep), values beyond +-1 get clipped on entry
    sweep2 = log_sweep(18000, 20.0, 22000.0, 0, 1000)                      # 19000 samples -> n_fft 32768
    rec2 = (np.convolve(sweep2.astype(np.float64), irs[0].astype(np.float64))[:12345] * 0.2).astype(np.float32)
    out["mono/recorded_f32"] = rec2
    out["mono/sweep"] = sweep2
    out["mono/default"] = rdec.deconvolve_impulse_response(rec2, sweep2, SR, rdec.DeconvolveSettings())
    # file level: stereo sweep file (downmixed to mono by the reference), PCM16 recording, float32 WAV written
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        wavfile.write(str(tmp / "rec.wav"), SR, rec16)
        sw16 = np.clip(np.round(np.stack([sweep, 0.5 * sweep], axis=1) * 32767.0), -32768, 32767).astype(np.int16)
        wavfile.write(str(tmp / "sweep.wav"), SR, sw16)
        res = rdec.deconvolve_from_wav_files(tmp / "rec.wav", tmp / "sweep.wav", None, tmp / "out" / "ir.wav")
        out["file/sweep_pcm16"] = sw16
        out["file/ir"] = res.samples
        rate, written = wavfile.read(str(tmp / "out" / "ir.wav"))
        assert rate == SR and written.dtype == np.float32 and np.array_equal(written, res.samples)
        out["file/default_name"] = np.array(rdec.default_output_ir_path("/x/y/rec.take1.wav").name)
    np.savez_compressed(HERE / "deconvolve.npz", **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
