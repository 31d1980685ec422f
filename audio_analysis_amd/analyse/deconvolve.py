"""
Sweep deconvolution: recorded system output + the sweep that was played -> impulse response.

Host-side mirror of the reference's analyse/deconvolve.py (DeconvolveSettings :53-69, DeconvolvedImpulseResponse
:72-77, deconvolve_impulse_response :124-193, deconvolve_from_wav_files :201-259, default_output_ir_path :262-268):
    H(w) = Y(w) conj(X(w)) / (|X(w)|^2 + eps),   eps = regularization_relative * max |X|^2
with the transforms zero-padded to the next power of two.  All numerics run on the device: the zero-padded float64
rFFTs and the inverse are the long-FFT kernels of the RT60 filter bank (two channels of a recording ride one inverse
transform), the spectral division and the DC / peak epilogue are ira_deconv_divide / ira_deconv_finish.  The result
stays usable on the device (`deconvolve_device`) so a bundle of recorded sweeps can go straight into the report
blocks; the reference-shaped functions below return NumPy float32 (N, C) like the reference.

Limit: n_fft <= 2^21 (43.6 s at 48 kHz) -- the largest transform the long-FFT kernels take.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from ..engine import ChannelBatch, Engine, get_engine
from .io import convert_wav_samples_to_float32, ensure_2d_channel_array, load_wav_file

MAX_LOG2_FFT = 21


@dataclass(frozen=True)
class DeconvolveSettings:
    regularization_relative: float = 1e-10
    normalise_peak: bool = True
    target_peak: float = 0.95
    remove_dc: bool = True
    output_length_mode: str = "recorded"          # "recorded" | "full_fft"


@dataclass(frozen=True)
class DeconvolvedImpulseResponse:
    samples: np.ndarray          # (N, C) float32
    sample_rate_hz: int
    recorded_file_path: Path
    sweep_file_path: Path


def _next_power_of_two(n: int) -> int:
    return 1 if n <= 1 else 1 << (int(n - 1).bit_length())


def _downmix_to_mono_1d(samples_2d: np.ndarray) -> np.ndarray:
    if samples_2d.ndim != 2:
        raise ValueError("Expected a 2D array (N,C).")
    return np.mean(samples_2d.astype(np.float64, copy=False), axis=1).astype(np.float32)


def deconvolve_device(eng: Engine, recorded: ChannelBatch, file_of_channel: Sequence[int], sweeps: ChannelBatch,
                      sweep_of_channel: Sequence[int], sample_rate_hz: int, settings: DeconvolveSettings):
    """
    Batched deconvolution on the device.  `recorded`: the channels of one or many recordings; file_of_channel[e] = the
    recording channel e belongs to (channels of a file share the peak normalisation); sweeps / sweep_of_channel[e] = the
    mono excitation each channel is deconvolved with.  Returns dict(h=flat float32 device buffer, off, n_out, n_fft):
    channel e's response is h[off[e] : off[e] + n_out[e]].
    """
    t = eng.torch
    if settings.output_length_mode not in ("recorded", "full_fft"):
        raise ValueError(f"Unknown output_length_mode: {settings.output_length_mode}")
    nb = recorded.count
    group = np.ascontiguousarray(file_of_channel, dtype=np.int32)
    sw_of = np.ascontiguousarray(sweep_of_channel, dtype=np.int64)
    n_rec = recorded.length.astype(np.int64)
    n_sw = sweeps.length.astype(np.int64)[sw_of]
    if nb == 0:
        return dict(h=eng.empty(0, t.float32), off=np.zeros(0, np.int64), n_out=np.zeros(0, np.int32),
                    n_fft=np.zeros(0, np.int32))
    if np.any(n_rec < 8) or np.any(n_sw < 8):
        raise ValueError("Recorded and sweep must both contain at least a few samples.")
    n_fft = np.array([_next_power_of_two(int(max(a, b))) for a, b in zip(n_rec, n_sw)], dtype=np.int64)
    if np.any(n_fft > (1 << MAX_LOG2_FFT)):
        raise ValueError(f"deconvolution transforms are limited to 2^{MAX_LOG2_FFT} points on the GPU path.")
    # sweep spectra: one per distinct (sweep, n_fft)
    keys, inverse = np.unique(np.stack([sw_of, n_fft], axis=1), axis=0, return_inverse=True)
    inverse = inverse.reshape(-1)
    xspec, xspec_off = eng.rfft_any(sweeps.x, sweeps.off[keys[:, 0]], keys[:, 1].astype(np.int32), False,
                                    data_len=np.minimum(sweeps.length[keys[:, 0]], keys[:, 1]).astype(np.int32),
                                    win_len=keys[:, 1].astype(np.int32))
    yspec, yspec_off = eng.rfft_any(recorded.x, recorded.off, n_fft.astype(np.int32), False,
                                    data_len=np.minimum(n_rec, n_fft).astype(np.int32), win_len=n_fft.astype(np.int32))
    eng.deconv_divide(yspec, yspec_off, xspec, xspec_off[inverse], n_fft.astype(np.int32),
                      float(settings.regularization_relative))
    # inverse transforms: all-pass "band" (a high-pass whose pass edge lies below 0 Hz), float32 out, n_fft samples each
    h_off = np.zeros(nb, dtype=np.int64)
    if nb > 1:
        h_off[1:] = np.cumsum(n_fft[:-1])
    h = eng.empty(int(n_fft.sum()), t.float32)
    allpass = np.zeros((nb, 8), dtype=np.float64)
    allpass[:, 0], allpass[:, 1], allpass[:, 2] = 2.0, -1.0, -1.0
    eng.band_irfft(yspec, yspec_off, n_fft.astype(np.int32), allpass, float(sample_rate_hz) / n_fft.astype(np.float64),
                   h, h_off)
    n_out = (n_rec if settings.output_length_mode == "recorded" else n_fft).astype(np.int32)
    eng.deconv_finish(h, h_off, n_out, group, bool(settings.remove_dc), bool(settings.normalise_peak),
                      float(settings.target_peak))
    return dict(h=h, off=h_off, n_out=n_out, n_fft=n_fft.astype(np.int32))


def deconvolve_impulse_response(
    recorded_samples_2d: np.ndarray,
    sweep_samples_1d: np.ndarray,
    sample_rate_hz: int,
    settings: DeconvolveSettings,
) -> np.ndarray:
    """IR for each channel of the recording using the same mono sweep; (N_out, C) float32 (reference :124-193)."""
    rec = ensure_2d_channel_array(convert_wav_samples_to_float32(np.asarray(recorded_samples_2d)))
    sweep = np.asarray(sweep_samples_1d, dtype=np.float32)
    if rec.shape[0] < 8 or sweep.size < 8:
        raise ValueError("Recorded and sweep must both contain at least a few samples.")
    eng = get_engine()
    c = int(rec.shape[1])
    batch = eng.upload([np.ascontiguousarray(rec[:, k]) for k in range(c)])
    sw = eng.upload([sweep.reshape(-1)])
    dev = deconvolve_device(eng, batch, [0] * c, sw, [0] * c, sample_rate_hz, settings)
    host = dev["h"].cpu().numpy()
    n_out = int(dev["n_out"][0])
    return np.stack([host[int(o) : int(o) + n_out] for o in dev["off"]], axis=1).astype(np.float32)


def _write_wav_float32(path: Path, sample_rate_hz: int, samples_2d: np.ndarray) -> None:
    from scipy.io import wavfile
    path.parent.mkdir(parents=True, exist_ok=True)
    wavfile.write(str(path), int(sample_rate_hz), samples_2d.astype(np.float32, copy=False))


def deconvolve_from_wav_files(
    recorded_wav_file_path: str | Path,
    sweep_wav_file_path: str | Path,
    settings: Optional[DeconvolveSettings] = None,
    output_ir_wav_file_path: Optional[str | Path] = None,
) -> DeconvolvedImpulseResponse:
    """Load recorded + sweep WAVs (48 kHz, mono or stereo; the sweep is mixed down to mono) and produce the IR;
    optionally write it as a float32 WAV (reference :201-259)."""
    settings = settings or DeconvolveSettings()
    recorded = load_wav_file(wav_file_path=recorded_wav_file_path, expected_channel_mode="mono_or_stereo",
                             allow_mono_and_upmix_to_stereo=False)
    sweep = load_wav_file(wav_file_path=sweep_wav_file_path, expected_channel_mode="mono_or_stereo",
                          allow_mono_and_upmix_to_stereo=False)
    if recorded.sample_rate_hz != sweep.sample_rate_hz:
        raise ValueError(f"Sample rate mismatch: recorded={recorded.sample_rate_hz} Hz, sweep={sweep.sample_rate_hz} Hz")
    samples = deconvolve_impulse_response(recorded.samples, _downmix_to_mono_1d(sweep.samples),
                                          recorded.sample_rate_hz, settings)
    ir = DeconvolvedImpulseResponse(samples=samples, sample_rate_hz=int(recorded.sample_rate_hz),
                                    recorded_file_path=Path(recorded.file_path), sweep_file_path=Path(sweep.file_path))
    if output_ir_wav_file_path is not None:
        _write_wav_float32(Path(output_ir_wav_file_path), ir.sample_rate_hz, ir.samples)
    return ir


def default_output_ir_path(recorded_wav_file_path: str | Path) -> Path:
    """<recorded_stem>_ir.wav in the same folder (reference :262-268)."""
    p = Path(recorded_wav_file_path)
    return p.with_name(f"{p.stem}_ir.wav")
