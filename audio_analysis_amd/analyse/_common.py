"""Host-side helpers shared by the drop-in modules: time selection, batching, WAV channel loading."""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import numpy as np

from ..engine import ChannelBatch, Engine, get_engine
from .io import get_analysis_channels, load_wav_file


def segment_bounds(n: int, peak: int, sample_rate_hz: int, trim_to_peak: bool, ignore_leading_seconds: float,
                   duration_seconds: Optional[float]) -> Tuple[int, int]:
    """
    (start, length) of the analysed slice: drop [0, peak) when trimming, skip round(ignore*sr) samples
    (clamped), optionally keep round(duration*sr) samples (clamped).  Integer arithmetic, bit-exact with
    the prologue every reference module repeats (e.g. decay.py:135-144, spectrogram.py:180-194).
    """
    start, length = (peak, n - peak) if trim_to_peak else (0, n)
    if ignore_leading_seconds > 0.0:
        skip = int(round(float(ignore_leading_seconds) * float(sample_rate_hz)))
        skip = max(0, min(skip, length))
        start, length = start + skip, length - skip
    if duration_seconds is not None:
        keep = int(round(float(duration_seconds) * float(sample_rate_hz)))
        length = max(0, min(keep, length))
    return start, length


def segment_bounds_batch(n: np.ndarray, peak: np.ndarray, sample_rate_hz: int, trim_to_peak: bool,
                         ignore_leading_seconds: float, duration_seconds: Optional[float]) -> Tuple[np.ndarray, np.ndarray]:
    """segment_bounds for arrays of lengths and peak indices (int64 in, int64 out): the same integer arithmetic, without a
    Python call per channel (the metrics pipeline asks for 256 channels per step, several times)."""
    n = np.asarray(n, dtype=np.int64)
    peak = np.asarray(peak, dtype=np.int64)
    start = peak.copy() if trim_to_peak else np.zeros_like(n)
    length = n - peak if trim_to_peak else n.copy()
    if ignore_leading_seconds > 0.0:
        skip = int(round(float(ignore_leading_seconds) * float(sample_rate_hz)))
        sk = np.maximum(0, np.minimum(skip, length))
        start, length = start + sk, length - sk
    if duration_seconds is not None:
        keep = int(round(float(duration_seconds) * float(sample_rate_hz)))
        length = np.maximum(0, np.minimum(keep, length))
    return start, length


def as_batch(channels: Sequence[np.ndarray]) -> Tuple[Engine, ChannelBatch]:
    eng = get_engine()
    return eng, eng.upload(list(channels))


def wav_channels(path, use_mono_downmix_for_stereo: bool, **load_kw):
    loaded = load_wav_file(wav_file_path=path, expected_channel_mode="mono_or_stereo",
                           allow_mono_and_upmix_to_stereo=False, **load_kw)
    return loaded, get_analysis_channels(loaded_audio=loaded, use_mono_downmix_for_stereo=use_mono_downmix_for_stereo)


def frame_time_axis(num_frames: int, hop_length: int, sample_rate_hz: int) -> np.ndarray:
    """Frame-start times in float32 arithmetic (reference spectrogram.py:158)."""
    return (np.arange(num_frames, dtype=np.float32) * float(hop_length) / float(sample_rate_hz)).astype(np.float32)
