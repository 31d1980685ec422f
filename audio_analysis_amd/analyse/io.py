"""
WAV ingest and channel policy -- host-side mirror of the reference's analyse/io.py
(semantics: io.py:46-113 conversion, :66-95 channel policy, :156-178 validation, :181-221 load).

Internal format everywhere: float32 in [-1, 1], shape (num_samples, num_channels), 48 kHz.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import List, Literal, Tuple, Union

import numpy as np
from scipy.io import wavfile

ChannelMode = Literal["mono", "stereo", "mono_or_stereo"]
DEFAULT_EXPECTED_SAMPLE_RATE_HZ = 48_000

_INT_SCALE = {np.dtype(np.int16): 32768.0, np.dtype(np.int32): 2147483648.0}


@dataclass(frozen=True)
class LoadedAudio:
    samples: np.ndarray
    sample_rate_hz: int
    file_path: Path


def convert_wav_samples_to_float32(samples_from_wav: np.ndarray) -> np.ndarray:
    """Any WAV sample dtype -> float32 clipped to [-1, 1] (int16 / 2^15, int32 / 2^31, floats pass through)."""
    dt = samples_from_wav.dtype
    if np.issubdtype(dt, np.floating):
        as_float = samples_from_wav.astype(np.float32, copy=False)
    elif np.issubdtype(dt, np.integer):
        if dt not in _INT_SCALE:
            raise ValueError(f"Unsupported integer PCM dtype: {dt}")
        as_float = samples_from_wav.astype(np.float32) / _INT_SCALE[dt]
    else:
        raise ValueError(f"Unsupported WAV dtype: {dt}")
    return np.clip(as_float, -1.0, 1.0).astype(np.float32)


def ensure_2d_channel_array(float_samples: np.ndarray) -> np.ndarray:
    if float_samples.ndim == 2:
        return float_samples
    if float_samples.ndim == 1:
        return float_samples.reshape((-1, 1))
    raise ValueError(f"Expected 1D or 2D audio array, got shape {float_samples.shape}")


def duplicate_mono_to_stereo(float_samples: np.ndarray) -> np.ndarray:
    s = ensure_2d_channel_array(float_samples)
    if s.shape[1] == 2:
        return s.astype(np.float32)
    if s.shape[1] == 1:
        return np.repeat(s, 2, axis=1).astype(np.float32)
    raise ValueError(f"Expected mono or stereo for upmix, got {s.shape[1]} channels")


def downmix_to_mono(float_samples: np.ndarray) -> np.ndarray:
    s = ensure_2d_channel_array(float_samples)
    return np.mean(s, axis=1, dtype=np.float32).reshape((-1, 1)).astype(np.float32)


def get_analysis_channels(
    loaded_audio: LoadedAudio,
    use_mono_downmix_for_stereo: bool = False,
) -> List[Tuple[str, np.ndarray]]:
    """[("mono", x)] for mono files; [("left", L), ("right", R)] or [("mono", 0.5*(L+R))] for stereo."""
    s = loaded_audio.samples
    n_ch = s.shape[1]
    if n_ch == 1:
        return [("mono", s[:, 0].astype(np.float32, copy=False))]
    if n_ch != 2:
        raise ValueError(f"Unsupported channel count: {n_ch}")
    left = s[:, 0].astype(np.float32, copy=False)
    right = s[:, 1].astype(np.float32, copy=False)
    if use_mono_downmix_for_stereo:
        return [("mono", 0.5 * (left + right))]
    return [("left", left), ("right", right)]


def validate_audio_format(
    loaded_audio: LoadedAudio,
    expected_sample_rate_hz: int = DEFAULT_EXPECTED_SAMPLE_RATE_HZ,
    expected_channel_mode: ChannelMode = "stereo",
) -> None:
    if loaded_audio.sample_rate_hz != expected_sample_rate_hz:
        raise ValueError(
            f"Expected sample rate {expected_sample_rate_hz} Hz, "
            f"but got {loaded_audio.sample_rate_hz} Hz for file {loaded_audio.file_path}"
        )
    n_ch = loaded_audio.samples.shape[1]
    where = f"for file {loaded_audio.file_path}"
    if expected_channel_mode == "mono" and n_ch != 1:
        raise ValueError(f"Expected mono (1 channel) but got {n_ch} channels {where}")
    if expected_channel_mode == "stereo" and n_ch != 2:
        raise ValueError(f"Expected stereo (2 channels) but got {n_ch} channels {where}")
    if expected_channel_mode == "mono_or_stereo" and n_ch not in (1, 2):
        raise ValueError(f"Expected mono or stereo (1 or 2 channels) but got {n_ch} channels {where}")


def load_wav_file(
    wav_file_path: Union[str, Path],
    expected_sample_rate_hz: int = DEFAULT_EXPECTED_SAMPLE_RATE_HZ,
    expected_channel_mode: ChannelMode = "stereo",
    allow_mono_and_upmix_to_stereo: bool = True,
) -> LoadedAudio:
    path = Path(wav_file_path)
    rate, raw = wavfile.read(str(path))
    samples = ensure_2d_channel_array(convert_wav_samples_to_float32(raw))
    if expected_channel_mode == "stereo" and allow_mono_and_upmix_to_stereo and samples.shape[1] == 1:
        samples = duplicate_mono_to_stereo(samples)
    loaded = LoadedAudio(samples=samples.astype(np.float32, copy=False), sample_rate_hz=int(rate), file_path=path)
    validate_audio_format(loaded, expected_sample_rate_hz, expected_channel_mode)
    return loaded


def get_channel(loaded_audio: LoadedAudio, channel_index: int) -> np.ndarray:
    n_ch = loaded_audio.samples.shape[1]
    if not 0 <= channel_index < n_ch:
        raise ValueError(f"channel_index out of range: {channel_index} for {n_ch} channels")
    return loaded_audio.samples[:, channel_index].astype(np.float32, copy=False)


def get_left_right(loaded_audio: LoadedAudio) -> Tuple[np.ndarray, np.ndarray]:
    validate_audio_format(loaded_audio, expected_channel_mode="stereo")
    return get_channel(loaded_audio, 0), get_channel(loaded_audio, 1)
