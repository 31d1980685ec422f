"""
Drop-in host layer for the reference's `analyse` package (hot path only).

Same module names, settings/result dataclasses, function signatures and error behaviour as
kianmcevoy/audio_analysis `analyse/`, but every numeric body runs on the GPU through libira.so
(audio_analysis_amd.engine).  There is no CPU fallback.
"""
from .io import (  # noqa: F401
    DEFAULT_EXPECTED_SAMPLE_RATE_HZ,
    LoadedAudio,
    convert_wav_samples_to_float32,
    downmix_to_mono,
    duplicate_mono_to_stereo,
    get_analysis_channels,
    get_channel,
    get_left_right,
    load_wav_file,
    validate_audio_format,
)

__all__ = [
    "LoadedAudio",
    "DEFAULT_EXPECTED_SAMPLE_RATE_HZ",
    "convert_wav_samples_to_float32",
    "downmix_to_mono",
    "duplicate_mono_to_stereo",
    "get_channel",
    "get_left_right",
    "load_wav_file",
    "validate_audio_format",
]
