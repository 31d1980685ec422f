"""
Drop-in host layer for the reference's `analyse` package (hot path only).

Same module names, settings/result dataclasses, function signatures and error behaviour as
kianmcevoy/audio_analysis `analyse/`, but every numeric body runs on the GPU through libira.so
(audio_analysis_amd.engine).  There is no CPU fallback.

The package namespace re-exports the WAV-loading names of `.io`, as the reference's package does.
"""
from . import io as _io

# public names of the package namespace = what the reference's `analyse/__init__.py` re-exports from its io module
__all__ = sorted(
    "LoadedAudio DEFAULT_EXPECTED_SAMPLE_RATE_HZ convert_wav_samples_to_float32 downmix_to_mono "
    "duplicate_mono_to_stereo get_channel get_left_right load_wav_file validate_audio_format".split())
globals().update({_name: getattr(_io, _name) for _name in __all__})
get_analysis_channels = _io.get_analysis_channels          # extra: used by this package's own modules
