"""
Impulse-response view: three PNGs (full waveform, early zoom, log-magnitude tail), no metrics.

Host-side mirror of the reference's analyse/impulse_response.py (ImpulseResponseViewSettings :43-50,
compute_log_magnitude :53-60, plot_impulse_response_waveform :63-132, plot_impulse_response_log_magnitude :135-184,
plot_ir_from_wav_file :196-239).  There are no numerics to accelerate here: it exists so
that the `ir` command and the first block of the default `report` behave like the reference's -- same file names
(<basename>.png, <basename>_early.png, <basename>_tail.png, including the reference's with_suffix() quirk for
basenames that contain dots), same Markdown block.  Drawn with the minimal matplotlib helpers of analyse/plotting.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Optional

import numpy as np

from .io import LoadedAudio, get_analysis_channels, load_wav_file


@dataclass(frozen=True)
class ImpulseResponseViewSettings:
    early_window_seconds: float = 0.08
    log_magnitude_floor_db: float = -120.0
    use_mono_downmix: bool = False


def compute_log_magnitude(samples: np.ndarray) -> np.ndarray:
    """Magnitude-like envelope for log plotting: |x| as float32 (reference :53-60)."""
    return np.abs(samples).astype(np.float32)


def _suffix_output_path(output_path: str | Path, suffix: str) -> Path:
    output_path = Path(output_path)
    return output_path.with_name(f"{output_path.stem}{suffix}{output_path.suffix}")


def _channels_with_alpha(loaded_audio: LoadedAudio, settings: ImpulseResponseViewSettings):
    chans = get_analysis_channels(loaded_audio, use_mono_downmix_for_stereo=settings.use_mono_downmix)
    return [(name, x, 1.0 if i == 0 else 0.5) for i, (name, x) in enumerate(chans)]


def plot_impulse_response_waveform(loaded_audio: LoadedAudio, settings: ImpulseResponseViewSettings,
                                   output_path: Optional[str | Path] = None, show_interactive: bool = True) -> None:
    from . import plotting
    n, sr = int(loaded_audio.samples.shape[0]), int(loaded_audio.sample_rate_hz)
    t = np.arange(n, dtype=np.float64) / float(sr)
    chans = _channels_with_alpha(loaded_audio, settings)
    fig, ax = plotting.new_axes(f"Waveform (full) - {loaded_audio.file_path.name}")
    for name, x, alpha in chans:
        ax.plot(t, x, alpha=alpha, label=name)
    ax.set_xlabel("Time (s)"); ax.set_ylabel("Amplitude")
    plotting.finish(fig, None if output_path is None else Path(output_path), show_interactive)

    early = max(1, min(int(round(settings.early_window_seconds * sr)), n))
    fig, ax = plotting.new_axes(f"Waveform (early {settings.early_window_seconds * 1000:.0f} ms) - "
                                f"{loaded_audio.file_path.name}")
    for name, x, alpha in chans:
        ax.plot(t[:early], x[:early], alpha=alpha, label=name)
    ax.set_xlabel("Time (s)"); ax.set_ylabel("Amplitude")
    plotting.finish(fig, None if output_path is None else _suffix_output_path(output_path, "_early"), show_interactive)


def plot_impulse_response_log_magnitude(loaded_audio: LoadedAudio, settings: ImpulseResponseViewSettings,
                                        output_path: Optional[str | Path] = None, show_interactive: bool = True) -> None:
    from . import plotting
    n, sr = int(loaded_audio.samples.shape[0]), int(loaded_audio.sample_rate_hz)
    t = np.arange(n, dtype=np.float64) / float(sr)
    fig, ax = plotting.new_axes(f"Log magnitude (tail) - {loaded_audio.file_path.name}")
    floor_db = float(settings.log_magnitude_floor_db)
    for name, x, alpha in _channels_with_alpha(loaded_audio, settings):
        mag = np.maximum(compute_log_magnitude(x), 10 ** (floor_db / 20.0))
        ax.plot(t, 20.0 * np.log10(mag), alpha=alpha, label=name)
    ax.set_ylim(bottom=floor_db)
    ax.set_xlabel("Time (s)"); ax.set_ylabel("Magnitude (dB)")
    if not settings.use_mono_downmix:
        ax.legend()
    plotting.finish(fig, None if output_path is None else Path(output_path), show_interactive)


def ir_output_paths(output_basename: Optional[str | Path]):
    """(waveform png, tail png) exactly as the reference derives them (:223-229)."""
    if output_basename is None:
        return None, None
    base = Path(output_basename)
    return base.with_suffix(".png"), base.with_name(f"{base.stem}_tail.png").with_suffix(".png")


def plot_ir_views(loaded_audio: LoadedAudio, settings: ImpulseResponseViewSettings, output_basename, show_interactive: bool
                  ) -> None:
    wave, tail = ir_output_paths(output_basename)
    plot_impulse_response_waveform(loaded_audio, settings, wave, show_interactive)
    plot_impulse_response_log_magnitude(loaded_audio, settings, tail, show_interactive)


def plot_ir_from_wav_file(wav_file_path: str | Path, settings: Optional[ImpulseResponseViewSettings] = None,
                          output_basename: Optional[str | Path] = None, show_interactive: bool = True) -> None:
    settings = settings or ImpulseResponseViewSettings()
    loaded = load_wav_file(wav_file_path, expected_channel_mode="mono_or_stereo", allow_mono_and_upmix_to_stereo=False)
    plot_ir_views(loaded, settings, output_basename, show_interactive)
