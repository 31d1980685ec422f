"""
Spectrogram (STFT magnitude in dB) on the GPU.

Host-side mirror of the reference's analyse/spectrogram.py (settings/result dataclasses :37-83,
analyse_spectrogram_for_channel :168-217, summary :390-399); the STFT itself is ira_stft_mag_db.
float32 butterflies by default (`AUDIO_ANALYSIS_AMD_STFT_PRECISION=64` selects float64).
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np

from ..engine import get_engine
from ._common import frame_time_axis, segment_bounds, segment_bounds_batch, wav_channels


@dataclass(frozen=True)
class SpectrogramAnalysisSettings:
    use_mono_downmix_for_stereo: bool = False
    trim_to_peak: bool = True
    ignore_leading_seconds: float = 0.0
    analysis_duration_seconds: Optional[float] = None
    n_fft: int = 4096
    hop_length: int = 512
    use_hann_window: bool = True
    floor_db: float = -120.0
    f_min_hz: float = 20.0
    f_max_hz: float = 20000.0
    dynamic_range_db: Optional[float] = 90.0


@dataclass(frozen=True)
class SpectrogramPlotSettings:
    vmin_db: Optional[float] = None
    vmax_db: Optional[float] = None


@dataclass(frozen=True)
class ChannelSpectrogramResult:
    channel_name: str
    sample_rate_hz: int
    analysis_start_sample_index: int
    analysis_length_samples: int
    time_seconds: np.ndarray
    frequency_hz: np.ndarray
    magnitude_db: np.ndarray


def stft_precision() -> int:
    return 64 if os.environ.get("AUDIO_ANALYSIS_AMD_STFT_PRECISION", "32") == "64" else 32


def select_stft_segments(eng, batch, sample_rate_hz: int, settings, what: str):
    """Time selection shared by spectrogram / waterfall / modal cloud -> (starts, lens, nframes)."""
    n_fft, hop = int(settings.n_fft), int(settings.hop_length)
    if n_fft <= 0 or hop <= 0:
        raise ValueError("n_fft and hop_length must be positive.")
    # any positive frame size, like the reference (numpy.fft.rfft): powers of two in [64, 16384] run on the STFT kernels,
    # everything else on the arbitrary-length transforms (Engine._stft_generic)
    peaks = eng.peaks(batch) if settings.trim_to_peak else np.zeros(batch.count, dtype=np.int64)
    starts, lens = segment_bounds_batch(batch.length, peaks, sample_rate_hz, settings.trim_to_peak,
                                        settings.ignore_leading_seconds, settings.analysis_duration_seconds)
    if np.any(lens < n_fft):
        raise ValueError(f"Not enough samples after trimming/selection for {what} (need at least n_fft).")
    nframes = (1 + (lens - n_fft) // hop).astype(np.int32)
    return starts, lens, nframes


def spectrogram_device(eng, batch, sample_rate_hz: int, settings: "SpectrogramAnalysisSettings",
                       frame_major: bool = False):
    """Device-resident spectrograms: flat float32 buffer of C-contiguous (F, T_i) matrices -- or, with frame_major and
    a configuration ira_stft_mag_db_tf implements, of their (T_i, F) transposes (dict key "frame_major" says which)."""
    starts, lens, nframes = select_stft_segments(eng, batch, sample_rate_hz, settings, "spectrogram")
    tf = bool(frame_major) and eng.stft_frame_major_ok(int(settings.n_fft), stft_precision())
    mag, mag_off, cols = eng.stft_mag_db(batch.x, batch.off + starts, nframes, int(settings.n_fft),
                                         int(settings.hop_length), bool(settings.use_hann_window),
                                         float(settings.floor_db), stft_precision(), frame_major=tf)
    return dict(mag=mag, mag_off=mag_off, cols=cols, starts=starts, lens=lens, frame_major=tf)


def analyse_spectrogram_batch(
    channels: Sequence[np.ndarray],
    sample_rate_hz: int,
    channel_names: Sequence[str],
    settings: SpectrogramAnalysisSettings,
) -> List[ChannelSpectrogramResult]:
    for c in channels:
        if c.ndim != 1:
            raise ValueError("analyse_spectrogram_for_channel expects a 1D mono array.")
    eng = get_engine()
    batch = eng.upload(list(channels))
    return spectrogram_results(spectrogram_device(eng, batch, sample_rate_hz, settings), sample_rate_hz,
                               channel_names, settings)


def spectrogram_results(dev, sample_rate_hz: int, channel_names, settings) -> List[ChannelSpectrogramResult]:
    """Device -> host copy of the (F, T) matrices and result dataclasses."""
    out, out_off, cols, starts, lens = dev["mag"], dev["mag_off"], dev["cols"], dev["starts"], dev["lens"]
    n_fft, hop = int(settings.n_fft), int(settings.hop_length)
    host = out.cpu().numpy()
    f = n_fft // 2 + 1
    freq = np.fft.rfftfreq(n_fft, d=1.0 / float(sample_rate_hz)).astype(np.float32)
    res = []
    for i, name in enumerate(channel_names):
        t = int(cols[i])
        flat = host[out_off[i] : out_off[i] + f * t]
        mag = flat.reshape(t, f).T.copy() if dev.get("frame_major") else flat.reshape(f, t).copy()
        res.append(ChannelSpectrogramResult(
            channel_name=str(name), sample_rate_hz=int(sample_rate_hz), analysis_start_sample_index=int(starts[i]),
            analysis_length_samples=int(lens[i]), time_seconds=frame_time_axis(t, hop, sample_rate_hz),
            frequency_hz=freq, magnitude_db=mag,
        ))
    return res


def spectrogram_summary_lines(dev, sample_rate_hz: int, channel_names, settings) -> List[str]:
    """One summarise_spectrogram_results_text line per channel from the batch geometry alone; the matrices stay in HBM."""
    bins = int(settings.n_fft) // 2 + 1
    return list(
        f"[{name}] start_sample={int(dev['starts'][i])}  len_samples={int(dev['lens'][i])}  "
        f"dur={float(int(dev['lens'][i])) / float(int(sample_rate_hz)):.3f}s  "
        f"stft(n_fft={bins * 2 - 2}, frames={int(dev['cols'][i])})"
        for i, name in enumerate(channel_names))


def analyse_spectrogram_for_channel(
    samples: np.ndarray,
    sample_rate_hz: int,
    channel_name: str,
    settings: SpectrogramAnalysisSettings,
) -> ChannelSpectrogramResult:
    return analyse_spectrogram_batch([samples], sample_rate_hz, [channel_name], settings)[0]


def analyse_spectrogram_from_wav_file(
    input_wav_file_path: str | Path,
    settings: Optional[SpectrogramAnalysisSettings] = None,
) -> List[ChannelSpectrogramResult]:
    settings = settings or SpectrogramAnalysisSettings()
    loaded, chans = wav_channels(input_wav_file_path, settings.use_mono_downmix_for_stereo)
    return analyse_spectrogram_batch([c for _, c in chans], loaded.sample_rate_hz, [n for n, _ in chans], settings)


def plot_spectrogram_from_wav_file(
    input_wav_file_path: str | Path,
    analysis_settings: Optional[SpectrogramAnalysisSettings] = None,
    plot_settings: Optional[SpectrogramPlotSettings] = None,
    output_basename: Optional[str | Path] = None,
    show_interactive: bool = True,
) -> List[ChannelSpectrogramResult]:
    analysis_settings = analysis_settings or SpectrogramAnalysisSettings()
    plot_settings = plot_settings or SpectrogramPlotSettings()
    results = analyse_spectrogram_from_wav_file(input_wav_file_path, analysis_settings)
    from . import plotting
    for r in results:
        plotting.render_spectrogram(r, analysis_settings, plot_settings,
                                    f"Spectrogram — {input_wav_file_path} — {r.channel_name}",
                                    plotting.png_path(output_basename, f"_spectrogram_{r.channel_name}"),
                                    show_interactive)
    return results


def summarise_spectrogram_results_text(results: List[ChannelSpectrogramResult]) -> str:
    rows = []
    for r in results:
        rows.append(
            f"[{r.channel_name}] start_sample={r.analysis_start_sample_index}  "
            f"len_samples={r.analysis_length_samples}  "
            f"dur={float(r.analysis_length_samples) / float(r.sample_rate_hz):.3f}s  "
            f"stft(n_fft={r.magnitude_db.shape[0] * 2 - 2}, frames={r.magnitude_db.shape[1]})"
        )
    return "\n".join(rows)
