"""
Filter response (magnitude + phase of the whole-segment spectrum) on the GPU.

Host-side mirror of the reference's analyse/filterplot.py (dataclasses :43-104,
analyse_filter_response_for_channel :112-203, summary :382-390).  Same float64 Bluestein rFFT as the
frequency-response module; the phase unwrap (numpy.unwrap semantics) is a parallel scan (ira_phase_unwrap).
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ..engine import get_engine
from ._common import wav_channels
from .frequency_response import spectrum_device
from .io import get_analysis_channels, load_wav_file


@dataclass(frozen=True)
class FilterAnalysisSettings:
    use_mono_downmix_for_stereo: bool = False
    trim_to_peak: bool = True
    ignore_leading_seconds: float = 0.0
    analysis_duration_seconds: Optional[float] = None
    use_hann_window: bool = True
    magnitude_floor_db: float = -120.0
    f_min_hz: float = 20.0
    f_max_hz: float = 20000.0
    phase_mode: str = "degrees"
    unwrap_phase: bool = True


@dataclass(frozen=True)
class FilterPlotSettings:
    secondary_channel_alpha: float = 0.7
    magnitude_ylim_db: Optional[Tuple[float, float]] = None
    phase_ylim: Optional[Tuple[float, float]] = None


@dataclass(frozen=True)
class ChannelFilterResponse:
    channel_name: str
    sample_rate_hz: int
    analysis_start_sample_index: int
    analysis_length_samples: int
    frequency_hz: np.ndarray
    magnitude_db: np.ndarray
    phase_response: np.ndarray
    peak_frequency_hz: float
    magnitude_at_1khz_db: float


def analyse_filter_response_batch(
    channels: Sequence[np.ndarray],
    sample_rate_hz: int,
    channel_names: Sequence[str],
    settings: FilterAnalysisSettings,
) -> List[ChannelFilterResponse]:
    for c in channels:
        if c.ndim != 1:
            raise ValueError("analyse_filter_response_for_channel expects a 1D mono array.")
    eng = get_engine()
    batch = eng.upload(list(channels))
    dev = spectrum_device(eng, batch, sample_rate_hz, settings, "filter response", want_phase=True,
                          unwrap=bool(settings.unwrap_phase), degrees=settings.phase_mode == "degrees")
    starts, lens, off = dev["starts"], dev["lens"], dev["off"]
    stats = dev["stats"].cpu().numpy()
    mag_host, ph_host = dev["mag"].cpu().numpy(), dev["phase"].cpu().numpy()
    out = []
    for i, name in enumerate(channel_names):
        n = int(lens[i])
        bins = n // 2 + 1
        if stats[i, 0] < 1.0:
            raise ValueError("Selected frequency range is empty.")
        out.append(ChannelFilterResponse(
            channel_name=name, sample_rate_hz=sample_rate_hz, analysis_start_sample_index=int(starts[i]),
            analysis_length_samples=n,
            frequency_hz=np.fft.rfftfreq(n, d=1.0 / float(sample_rate_hz)).astype(np.float32),
            magnitude_db=mag_host[off[i] : off[i] + bins].copy(),
            phase_response=ph_host[off[i] : off[i] + bins].copy(),
            peak_frequency_hz=float(stats[i, 2]), magnitude_at_1khz_db=float(stats[i, 7]),
        ))
    return out


def analyse_filter_response_for_channel(
    samples: np.ndarray,
    sample_rate_hz: int,
    channel_name: str,
    settings: FilterAnalysisSettings,
) -> ChannelFilterResponse:
    return analyse_filter_response_batch([samples], sample_rate_hz, [channel_name], settings)[0]


def analyse_filter_response_from_wav_file(
    input_wav_file_path: str | Path,
    settings: FilterAnalysisSettings,
) -> List[ChannelFilterResponse]:
    loaded = load_wav_file(wav_file_path=Path(input_wav_file_path), expected_sample_rate_hz=48000,
                           expected_channel_mode="mono_or_stereo", allow_mono_and_upmix_to_stereo=False)
    chans = get_analysis_channels(loaded_audio=loaded,
                                  use_mono_downmix_for_stereo=bool(settings.use_mono_downmix_for_stereo))
    return analyse_filter_response_batch([c for _, c in chans], int(loaded.sample_rate_hz), [n for n, _ in chans],
                                         settings)


def plot_filter_response_from_wav_file(
    input_wav_file_path: str | Path,
    analysis_settings: Optional[FilterAnalysisSettings] = None,
    plot_settings: Optional[FilterPlotSettings] = None,
    output_basename: Optional[str | Path] = None,
    show_interactive: bool = True,
) -> List[ChannelFilterResponse]:
    analysis_settings = analysis_settings or FilterAnalysisSettings()
    plot_settings = plot_settings or FilterPlotSettings()
    results = analyse_filter_response_from_wav_file(input_wav_file_path, analysis_settings)
    from . import plotting
    plotting.render_filter_response(results, analysis_settings, plot_settings,
                                    f"Filter frequency response — {input_wav_file_path}",
                                    plotting.png_path(output_basename, "_filter"), show_interactive)
    return results


def summarise_filter_response_results_text(channel_results: List[ChannelFilterResponse]) -> str:
    return "\n".join(
        f"[{r.channel_name}] start_sample={r.analysis_start_sample_index}  "
        f"len_samples={r.analysis_length_samples}  "
        f"peak={r.peak_frequency_hz:.1f}Hz  @1kHz={r.magnitude_at_1khz_db:.1f}dB"
        for r in channel_results
    )
