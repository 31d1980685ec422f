"""
Modal cloud: per-log-frequency-bin decay-time estimates from the STFT, on the GPU.

Host-side mirror of the reference's analyse/modalcloud.py (dataclasses :45-113, log bins :166-207,
per-bin fit :238-281, analyse_modal_cloud_for_channel :289-391, summary :557-567).  Device work:
STFT (n_fft 8192, float64 butterflies -- the per-bin decay fits are discrete functions of the float32 dB
values, and 94 frames/s leave no room for float32-FFT noise to flip a crossing), log-bin aggregation
(ira_logbin_aggregate) and the peak-normalised crossing/line fits (ira_curve_fits, rel_to_peak).
"""
from __future__ import annotations

from dataclasses import dataclass
from math import ceil, log2
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ..engine import get_engine
from ._common import wav_channels
from .spectrogram import select_stft_segments


@dataclass(frozen=True)
class ModalCloudAnalysisSettings:
    use_mono_downmix_for_stereo: bool = False
    trim_to_peak: bool = True
    ignore_leading_seconds: float = 0.0
    analysis_duration_seconds: Optional[float] = None
    n_fft: int = 8192
    hop_length: int = 512
    use_hann_window: bool = True
    f_min_hz: float = 20.0
    f_max_hz: float = 20000.0
    log_bins_per_octave: int = 24
    min_bins: int = 24
    floor_db: float = -120.0
    fit_lower_limit_db: float = -80.0
    t30_range_db: Tuple[float, float] = (-5.0, -35.0)
    t20_range_db: Tuple[float, float] = (-5.0, -25.0)
    edt_range_db: Tuple[float, float] = (0.0, -10.0)
    metric: str = "t30"
    min_fit_points: int = 10
    min_peak_db_above_floor: float = 20.0


@dataclass(frozen=True)
class ModalCloudPlotSettings:
    secondary_channel_alpha: float = 0.7
    show_median_curve: bool = True
    median_octave_window: float = 0.25
    ylim_seconds: Optional[Tuple[float, float]] = None


@dataclass(frozen=True)
class ModalPoint:
    centre_hz: float
    rt60_seconds: float
    r_squared: float


@dataclass(frozen=True)
class ChannelModalCloudResult:
    channel_name: str
    sample_rate_hz: int
    analysis_start_sample_index: int
    analysis_length_samples: int
    metric: str
    points: List[ModalPoint]


def _build_log_bins(f_min_hz: float, f_max_hz: float, bins_per_octave: int, min_bins: int) -> np.ndarray:
    """Log-spaced bin edges (float32), ~bins_per_octave per octave between f_min and f_max."""
    lo = float(max(1.0, f_min_hz))
    hi = float(max(lo * 1.001, f_max_hz))
    span = float(log2(hi / lo))
    count = int(max(min_bins, ceil(span * float(max(4, bins_per_octave)))))
    return (lo * (2.0 ** np.linspace(0.0, span, count + 1, dtype=np.float64))).astype(np.float32)


def log_bin_rows(freq_sel: np.ndarray, edges_f32: np.ndarray):
    """
    For each log bin the contiguous run of selected rFFT rows with lo <= f < hi (float32 compares, as the
    reference's boolean mask) -> (centres float32, first_row int32, row_count int32).
    """
    e = edges_f32.astype(np.float64)
    centres = np.sqrt(e[:-1] * e[1:]).astype(np.float32)
    lo32 = edges_f32[:-1].astype(np.float32)
    hi32 = edges_f32[1:].astype(np.float32)
    first = np.searchsorted(freq_sel, lo32, side="left").astype(np.int32)    # first row with f >= lo
    end = np.searchsorted(freq_sel, hi32, side="left").astype(np.int32)      # first row with f >= hi
    count = np.maximum(end - first, 0).astype(np.int32)
    return centres, first, count


def analyse_modal_cloud_batch(
    channels: Sequence[np.ndarray],
    sample_rate_hz: int,
    channel_names: Sequence[str],
    settings: ModalCloudAnalysisSettings,
) -> List[ChannelModalCloudResult]:
    eng = get_engine()
    batch = eng.upload(list(channels))
    dev = modal_cloud_device(eng, batch, sample_rate_hz, settings)
    return modal_records_to_results(dev, dev["fits"].cpu().numpy(), sample_rate_hz, channel_names)


def modal_records_to_results(dev, rec: np.ndarray, sample_rate_hz: int, channel_names):
    centres, nbins = dev["centres"], dev["nbins"]
    rec = rec.reshape(len(channel_names), nbins, 8)
    out = []
    for i, name in enumerate(channel_names):
        pts = [ModalPoint(centre_hz=float(centres[b]), rt60_seconds=float(rec[i, b, 6]), r_squared=float(rec[i, b, 5]))
               for b in range(nbins) if rec[i, b, 0] == 1.0]
        pts.sort(key=lambda p: p.centre_hz)
        out.append(ChannelModalCloudResult(
            channel_name=str(name), sample_rate_hz=int(sample_rate_hz),
            analysis_start_sample_index=int(dev["starts"][i]), analysis_length_samples=int(dev["lens"][i]),
            metric=dev["metric"], points=pts,
        ))
    return out


def modal_cloud_device(eng, batch, sample_rate_hz: int, settings: ModalCloudAnalysisSettings, fused: bool = True):
    """Device-resident modal cloud: per-(channel, log bin) fit records stay in HBM.  fused=False forces the two-kernel
    path (STFT matrix + ira_logbin_aggregate) where the fused kernel would apply (A/B, tests)."""
    starts, lens, nframes = select_stft_segments(eng, batch, sample_rate_hz, settings, "modal cloud")
    n_fft, hop = int(settings.n_fft), int(settings.hop_length)
    metric = str(settings.metric).lower()
    if metric == "t20":
        rng = settings.t20_range_db
    elif metric == "edt":
        rng = settings.edt_range_db
    else:
        metric, rng = "t30", settings.t30_range_db
    hi_db, lo_db = float(rng[0]), float(rng[1])
    if lo_db > hi_db:
        raise ValueError("range_db should be (higher_db, lower_db), e.g. (-5, -35).")

    freq = np.fft.rfftfreq(n_fft, d=1.0 / float(sample_rate_hz)).astype(np.float32)
    nyq = 0.5 * float(sample_rate_hz)
    f_lo = float(np.clip(settings.f_min_hz, 1.0, nyq))
    f_hi = float(np.clip(settings.f_max_hz, f_lo, nyq))
    rows = np.nonzero((freq >= f_lo) & (freq <= f_hi))[0]
    edges = _build_log_bins(f_lo, f_hi, int(settings.log_bins_per_octave), int(settings.min_bins))
    k_base = int(rows[0]) if rows.size else 0
    centres, first, count = log_bin_rows(freq[rows], edges)
    nbins = int(centres.size)

    # The STFT matrix is only an intermediate here.  n_fft 8192: one fused kernel (STFT + aggregation, the dB matrix is
    # never written); otherwise STFT (frame-major where the library has it) followed by ira_logbin_aggregate.
    rows_ok = nbins == 0 or int(k_base + (first + count).max()) <= n_fft // 2 + 1
    if eng.stft_logbin_ok(n_fft) and nbins > 0 and rows_ok and fused:
        cols = nframes.astype(np.int32)
        curves, cur_off = eng.stft_logbin(batch.x, batch.off + starts, nframes, n_fft, hop,
                                          bool(settings.use_hann_window), float(settings.floor_db), k_base, first, count)
    else:
        tf = eng.stft_frame_major_ok(n_fft, 64)
        mag, mag_off, cols = eng.stft_mag_db(batch.x, batch.off + starts, nframes, n_fft, hop,
                                             bool(settings.use_hann_window), float(settings.floor_db), 64,
                                             frame_major=tf)
        curves, cur_off = eng.logbin_aggregate(mag, mag_off, cols, k_base, first, count,
                                               frame_major_rows=(n_fft // 2 + 1) if tf else 0)
    cols64 = np.asarray(cols, dtype=np.int64)
    c_off = (np.asarray(cur_off, dtype=np.int64)[:, None] + np.arange(nbins, dtype=np.int64)[None, :] * cols64[:, None]).reshape(-1)
    c_len = np.repeat(cols64, nbins)
    fits, _ = eng.curve_fits(curves, c_off, c_len, float(hop), float(sample_rate_hz),
                             [(hi_db, max(lo_db, float(settings.fit_lower_limit_db)))], int(settings.min_fit_points),
                             rel_to_peak=True, floor_db=float(settings.floor_db),
                             min_peak_above_floor=float(settings.min_peak_db_above_floor))
    return dict(fits=fits, centres=centres, nbins=nbins, starts=starts, lens=lens, metric=metric, curves=curves,
                cols=cols)


def analyse_modal_cloud_for_channel(
    samples: np.ndarray,
    sample_rate_hz: int,
    channel_name: str,
    settings: ModalCloudAnalysisSettings,
) -> ChannelModalCloudResult:
    return analyse_modal_cloud_batch([samples], sample_rate_hz, [channel_name], settings)[0]


def analyse_modal_cloud_from_wav_file(
    input_wav_file_path: str | Path,
    settings: Optional[ModalCloudAnalysisSettings] = None,
) -> List[ChannelModalCloudResult]:
    settings = settings or ModalCloudAnalysisSettings()
    loaded, chans = wav_channels(input_wav_file_path, settings.use_mono_downmix_for_stereo)
    return analyse_modal_cloud_batch([c for _, c in chans], loaded.sample_rate_hz, [n for n, _ in chans], settings)


def plot_modal_cloud_from_wav_file(
    input_wav_file_path: str | Path,
    analysis_settings: Optional[ModalCloudAnalysisSettings] = None,
    plot_settings: Optional[ModalCloudPlotSettings] = None,
    output_basename: Optional[str | Path] = None,
    show_interactive: bool = True,
) -> List[ChannelModalCloudResult]:
    analysis_settings = analysis_settings or ModalCloudAnalysisSettings()
    plot_settings = plot_settings or ModalCloudPlotSettings()
    results = analyse_modal_cloud_from_wav_file(input_wav_file_path, analysis_settings)
    from . import plotting
    for r in results:
        plotting.render_modal_cloud(r, analysis_settings, plot_settings,
                                    f"Modal cloud — {input_wav_file_path} — {r.channel_name}",
                                    plotting.png_path(output_basename, f"_modalcloud_{r.channel_name}"),
                                    show_interactive)
    return results


def summarise_modal_cloud_results_text(results: List[ChannelModalCloudResult]) -> str:
    lines: List[str] = []
    for r in results:
        lines.append(
            f"[{r.channel_name}] metric={r.metric} start_sample={r.analysis_start_sample_index} "
            f"dur={float(r.analysis_length_samples) / float(r.sample_rate_hz):.3f}s points={len(r.points)}"
        )
        if r.points:
            rt = np.array([p.rt60_seconds for p in r.points], dtype=np.float64)
            lines.append(f"  rt60: median={np.median(rt):.3f}s  p90={np.percentile(rt, 90):.3f}s  max={np.max(rt):.3f}s")
    return "\n".join(lines)
