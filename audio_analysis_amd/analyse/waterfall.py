"""
Waterfall (cumulative-spectral-decay style slices) on the GPU.

Host-side mirror of the reference's analyse/waterfall.py (dataclasses :43-110, slice selection :233-286,
relative-dB slices :289-341, analyse_waterfall_for_channel :349-410, summary :615-623).
Only the <= num_slices selected frames are transformed (the reference computes the whole STFT at :378-385
and then indexes a handful of columns at :309 -- frames are independent, so the selected columns are the
same); they use float64 butterflies since there are so few of them.  Frame selection is integer index logic
and stays on the host; normalisation/clipping is ira_waterfall_rel.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ..engine import get_engine
from ._common import frame_time_axis, wav_channels
from .frequency_response import smooth_log_frequency
from .spectrogram import select_stft_segments


@dataclass(frozen=True)
class WaterfallAnalysisSettings:
    use_mono_downmix_for_stereo: bool = False
    trim_to_peak: bool = True
    ignore_leading_seconds: float = 0.0
    analysis_duration_seconds: Optional[float] = None
    n_fft: int = 4096
    hop_length: int = 512
    use_hann_window: bool = True
    f_min_hz: float = 20.0
    f_max_hz: float = 20000.0
    slice_mode: str = "auto"
    num_slices: int = 18
    slice_spacing_seconds: float = 0.05
    start_time_seconds: float = 0.0
    end_time_seconds: Optional[float] = None
    db_reference: str = "global_max"
    smoothing_log_bins: int = 0
    log_bins_per_octave: int = 96
    dynamic_range_db: float = 80.0
    floor_db: float = -120.0


@dataclass(frozen=True)
class WaterfallPlotSettings:
    style: str = "3d"
    secondary_channel_alpha: float = 0.7
    elev_deg: float = 30.0
    azim_deg: float = -60.0
    ridge_offset_db: float = 6.0
    zlim_db: Optional[Tuple[float, float]] = None


@dataclass(frozen=True)
class ChannelWaterfallResult:
    channel_name: str
    sample_rate_hz: int
    analysis_start_sample_index: int
    analysis_length_samples: int
    slice_times_seconds: np.ndarray
    frequency_hz: np.ndarray
    slice_magnitude_rel_db: np.ndarray


def _select_slice_frame_indices(frame_times_seconds: np.ndarray, settings: WaterfallAnalysisSettings) -> np.ndarray:
    """Ordered unique STFT frame indices for the slices (integer logic, bit-exact with the reference)."""
    ft = frame_times_seconds
    if ft.size == 0:
        return np.zeros((0,), dtype=np.int32)
    t_start = float(max(0.0, settings.start_time_seconds))
    t_end = float(ft[-1]) if settings.end_time_seconds is None else float(settings.end_time_seconds)
    if t_end <= t_start:
        t_end = float(ft[-1])
    window = np.nonzero((ft >= t_start) & (ft <= t_end))[0]
    if window.size == 0:
        return np.zeros((0,), dtype=np.int32)
    first, last = int(window[0]), int(window[-1])
    mode = str(settings.slice_mode).lower()
    if mode == "uniform_frames":
        return np.unique(np.linspace(first, last, int(max(1, settings.num_slices)), dtype=np.int32))
    if mode == "uniform_time":
        wanted = np.arange(t_start, t_end + 1e-9, float(max(1e-4, settings.slice_spacing_seconds)), dtype=np.float64)
        if_empty = [first, last]
    else:
        wanted = np.linspace(t_start, t_end, int(max(2, settings.num_slices)), dtype=np.float64)
        if_empty = []
    nearest = [int(np.argmin(np.abs(ft - float(w)))) for w in wanted]      # float32 distance, first minimum
    kept = [j for j in nearest if first <= j <= last]
    return np.unique(np.array(kept if kept else if_empty, dtype=np.int32))


def analyse_waterfall_batch(
    channels: Sequence[np.ndarray],
    sample_rate_hz: int,
    channel_names: Sequence[str],
    settings: WaterfallAnalysisSettings,
) -> List[ChannelWaterfallResult]:
    eng = get_engine()
    batch = eng.upload(list(channels))
    return waterfall_results(waterfall_device(eng, batch, sample_rate_hz, settings), sample_rate_hz, channel_names,
                             settings)


def waterfall_results(dev, sample_rate_hz: int, channel_names, settings) -> List[ChannelWaterfallResult]:
    rel_host = dev["rel"].cpu().numpy()
    out = []
    for i, name in enumerate(channel_names):
        s, nsel = int(dev["cols"][i]), dev["nsel"]
        ft = frame_time_axis(int(dev["nframes"][i]), int(settings.hop_length), sample_rate_hz)
        o = int(dev["rel_off"][i])
        out.append(ChannelWaterfallResult(
            channel_name=str(name), sample_rate_hz=int(sample_rate_hz),
            analysis_start_sample_index=int(dev["starts"][i]), analysis_length_samples=int(dev["lens"][i]),
            slice_times_seconds=ft[dev["picks"][i]].astype(np.float32), frequency_hz=dev["f_sel"],
            slice_magnitude_rel_db=rel_host[o : o + s * nsel].reshape(s, nsel).copy(),
        ))
    return out


def waterfall_summary_lines(dev, sample_rate_hz: int, channel_names) -> List[str]:
    """One summarise_waterfall_results_text line per channel from the batch geometry alone; the slice blocks stay in HBM."""
    return list(
        f"[{name}] start_sample={int(dev['starts'][i])}  "
        f"dur={float(int(dev['lens'][i])) / float(int(sample_rate_hz)):.3f}s  "
        f"slices={int(np.asarray(dev['picks'][i]).size)}  f_bins={int(np.asarray(dev['f_sel']).size)}"
        for i, name in enumerate(channel_names))


def waterfall_device(eng, batch, sample_rate_hz: int, settings: WaterfallAnalysisSettings):
    """Device-resident waterfall: (S_i, nsel) relative-dB slice blocks in one flat float32 buffer."""
    starts, lens, nframes = select_stft_segments(eng, batch, sample_rate_hz, settings, "waterfall")
    n_fft, hop = int(settings.n_fft), int(settings.hop_length)
    picks = []
    by_count = {}                      # the selection depends only on the frame count (and the settings)
    for i in range(batch.count):
        tcount = int(nframes[i])
        if tcount not in by_count:
            idx = _select_slice_frame_indices(frame_time_axis(tcount, hop, sample_rate_hz), settings)
            if idx.size < 2:
                raise ValueError("Not enough slices selected for waterfall (increase duration or num_slices).")
            by_count[tcount] = idx.astype(np.int32)
        picks.append(by_count[tcount])
    freq = np.fft.rfftfreq(n_fft, d=1.0 / float(sample_rate_hz)).astype(np.float32)
    nyq = float(freq[-1]) if freq.size else 0.0
    f_lo = float(np.clip(settings.f_min_hz, 1.0, nyq))
    f_hi = float(np.clip(settings.f_max_hz, f_lo, nyq))
    rows = np.nonzero((freq >= f_lo) & (freq <= f_hi))[0]
    if rows.size == 0:
        raise ValueError("Waterfall frequency selection is empty (check f_min_hz/f_max_hz).")
    k_lo, nsel = int(rows[0]), int(rows.size)
    f_sel = freq[rows].astype(np.float32)

    mag, mag_off, cols = eng.stft_mag_db(batch.x, batch.off + starts, nframes, n_fft, hop,
                                         bool(settings.use_hann_window), float(settings.floor_db), 64, frame_sel=picks)
    smooth = bool(settings.smoothing_log_bins and int(settings.smoothing_log_bins) > 1)
    if smooth:
        # optional, default-off: per-slice log-frequency smoothing (waterfall.py:140-185, float32 round trip around the
        # convolution) on the device, in place: one curve per (channel, slice) = a column of the (F, S) matrix
        f_all = n_fft // 2 + 1
        step = 1.0 / (n_fft * (1.0 / float(sample_rate_hz)))
        c_off = np.concatenate([mag_off[i] + np.arange(int(cols[i]), dtype=np.int64) for i in range(batch.count)])
        c_stride = np.concatenate([np.full(int(cols[i]), int(cols[i]), dtype=np.int32) for i in range(batch.count)])
        ncurves = int(c_off.size)
        done = eng.log_smooth(mag, c_off, c_stride, np.full(ncurves, k_lo, np.int32), np.full(ncurves, nsel, np.int32),
                              np.full(ncurves, step, np.float64), int(settings.smoothing_log_bins),
                              int(settings.log_bins_per_octave), through_float32=True)
        if not done:                                              # grid beyond the kernel's LDS budget: host restatement
            host = mag.cpu().numpy().copy()
            for i in range(batch.count):
                s = int(cols[i])
                block = host[mag_off[i] : mag_off[i] + f_all * s].reshape(f_all, s)
                for j in range(s):
                    block[rows, j] = smooth_log_frequency(f_sel, block[rows, j].astype(np.float32), f_lo, f_hi,
                                                          int(settings.smoothing_log_bins),
                                                          int(settings.log_bins_per_octave), through_float32=True)
            mag = eng.to_dev(host)
    dyn = float(max(10.0, settings.dynamic_range_db))
    rel, rel_off = eng.waterfall_rel(mag, mag_off, cols, k_lo, nsel,
                                     str(settings.db_reference).lower() == "slice_max", dyn)
    return dict(rel=rel, rel_off=rel_off, cols=cols, picks=picks, starts=starts, lens=lens, nframes=nframes,
                f_sel=f_sel, nsel=nsel)


def analyse_waterfall_for_channel(
    samples: np.ndarray,
    sample_rate_hz: int,
    channel_name: str,
    settings: WaterfallAnalysisSettings,
) -> ChannelWaterfallResult:
    return analyse_waterfall_batch([samples], sample_rate_hz, [channel_name], settings)[0]


def analyse_waterfall_from_wav_file(
    input_wav_file_path: str | Path,
    settings: Optional[WaterfallAnalysisSettings] = None,
) -> List[ChannelWaterfallResult]:
    settings = settings or WaterfallAnalysisSettings()
    loaded, chans = wav_channels(input_wav_file_path, settings.use_mono_downmix_for_stereo)
    return analyse_waterfall_batch([c for _, c in chans], loaded.sample_rate_hz, [n for n, _ in chans], settings)


def plot_waterfall_from_wav_file(
    input_wav_file_path: str | Path,
    analysis_settings: Optional[WaterfallAnalysisSettings] = None,
    plot_settings: Optional[WaterfallPlotSettings] = None,
    output_basename: Optional[str | Path] = None,
    show_interactive: bool = True,
) -> List[ChannelWaterfallResult]:
    analysis_settings = analysis_settings or WaterfallAnalysisSettings()
    plot_settings = plot_settings or WaterfallPlotSettings()
    results = analyse_waterfall_from_wav_file(input_wav_file_path, analysis_settings)
    from . import plotting
    for r in results:
        plotting.render_waterfall(r, analysis_settings, plot_settings,
                                  f"Waterfall — {input_wav_file_path} — {r.channel_name}",
                                  plotting.png_path(output_basename, f"_waterfall_{r.channel_name}"), show_interactive)
    return results


def summarise_waterfall_results_text(results: List[ChannelWaterfallResult]) -> str:
    return "\n".join(
        f"[{r.channel_name}] start_sample={r.analysis_start_sample_index}  "
        f"dur={float(r.analysis_length_samples) / float(r.sample_rate_hz):.3f}s  "
        f"slices={int(r.slice_times_seconds.size)}  f_bins={int(r.frequency_hz.size)}"
        for r in results
    )
