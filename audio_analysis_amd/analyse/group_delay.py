"""
Group delay vs frequency on the GPU.

Host-side mirror of the reference's analyse/group_delay.py (dataclasses :39-73, _compute_group_delay_from_ir
:89-137, the time selection of plot_group_delay_from_wav_file :159-171, summary :209-220).  Device work per
channel: one zero-padded / truncated Hann-windowed rFFT of n_fft points (ira_rfft_any with data/window lengths;
channels that share n_fft ride one complex transform in pairs), atan2 (ira_spectrum_mag_phase), numpy.unwrap in
float64 (ira_phase_unwrap) and -numpy.gradient on the rad/sample axis (ira_group_delay).  The optional
moving-average smoothing (default off) and the frequency mask are host-side NumPy on the device result.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ..engine import get_engine
from ._common import wav_channels
from .frequency_response import rfft_bin_step


@dataclass(frozen=True)
class GroupDelayAnalysisSettings:
    use_mono_downmix_for_stereo: bool = False
    trim_to_peak: bool = True
    ignore_leading_seconds: float = 0.0
    analysis_duration_seconds: Optional[float] = None
    use_hann_window: bool = True
    fft_size: Optional[int] = None        # None: next power of two >= segment length, capped at 2^20
    f_min_hz: float = 20.0
    f_max_hz: float = 20000.0
    unwrap_phase: bool = True
    smoothing_bins: int = 0


@dataclass(frozen=True)
class GroupDelayPlotSettings:
    secondary_channel_alpha: float = 0.7
    ylim_samples: Optional[Tuple[float, float]] = None
    show_zero_line: bool = True


@dataclass(frozen=True)
class ChannelGroupDelayResult:
    channel_name: str
    sample_rate_hz: int
    frequency_hz: np.ndarray
    group_delay_samples: np.ndarray


def _next_pow2(n: int) -> int:
    n = int(max(1, n))
    return 1 << (n - 1).bit_length()


def _moving_average(x: np.ndarray, window: int) -> np.ndarray:
    if window <= 1:
        return x
    window = int(window)
    return np.convolve(x, np.ones(window, dtype=np.float64) / float(window), mode="same")


def group_delay_segment(num_samples: int, peak: int, sample_rate_hz: int,
                        settings: GroupDelayAnalysisSettings) -> Tuple[int, int]:
    """(start, length) as the reference selects it (group_delay.py:159-171): unlike the other modules the ignore time
    is added to the start before clipping, and a duration always keeps at least one sample."""
    start = int(peak) if settings.trim_to_peak else 0
    start += int(round(float(settings.ignore_leading_seconds) * sample_rate_hz))
    start = max(0, min(start, num_samples))
    if settings.analysis_duration_seconds is None:
        return start, num_samples - start
    want = max(1, int(round(float(settings.analysis_duration_seconds) * sample_rate_hz)))
    return start, max(0, min(want, num_samples - start))


def fft_size_for(segment_length: int, settings: GroupDelayAnalysisSettings) -> int:
    if settings.fft_size is not None:
        return int(settings.fft_size)
    return min(_next_pow2(segment_length), 1 << 20)


def group_delay_device(eng, batch, sample_rate_hz: int, settings: GroupDelayAnalysisSettings,
                       starts: Optional[np.ndarray] = None, lens: Optional[np.ndarray] = None):
    """Device-resident group delay of every channel: float64, n_fft/2+1 values per channel at `off` (unmasked,
    unsmoothed).  starts/lens override the time selection (single-segment entry)."""
    n = batch.count
    if starts is None:
        peaks = eng.peaks(batch) if settings.trim_to_peak else np.zeros(n, dtype=np.int64)
        starts = np.empty(n, dtype=np.int64)
        lens = np.empty(n, dtype=np.int64)
        for i in range(n):
            starts[i], lens[i] = group_delay_segment(int(batch.length[i]), int(peaks[i]), sample_rate_hz, settings)
    if np.any(lens < 1):
        raise ValueError("Not enough samples after trimming/selection to analyse group delay.")
    n_fft = np.array([fft_size_for(int(v), settings) for v in lens], dtype=np.int64)
    if np.any(n_fft < 2) or np.any(n_fft > (1 << 21)):
        raise ValueError("group delay needs 2 <= fft_size <= 2097152.")
    tag, eng.event_tag = eng.event_tag, "[gd]"             # per-call device times of this block (bench.py) under their own names
    try:
        spec, off = eng.rfft_any(batch.x, batch.off + starts, n_fft.astype(np.int32), bool(settings.use_hann_window),
                                 data_len=np.minimum(lens, n_fft).astype(np.int32), win_len=lens.astype(np.int32))
        _, ph = eng.spectrum_mag_phase(spec, off, n_fft.astype(np.int32), -400.0, want_phase=True)
        ph64 = eng.phase_unwrap(ph, off, n_fft.astype(np.int32), bool(settings.unwrap_phase), False, as_float64=True)
        steps = np.array([rfft_bin_step(int(v), sample_rate_hz) for v in n_fft], dtype=np.float64)
        gd = eng.group_delay(ph64, off, n_fft, steps, float(sample_rate_hz))
    finally:
        eng.event_tag = tag
    return dict(gd=gd, off=off, n_fft=n_fft, starts=starts, lens=lens)


_MASK_CACHE: dict = {}


def mask_range(n_fft: int, sample_rate_hz: int, f_min_hz: float, f_max_hz: float) -> Tuple[int, int]:
    """(first bin, count) of the contiguous run selected by (freq >= f_min) & (freq <= f_max) on rfftfreq(n_fft)."""
    key = (int(n_fft), int(sample_rate_hz), float(f_min_hz), float(f_max_hz))
    if key not in _MASK_CACHE:
        freq = np.fft.rfftfreq(int(n_fft), d=1.0 / float(sample_rate_hz))
        idx = np.nonzero((freq >= float(f_min_hz)) & (freq <= float(f_max_hz)))[0]
        _MASK_CACHE[key] = (int(idx[0]), int(idx.size)) if idx.size else (0, 0)
    return _MASK_CACHE[key]


def _lerp(a: float, b: float, t: float) -> float:
    """numpy's interpolation between neighbouring order statistics (numpy/lib/_function_base_impl.py, _lerp)."""
    d = np.float64(b) - np.float64(a)
    if t >= 0.5:
        return float(np.float64(b) - d * (1.0 - t))
    return float(np.float64(a) + d * t)


def quantile_ranks(m: int) -> Tuple[np.ndarray, np.ndarray]:
    """Ranks (6,) and interpolation weights for [median lo, median hi, p10 lo, p10 hi, p90 lo, p90 hi] of m values,
    following numpy.median (mean of the two middle values) and numpy.percentile(method='linear')."""
    ranks = np.zeros(6, dtype=np.int64)
    gam = np.zeros(3, dtype=np.float64)
    if m > 0:
        ranks[0], ranks[1] = (m - 1) // 2, m // 2
        for j, pct in enumerate((10, 90)):
            virt = (m - 1) * np.true_divide(pct, 100)
            prev = np.floor(virt)
            ranks[2 + 2 * j] = int(prev)
            ranks[3 + 2 * j] = min(int(prev) + 1, m - 1)
            gam[1 + j] = virt - prev
    return ranks, gam


def summary_statistics_device(eng, dev, sample_rate_hz: int, settings: GroupDelayAnalysisSettings):
    """(median, p10, p90) of the masked group delay per channel WITHOUT bringing the curves to the host: the six order
    statistics come from ira_order_stats; returns (HostFuture of (n, 6) values, gammas (n, 3), counts (n,))."""
    if settings.smoothing_bins and settings.smoothing_bins > 1:
        raise ValueError("device-side group-delay statistics do not cover the optional smoothing; use the host path.")
    n = int(dev["n_fft"].size)
    off = np.empty(n, dtype=np.int64); cnt = np.empty(n, dtype=np.int32)
    ranks = np.zeros((n, 6), dtype=np.int64); gam = np.zeros((n, 3), dtype=np.float64)
    for i in range(n):
        k0, m = mask_range(int(dev["n_fft"][i]), sample_rate_hz, settings.f_min_hz, settings.f_max_hz)
        off[i], cnt[i] = int(dev["off"][i]) + k0, m
        ranks[i], gam[i] = quantile_ranks(m)
    stats = eng.order_stats(dev["gd"], off, cnt, ranks)
    return eng.fetch(stats), gam, cnt


def finish_summary_statistics(values: np.ndarray, gam: np.ndarray, cnt: np.ndarray) -> np.ndarray:
    """(n, 3) float64 [median, p10, p90] from the order statistics; NaN rows where the mask selected nothing."""
    out = np.full((values.shape[0], 3), np.nan)
    for i in range(values.shape[0]):
        if cnt[i] <= 0:
            continue
        v = values[i]
        out[i, 0] = float(v[0]) if cnt[i] % 2 == 1 else float(np.mean(np.array([v[0], v[1]])))
        out[i, 1] = _lerp(v[2], v[3], gam[i, 1])
        out[i, 2] = _lerp(v[4], v[5], gam[i, 2])
    return out


def group_delay_summary_lines(eng, dev, sample_rate_hz: int, channel_names: Sequence[str],
                              settings: GroupDelayAnalysisSettings) -> List[Optional[str]]:
    """One summarise_group_delay_results_text line per channel (None where the mask selects nothing) from device-side
    order statistics (ira_order_stats); the curves stay in HBM.  (The optional moving-average smoothing is a host step:
    the full path is used.)"""
    if settings.smoothing_bins and settings.smoothing_bins > 1:
        out = []
        for r in group_delay_results(dev, sample_rate_hz, channel_names, settings):
            text = summarise_group_delay_results_text([r])
            out.append(text.split("\n", 1)[1] if text.startswith("Group delay summary:") else None)
        return out
    fut, gam, cnt = summary_statistics_device(eng, dev, sample_rate_hz, settings)
    eng.sync()
    st = finish_summary_statistics(fut.get(), gam, cnt)
    return [f"- {name}: gd median={st[i, 0]:.3f} samples, p10={st[i, 1]:.3f}, p90={st[i, 2]:.3f}" if cnt[i] > 0 else None
            for i, name in enumerate(channel_names)]


def join_group_delay_summary(lines: Sequence[Optional[str]]) -> str:
    kept = [ln for ln in lines if ln]
    return "Group delay summary:\n" + "\n".join(kept) if kept else "No group delay results."


def group_delay_results(dev, sample_rate_hz: int, channel_names: Sequence[str],
                        settings: GroupDelayAnalysisSettings, gd_host: Optional[np.ndarray] = None):
    host = dev["gd"].cpu().numpy() if gd_host is None else gd_host
    out: List[ChannelGroupDelayResult] = []
    for i, name in enumerate(channel_names):
        n_fft = int(dev["n_fft"][i])
        gd = host[int(dev["off"][i]) : int(dev["off"][i]) + n_fft // 2 + 1].astype(np.float64, copy=True)
        freq = np.fft.rfftfreq(n_fft, d=1.0 / float(sample_rate_hz))
        if settings.smoothing_bins and settings.smoothing_bins > 1:
            gd = _moving_average(gd, int(settings.smoothing_bins))
        mask = (freq >= float(settings.f_min_hz)) & (freq <= float(settings.f_max_hz))
        out.append(ChannelGroupDelayResult(channel_name=name, sample_rate_hz=sample_rate_hz,
                                           frequency_hz=freq[mask].astype(np.float64, copy=False),
                                           group_delay_samples=gd[mask].astype(np.float64, copy=False)))
    return out


def analyse_group_delay_batch(channels: Sequence[np.ndarray], sample_rate_hz: int, channel_names: Sequence[str],
                              settings: GroupDelayAnalysisSettings) -> List[ChannelGroupDelayResult]:
    for c in channels:
        if c.ndim != 1:
            raise ValueError("group delay expects 1D mono arrays.")
    eng = get_engine()
    batch = eng.upload([c.astype(np.float32, copy=False) for c in channels])
    return group_delay_results(group_delay_device(eng, batch, sample_rate_hz, settings), sample_rate_hz,
                               channel_names, settings)


def _compute_group_delay_from_ir(samples: np.ndarray, sample_rate_hz: int,
                                 settings: GroupDelayAnalysisSettings) -> ChannelGroupDelayResult:
    """Same contract as the reference helper (group_delay.py:89-137): `samples` is the already selected segment."""
    assert samples.ndim == 1
    eng = get_engine()
    seg = samples.astype(np.float32, copy=False)
    batch = eng.upload([seg])
    dev = group_delay_device(eng, batch, sample_rate_hz, settings, starts=np.zeros(1, dtype=np.int64),
                             lens=np.array([seg.size], dtype=np.int64))
    return group_delay_results(dev, sample_rate_hz, [""], settings)[0]


def analyse_group_delay_from_wav_file(input_wav_file_path: str | Path,
                                      settings: Optional[GroupDelayAnalysisSettings] = None
                                      ) -> List[ChannelGroupDelayResult]:
    settings = settings or GroupDelayAnalysisSettings()
    loaded, chans = wav_channels(input_wav_file_path, settings.use_mono_downmix_for_stereo)
    return analyse_group_delay_batch([c for _, c in chans], loaded.sample_rate_hz, [n for n, _ in chans], settings)


def plot_group_delay_from_wav_file(
    input_wav_file_path: str,
    settings: GroupDelayAnalysisSettings,
    plot_settings: GroupDelayPlotSettings,
    output_basename: Optional[str | Path] = None,
    show_interactive: bool = True,
) -> List[ChannelGroupDelayResult]:
    results = analyse_group_delay_from_wav_file(input_wav_file_path, settings)
    from . import plotting
    for r in results:
        path = None
        if output_basename is not None:
            path = str(Path(output_basename).with_suffix("")) + f"_groupdelay_{r.channel_name}.png"
        plotting.render_group_delay(r, settings, plot_settings, f"Group delay ({r.channel_name})", path,
                                    show_interactive)
    return results


def summarise_group_delay_results_text(results: List[ChannelGroupDelayResult]) -> str:
    lines: List[str] = []
    for r in results:
        gd = r.group_delay_samples
        if gd.size == 0:
            continue
        lines.append(f"- {r.channel_name}: gd median={float(np.median(gd)):.3f} samples, "
                     f"p10={float(np.percentile(gd, 10)):.3f}, p90={float(np.percentile(gd, 90)):.3f}")
    if not lines:
        return "No group delay results."
    return "Group delay summary:\n" + "\n".join(lines)
