"""
Diffusion / decorrelation metrics over time on the GPU.

Host-side mirror of the reference's analyse/diffusion.py (dataclasses :43-83, helpers :92-136,
analyse_diffusion_for_channel :234-291, analyse_diffusion_from_wav_file :294-376, summary :457-476).  Device work:
one workgroup per 50 ms window (ira_diffusion): float32 mean removal and echo-density count reproduced bit for bit
(numpy's pairwise float32 mean), windowed autocorrelation peak over lags 1..max_lag in float64; for true stereo
files the zero-lag correlation and the IACC maximum over +-max_lag per window (ira_diffusion_stereo).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ..engine import get_engine
from ._common import wav_channels


@dataclass(frozen=True)
class DiffusionAnalysisSettings:
    use_mono_downmix_for_stereo: bool = False
    trim_to_peak: bool = True
    ignore_leading_seconds: float = 0.0
    window_seconds: float = 0.050
    hop_seconds: float = 0.010
    max_lag_milliseconds: float = 10.0
    echo_density_threshold_rms: float = 1.0
    echo_density_normalise_to_gaussian: bool = True


@dataclass(frozen=True)
class DiffusionTimeSeries:
    time_seconds: np.ndarray
    max_abs_autocorr: np.ndarray
    echo_density: np.ndarray
    corr0: Optional[np.ndarray] = None
    iacc_max: Optional[np.ndarray] = None


@dataclass(frozen=True)
class DiffusionChannelResult:
    channel_name: str
    sample_rate_hz: int
    series: DiffusionTimeSeries


def _frame_count(num_samples: int, win: int, hop: int) -> int:
    if num_samples < win:
        return 0
    return 1 + (num_samples - win) // hop


def _expected_gaussian_abs_exceedance(threshold_rms: float) -> float:
    """P(|x| > k sigma) of a Gaussian, via erf (reference diffusion.py:126-136)."""
    k = float(threshold_rms)
    phi = 0.5 * (1.0 + math.erf(k / np.sqrt(2.0)))
    return 2.0 * (1.0 - phi)


def window_geometry(sample_rate_hz: int, settings: DiffusionAnalysisSettings) -> Tuple[int, int, int]:
    """(win, hop, max_lag) in samples (reference diffusion.py:246-258)."""
    win = max(16, int(round(settings.window_seconds * float(sample_rate_hz))))
    hop = max(1, int(round(settings.hop_seconds * float(sample_rate_hz))))
    max_lag = max(1, int(round((settings.max_lag_milliseconds / 1000.0) * float(sample_rate_hz))))
    if win > 8192 or max_lag > 4096:
        raise ValueError("diffusion windows are limited to 8192 samples and lags to 4096 samples on the GPU path.")
    return win, hop, max_lag


def trim_start(num_samples: int, peak: int, sample_rate_hz: int, settings: DiffusionAnalysisSettings) -> int:
    """Start index of the analysed tail (reference _trim_and_ignore, diffusion.py:92-117)."""
    start = int(peak) if settings.trim_to_peak else 0
    remaining = num_samples - start
    if settings.ignore_leading_seconds > 0.0:
        ig = int(round(settings.ignore_leading_seconds * float(sample_rate_hz)))
        start += max(0, min(ig, remaining))
    return start


def _time_axis(frames: int, win: int, hop: int, sample_rate_hz: int) -> np.ndarray:
    return ((np.arange(frames, dtype=np.float64) * float(hop) + win * 0.5) / float(sample_rate_hz)).astype(np.float32)


def diffusion_device(eng, batch, sample_rate_hz: int, settings: DiffusionAnalysisSettings):
    """Per-channel series for a device-resident batch, left in HBM: dict(ac, ed, off, frames, win, hop)."""
    win, hop, max_lag = window_geometry(sample_rate_hz, settings)
    peaks = eng.peaks(batch) if settings.trim_to_peak else np.zeros(batch.count, dtype=np.int64)
    starts = np.array([trim_start(int(batch.length[i]), int(peaks[i]), sample_rate_hz, settings)
                       for i in range(batch.count)], dtype=np.int64)
    frames = np.array([_frame_count(int(batch.length[i] - starts[i]), win, hop) for i in range(batch.count)],
                      dtype=np.int32)
    if np.any(frames <= 0):
        raise ValueError("Not enough samples for diffusion analysis windows.")
    gauss = -1.0
    if settings.echo_density_normalise_to_gaussian:
        gauss = _expected_gaussian_abs_exceedance(settings.echo_density_threshold_rms)
    ac, ed, off = eng.diffusion(batch.x, batch.off + starts, frames, win, hop, max_lag,
                                float(settings.echo_density_threshold_rms), gauss)
    return dict(ac=ac, ed=ed, off=off, frames=frames, starts=starts, win=win, hop=hop, max_lag=max_lag)


def diffusion_results(dev, sample_rate_hz: int, channel_names: Sequence[str]) -> List[DiffusionChannelResult]:
    ac, ed = dev["ac"].cpu().numpy(), dev["ed"].cpu().numpy()
    out = []
    for i, name in enumerate(channel_names):
        o, f = int(dev["off"][i]), int(dev["frames"][i])
        out.append(DiffusionChannelResult(
            channel_name=name, sample_rate_hz=sample_rate_hz,
            series=DiffusionTimeSeries(time_seconds=_time_axis(f, dev["win"], dev["hop"], sample_rate_hz),
                                       max_abs_autocorr=ac[o : o + f].copy(), echo_density=ed[o : o + f].copy())))
    return out


def analyse_diffusion_batch(channels: Sequence[np.ndarray], sample_rate_hz: int, channel_names: Sequence[str],
                            settings: DiffusionAnalysisSettings) -> List[DiffusionChannelResult]:
    eng = get_engine()
    batch = eng.upload([c.astype(np.float32, copy=False) for c in channels])
    return diffusion_results(diffusion_device(eng, batch, sample_rate_hz, settings), sample_rate_hz, channel_names)


def analyse_diffusion_for_channel(samples: np.ndarray, sample_rate_hz: int, channel_name: str,
                                  settings: DiffusionAnalysisSettings) -> DiffusionChannelResult:
    return analyse_diffusion_batch([samples], sample_rate_hz, [channel_name], settings)[0]


def stereo_series(left: np.ndarray, right: np.ndarray, sample_rate_hz: int, settings: DiffusionAnalysisSettings
                  ) -> Tuple[np.ndarray, np.ndarray]:
    """corr0 and IACC max per window; both channels start at the peak of their float32 MEAN (diffusion.py:323-358)."""
    eng = get_engine()
    win, hop, max_lag = window_geometry(sample_rate_hz, settings)
    comb = ((left.astype(np.float64) + right.astype(np.float64)) * 0.5).astype(np.float32)
    peak = int(np.argmax(np.abs(comb.astype(np.float64)))) if settings.trim_to_peak else 0
    start = trim_start(int(comb.size), peak, sample_rate_hz, settings)
    frames = _frame_count(int(comb.size) - start, win, hop)
    if frames <= 0:
        return np.zeros(0, dtype=np.float32), np.zeros(0, dtype=np.float32)
    batch = eng.upload([left.astype(np.float32, copy=False), right.astype(np.float32, copy=False)])
    c0, ia, _ = eng.diffusion_stereo(batch.x, batch.off[:1] + start, batch.off[1:] + start,
                                     np.array([frames], dtype=np.int32), win, hop, max_lag)
    return c0.cpu().numpy()[:frames].copy(), ia.cpu().numpy()[:frames].copy()


def stereo_series_device(eng, split_batch, left_index: Sequence[int], mix_peaks: Sequence[int], sample_rate_hz: int,
                         settings: DiffusionAnalysisSettings) -> List[Tuple[np.ndarray, np.ndarray]]:
    """
    stereo_series for MANY files of one device batch: file j's left channel is split_batch channel left_index[j], its
    right channel the next one; mix_peaks[j] = argmax|0.5*(L+R)| of that file (reference diffusion.py:326-335).  One
    launch for all files.
    """
    win, hop, max_lag = window_geometry(sample_rate_hz, settings)
    li = np.asarray(left_index, dtype=np.int64)
    n = split_batch.length[li]
    starts = np.array([trim_start(int(n[j]), int(mix_peaks[j]) if settings.trim_to_peak else 0, sample_rate_hz, settings)
                       for j in range(li.size)], dtype=np.int64)
    frames = np.array([max(0, _frame_count(int(n[j] - starts[j]), win, hop)) for j in range(li.size)], dtype=np.int32)
    empty = (np.zeros(0, dtype=np.float32), np.zeros(0, dtype=np.float32))
    out: List[Tuple[np.ndarray, np.ndarray]] = [empty] * int(li.size)
    live = np.flatnonzero(frames > 0)
    if live.size:
        c0, ia, off = eng.diffusion_stereo(split_batch.x, split_batch.off[li[live]] + starts[live],
                                           split_batch.off[li[live] + 1] + starts[live], frames[live], win, hop, max_lag)
        c0, ia = c0.cpu().numpy(), ia.cpu().numpy()
        for k, j in enumerate(live):
            o, f = int(off[k]), int(frames[j])
            out[int(j)] = (c0[o : o + f].copy(), ia[o : o + f].copy())
    return out


def analyse_diffusion_from_wav_file(input_wav_file_path: str | Path,
                                    settings: Optional[DiffusionAnalysisSettings] = None
                                    ) -> List[DiffusionChannelResult]:
    settings = settings or DiffusionAnalysisSettings()
    loaded, chans = wav_channels(input_wav_file_path, settings.use_mono_downmix_for_stereo)
    results = analyse_diffusion_batch([c for _, c in chans], loaded.sample_rate_hz, [n for n, _ in chans], settings)
    if (not settings.use_mono_downmix_for_stereo) and len(chans) == 2:
        corr0, iacc = stereo_series(chans[0][1], chans[1][1], loaded.sample_rate_hz, settings)
        results = [DiffusionChannelResult(
            channel_name=r.channel_name, sample_rate_hz=r.sample_rate_hz,
            series=DiffusionTimeSeries(time_seconds=r.series.time_seconds, max_abs_autocorr=r.series.max_abs_autocorr,
                                       echo_density=r.series.echo_density, corr0=corr0, iacc_max=iacc))
            for r in results]
    return results


def plot_diffusion_from_wav_file(
    input_wav_file_path: str | Path,
    analysis_settings: Optional[DiffusionAnalysisSettings] = None,
    output_basename: Optional[str | Path] = None,
    show_interactive: bool = True,
) -> List[DiffusionChannelResult]:
    analysis_settings = analysis_settings or DiffusionAnalysisSettings()
    results = analyse_diffusion_from_wav_file(input_wav_file_path, analysis_settings)
    path = None
    if output_basename is not None:
        ob = Path(output_basename)
        path = ob.with_name(f"{ob.stem}_diffusion.png").with_suffix(".png")
    from . import plotting
    plotting.render_diffusion(results, f"Diffusion — {input_wav_file_path}", path, show_interactive)
    return results


def summarise_diffusion_results_text(results: List[DiffusionChannelResult]) -> str:
    lines: List[str] = []
    for r in results:
        lines.append(f"[{r.channel_name}]")
        lines.append(f"  median_max_abs_autocorr={float(np.nanmedian(r.series.max_abs_autocorr)):.3f}")
        lines.append(f"  median_echo_density={float(np.nanmedian(r.series.echo_density)):.3f}")
        if r.series.corr0 is not None and r.series.iacc_max is not None:
            lines.append(f"  median_corr0={float(np.nanmedian(r.series.corr0)):.3f}")
            lines.append(f"  median_iacc_max={float(np.nanmedian(r.series.iacc_max)):.3f}")
    return "\n".join(lines)
