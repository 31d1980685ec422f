"""
Schroeder decay analysis (EDC + EDT/T20/T30 line fits) on the GPU.

Host-side mirror of the reference's analyse/decay.py: same settings/result dataclasses, function names,
argument meaning and ValueError behaviour (decay.py:44-100, 115-170, 202-260, 268-365, 502-542); the
numeric bodies run in libira.so (ira_peak_index, ira_edc_db, ira_curve_fits).
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from ..engine import get_engine
from ._common import segment_bounds, segment_bounds_batch, wav_channels


@dataclass(frozen=True)
class DecayAnalysisSettings:
    use_mono_downmix_for_stereo: bool = False
    trim_to_peak: bool = True
    ignore_leading_seconds: float = 0.0
    edc_floor_db: float = -120.0
    edc_epsilon: float = 1e-20
    fit_lower_limit_db: float = -80.0
    t20_range_db: Tuple[float, float] = (-5.0, -25.0)
    t30_range_db: Tuple[float, float] = (-5.0, -35.0)
    compute_edt: bool = False
    edt_range_db: Tuple[float, float] = (0.0, -10.0)
    edc_smoothing_window_samples: int = 0


@dataclass(frozen=True)
class LinearDecayFit:
    name: str
    range_db: Tuple[float, float]
    start_time_seconds: float
    end_time_seconds: float
    slope_db_per_second: float
    intercept_db: float
    r_squared: float
    rt60_seconds: float


@dataclass(frozen=True)
class ChannelDecayAnalysis:
    channel_name: str
    sample_rate_hz: int
    analysis_start_sample_index: int
    time_seconds: np.ndarray
    edc_db: np.ndarray
    early_decay_10db_time_seconds: Optional[float]
    fits: Dict[str, LinearDecayFit]


@dataclass(frozen=True)
class DecayPlotSettings:
    show_fit_lines: bool = True
    secondary_channel_alpha: float = 0.7
    ylim_db: Tuple[float, float] = (-120.0, 5.0)


# ---------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------


def _check_range(range_db) -> Tuple[float, float]:
    hi, lo = float(range_db[0]), float(range_db[1])
    if lo > hi:
        raise ValueError("range_db should be (higher_db, lower_db), e.g. (-5, -25).")
    return hi, lo


def _fit_from_record(rec: np.ndarray, name: str, range_db: Tuple[float, float]) -> Optional[LinearDecayFit]:
    """rec = [valid, start_t, end_t, slope, intercept, r2, rt60, npts] from ira_curve_fits."""
    if rec[0] != 1.0:
        return None
    return LinearDecayFit(
        name=name, range_db=(float(range_db[0]), float(range_db[1])), start_time_seconds=float(rec[1]),
        end_time_seconds=float(rec[2]), slope_db_per_second=float(rec[3]), intercept_db=float(rec[4]),
        r_squared=float(rec[5]), rt60_seconds=float(rec[6]),
    )


def _time_axis(n: int, sample_rate_hz: int) -> np.ndarray:
    return (np.arange(n, dtype=np.float32) / float(sample_rate_hz)).astype(np.float32)


def _decay_bounds(eng, batch, sample_rate_hz: int, settings: DecayAnalysisSettings):
    """Time selection of decay.py:135-147 for every channel of the batch -> (starts, lens)."""
    n = batch.length
    peaks = eng.peaks(batch) if settings.trim_to_peak else np.zeros(batch.count, dtype=np.int64)
    starts, lens = segment_bounds_batch(n, peaks, sample_rate_hz, settings.trim_to_peak, settings.ignore_leading_seconds, None)
    if np.any(lens < 4):
        raise ValueError("Not enough samples after trimming/ignoring to compute EDC.")
    return starts, lens


def _edc_on_device(eng, batch, sample_rate_hz: int, settings: DecayAnalysisSettings):
    """Shared body: time selection + EDC kernel (+ optional host smoothing).  Returns device curve + bounds."""
    starts, lens = _decay_bounds(eng, batch, sample_rate_hz, settings)
    smooth = int(settings.edc_smoothing_window_samples or 0)
    if smooth > 1:
        # Optional, default-off: box smoothing of the unfloored f64 dB curve (decay.py:161-164), then floor + float32
        # cast -- on the device (ira_edc_box_smooth); the fits then read the smoothed curve (ira_curve_fits).
        if np.any(lens < smooth):
            raise ValueError("edc_smoothing_window_samples exceeds the analysed length")
        _, edc_off, raw64 = eng.edc_db(batch.x, batch.off + starts, lens, settings.edc_epsilon,
                                       settings.edc_floor_db, want_f64=True)
        edc = eng.edc_box_smooth(raw64, edc_off, lens, smooth, settings.edc_floor_db)
    else:
        edc, edc_off = eng.edc_db(batch.x, batch.off + starts, lens, settings.edc_epsilon, settings.edc_floor_db)
    return edc, edc_off, starts, lens


# ---------------------------------------------------------------------------------------------------
# public primitives (imported by rt60bands in the reference, decay.py:115-119 / :202-208)
# ---------------------------------------------------------------------------------------------------


def compute_schroeder_edc_db(
    samples: np.ndarray,
    sample_rate_hz: int,
    settings: DecayAnalysisSettings,
) -> Tuple[np.ndarray, np.ndarray, int]:
    """Returns (time_seconds f32, edc_db f32, analysis_start_sample_index)."""
    if samples.ndim != 1:
        raise ValueError("compute_schroeder_edc_db expects a 1D mono array.")
    eng = get_engine()
    batch = eng.upload([samples])
    edc, _, starts, lens = _edc_on_device(eng, batch, sample_rate_hz, settings)
    edc_host = edc[: int(lens[0])].cpu().numpy().copy()
    return _time_axis(int(lens[0]), sample_rate_hz), edc_host, int(starts[0])


def fit_decay_slope_over_db_range(
    time_seconds: np.ndarray,
    edc_db: np.ndarray,
    range_db: Tuple[float, float],
    fit_lower_limit_db: float,
    fit_name: str,
) -> Optional[LinearDecayFit]:
    """Line fit of an arbitrary (time, dB) curve between two dB levels; None if the range is unavailable."""
    hi, lo = _check_range(range_db)
    eng = get_engine()
    y = eng.to_dev(np.ascontiguousarray(edc_db, dtype=np.float32))
    t = eng.to_dev(np.ascontiguousarray(time_seconds, dtype=np.float32))
    n = int(edc_db.size)
    fits, _ = eng.curve_fits(y, np.zeros(1, np.int64), np.array([n], np.int64), 1.0, 1.0,
                             [(hi, max(lo, float(fit_lower_limit_db)))], 8, t_axis_dev=t)
    return _fit_from_record(fits.cpu().numpy()[0, 0], fit_name, (hi, lo))


# ---------------------------------------------------------------------------------------------------
# analysis entry points
# ---------------------------------------------------------------------------------------------------


def decay_fit_specs(settings: DecayAnalysisSettings):
    """[(name, range_db)] in the order the fits are computed, plus the effective (hi, lo) pairs."""
    specs = ([("EDT", settings.edt_range_db)] if settings.compute_edt else []) + [
        ("T20", settings.t20_range_db), ("T30", settings.t30_range_db)]
    ranges = []
    for _, rng in specs:
        hi, lo = _check_range(rng)
        ranges.append((hi, max(lo, float(settings.fit_lower_limit_db))))
    return specs, ranges


def decay_device(eng, batch, sample_rate_hz: int, settings: DecayAnalysisSettings):
    """Device-resident decay analysis of a batch: EDC curves + fit/crossing records stay in HBM."""
    specs, ranges = decay_fit_specs(settings)
    if int(settings.edc_smoothing_window_samples or 0) > 1:
        # optional dB smoothing changes the curve the fits see: materialise it, then fit the curve (ira_curve_fits)
        edc, edc_off, starts, lens = _edc_on_device(eng, batch, sample_rate_hz, settings)
        fits_dev, cross_dev = eng.curve_fits(edc, edc_off, lens, 1.0, float(sample_rate_hz), ranges, 8,
                                             cross=(0.0, -10.0))
    else:
        # fused path (ira_edc_fits): crossings and fits come straight from the samples; the curve is written once for
        # the caller and not read back
        starts, lens = _decay_bounds(eng, batch, sample_rate_hz, settings)
        fits_dev, cross_dev, edc, edc_off = eng.edc_fits(batch.x, batch.off + starts, lens, settings.edc_epsilon,
                                                         settings.edc_floor_db, 1.0, float(sample_rate_hz), ranges, 8,
                                                         cross=(0.0, -10.0), want_edc=True)
    return dict(edc=edc, edc_off=edc_off, starts=starts, lens=lens, fits=fits_dev, cross=cross_dev, specs=specs)


def decay_records_to_results(dev, fits: np.ndarray, cross: np.ndarray, edc_host, sample_rate_hz, channel_names,
                             with_time_axis: bool = True):
    """Result dataclasses from the small device records; edc_host=None / with_time_axis=False leave the two per-sample
    arrays out (summary-only callers: summarise_decay_results_text reads neither)."""
    out: List[ChannelDecayAnalysis] = []
    for i, name in enumerate(channel_names):
        t0, t10 = cross[i, 0], cross[i, 1]
        early = float(t10 - t0) if (not np.isnan(t0) and not np.isnan(t10) and t10 >= t0) else None
        fd: Dict[str, LinearDecayFit] = {}
        for j, (fname, rng) in enumerate(dev["specs"]):
            f = _fit_from_record(fits[i, j], fname, rng)
            if f is not None:
                fd[fname] = f
        ln = int(dev["lens"][i])
        o = int(dev["edc_off"][i])
        out.append(ChannelDecayAnalysis(
            channel_name=name, sample_rate_hz=sample_rate_hz, analysis_start_sample_index=int(dev["starts"][i]),
            time_seconds=_time_axis(ln, sample_rate_hz) if with_time_axis else None,
            edc_db=edc_host[o : o + ln].copy() if edc_host is not None else None,
            early_decay_10db_time_seconds=early, fits=fd,
        ))
    return out


def decay_results_without_curves(dev, sample_rate_hz: int, channel_names) -> List[ChannelDecayAnalysis]:
    """Result records of a device batch from its fit / crossing records alone (edc_db and time_seconds are None; no EDC
    curve leaves HBM): everything summarise_decay_results_text reads."""
    return decay_records_to_results(dev, dev["fits"].cpu().numpy(), dev["cross"].cpu().numpy(), None, sample_rate_hz,
                                    channel_names, with_time_axis=False)


def analyse_decay_batch(
    channels: Sequence[np.ndarray],
    sample_rate_hz: int,
    channel_names: Sequence[str],
    settings: DecayAnalysisSettings,
) -> List[ChannelDecayAnalysis]:
    """Batched form: every channel in one set of kernel launches."""
    for c in channels:
        if c.ndim != 1:
            raise ValueError("compute_schroeder_edc_db expects a 1D mono array.")
    decay_fit_specs(settings)          # validates the ranges before any device work
    eng = get_engine()
    batch = eng.upload(list(channels))
    dev = decay_device(eng, batch, sample_rate_hz, settings)
    return decay_records_to_results(dev, dev["fits"].cpu().numpy(), dev["cross"].cpu().numpy(),
                                    dev["edc"].cpu().numpy(), sample_rate_hz, channel_names)


def analyse_decay_for_channel(
    samples: np.ndarray,
    sample_rate_hz: int,
    channel_name: str,
    settings: DecayAnalysisSettings,
) -> ChannelDecayAnalysis:
    return analyse_decay_batch([samples], sample_rate_hz, [channel_name], settings)[0]


def analyse_decay_from_wav_file(
    input_wav_file_path: str | Path,
    settings: Optional[DecayAnalysisSettings] = None,
) -> List[ChannelDecayAnalysis]:
    if settings is None:
        settings = DecayAnalysisSettings()
    loaded, chans = wav_channels(input_wav_file_path, settings.use_mono_downmix_for_stereo)
    return analyse_decay_batch([c for _, c in chans], loaded.sample_rate_hz, [n for n, _ in chans], settings)


def plot_decay_from_wav_file(
    input_wav_file_path: str | Path,
    analysis_settings: Optional[DecayAnalysisSettings] = None,
    plot_settings: Optional[DecayPlotSettings] = None,
    output_basename: Optional[str | Path] = None,
    show_interactive: bool = True,
) -> List[ChannelDecayAnalysis]:
    """Analyse, then render <basename>_decay.png on the CPU (plotting is outside the accelerated path)."""
    analysis_settings = analysis_settings or DecayAnalysisSettings()
    plot_settings = plot_settings or DecayPlotSettings()
    results = analyse_decay_from_wav_file(input_wav_file_path, analysis_settings)
    from . import plotting
    plotting.render_decay(results, analysis_settings, plot_settings, f"Decay (EDC) — {input_wav_file_path}",
                          plotting.png_path(output_basename, "_decay"), show_interactive)
    return results


def summarise_decay_results_text(channel_analyses: List[ChannelDecayAnalysis]) -> str:
    """Plain-text, diff-stable summary in the reference's exact format (decay.py:502-542)."""
    out: List[str] = []
    for res in channel_analyses:
        out.append(f"[{res.channel_name}] analysis_start_sample_index={res.analysis_start_sample_index}")
        e = res.early_decay_10db_time_seconds
        out.append("  early_0_to_-10_time=NA" if e is None else f"  early_0_to_-10_time={e:.4f}s")
        if not res.fits:
            out += ["  fits=NA", ""]
            continue
        for key in ("EDT", "T20", "T30"):
            f = res.fits.get(key)
            if f is None:
                out.append(f"  {key}: NA")
            else:
                out.append(
                    f"  {f.name}: range=[{f.range_db[0]:.1f},{f.range_db[1]:.1f}]dB "
                    f"time=[{f.start_time_seconds:.4f},{f.end_time_seconds:.4f}]s "
                    f"slope={f.slope_db_per_second:.6f}dB/s r2={f.r_squared:.6f} rt60={f.rt60_seconds:.4f}s"
                )
        out.append("")
    return "\n".join(out).rstrip() + "\n"
