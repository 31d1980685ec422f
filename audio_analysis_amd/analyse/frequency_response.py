"""
Frequency response (whole-segment spectrum) on the GPU.

Host-side mirror of the reference's analyse/frequency_response.py (dataclasses :43-102,
analyse_frequency_response_for_channel :173-271, summary :424-432).  The windowed arbitrary-length rFFT is
ira_rfft_any (float64 Bluestein); dB conversion and peak/centroid statistics are ira_spectrum_* kernels.
The optional log-frequency smoothing (default off, :117-169) runs on the device too (ira_log_smooth_db, in place, before
the statistics kernel, which therefore sees the smoothed curve exactly as the reference recomputes them); a host
restatement remains for grids beyond the kernel's LDS budget.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from ..engine import get_engine
from ._common import segment_bounds, segment_bounds_batch, wav_channels


@dataclass(frozen=True)
class FrequencyResponseAnalysisSettings:
    use_mono_downmix_for_stereo: bool = False
    trim_to_peak: bool = True
    ignore_leading_seconds: float = 0.0
    analysis_duration_seconds: Optional[float] = None
    use_hann_window: bool = True
    magnitude_floor_db: float = -120.0
    f_min_hz: float = 20.0
    f_max_hz: float = 20000.0
    smoothing_log_bins: int = 0
    log_bins_per_octave: int = 96


@dataclass(frozen=True)
class FrequencyResponsePlotSettings:
    secondary_channel_alpha: float = 0.7
    ylim_db: Optional[Tuple[float, float]] = None


@dataclass(frozen=True)
class ChannelFrequencyResponse:
    channel_name: str
    sample_rate_hz: int
    analysis_start_sample_index: int
    analysis_length_samples: int
    frequency_hz: np.ndarray
    magnitude_db: np.ndarray
    peak_frequency_hz: float
    spectral_centroid_hz: float


def rfft_bin_step(length: int, sample_rate_hz: int) -> float:
    """The float64 step numpy.fft.rfftfreq(length, d=1/sr) multiplies the bin index by."""
    return 1.0 / (length * (1.0 / float(sample_rate_hz)))


def smooth_log_frequency(frequency_hz, magnitude_db, f_min_hz, f_max_hz, smoothing_log_bins, log_bins_per_octave,
                         through_float32: bool = False):
    """
    Moving average of a dB curve on a uniform log2(f) grid, interpolated back (host-side, optional).
    `through_float32` reproduces the waterfall module's variant, which round-trips the gridded curve
    through float32 around the convolution.
    """
    if smoothing_log_bins <= 1:
        return magnitude_db
    f = frequency_hz.astype(np.float64)
    lo = float(max(1.0, f_min_hz))
    hi = float(max(lo, f_max_hz))
    inside = (f >= lo) & (f <= hi)
    if not np.any(inside):
        return magnitude_db
    fs = f[inside]
    ms = magnitude_db.astype(np.float64)[inside]
    a, b = float(np.log2(fs[0])), float(np.log2(fs[-1]))
    count = int(max(8, np.ceil((b - a) * int(max(16, log_bins_per_octave))))) + 1
    grid = 2.0 ** np.linspace(a, b, count, dtype=np.float64)
    box = np.ones(int(smoothing_log_bins), dtype=np.float64) / float(smoothing_log_bins)
    on_grid = np.interp(grid, fs, ms)
    if through_float32:
        on_grid = np.convolve(on_grid.astype(np.float32).astype(np.float64), box, mode="same")
        on_grid = on_grid.astype(np.float32).astype(np.float64)
    else:
        on_grid = np.convolve(on_grid, box, mode="same")
    result = magnitude_db.astype(np.float32, copy=True)
    result[inside] = np.interp(fs, grid, on_grid).astype(np.float32)
    return result


def selected_bin_range(nbins: int, step: float, lo_hz: float, hi_hz: float):
    """(first bin, count) of the rFFT bins whose FLOAT32 frequency float32(k * step) lies in [lo_hz, hi_hz] -- the
    reference's boolean mask on rfftfreq(n).astype(float32) is one contiguous run; found by arithmetic plus a check of
    the neighbouring bins' float32 values instead of building the whole axis."""
    def f32(k):
        return float(np.float32(float(k) * step))
    k0 = int(np.clip(np.floor(lo_hz / step), 0, nbins))
    k0 = max(0, k0 - 2)
    while k0 < nbins and f32(k0) < lo_hz:
        k0 += 1
    k1 = int(np.clip(np.ceil(hi_hz / step), 0, nbins - 1))
    k1 = min(nbins - 1, k1 + 2)
    while k1 >= 0 and f32(k1) > hi_hz:
        k1 -= 1
    return k0, max(0, k1 - k0 + 1)


def spectrum_segments(eng, batch, sample_rate_hz: int, settings, what: str):
    """Time selection for fr / filter -> (starts, lens)."""
    peaks = eng.peaks(batch) if settings.trim_to_peak else np.zeros(batch.count, dtype=np.int64)
    starts, lens = segment_bounds_batch(batch.length, peaks, sample_rate_hz, settings.trim_to_peak,
                                        settings.ignore_leading_seconds, settings.analysis_duration_seconds)
    if np.any(lens < 32):
        raise ValueError(f"Not enough samples after trimming/selection to analyse {what}.")
    return starts, lens


def spectrum_device(eng, batch, sample_rate_hz: int, settings, what: str, want_phase: bool = False,
                    unwrap: bool = True, degrees: bool = True):
    """
    Device-resident whole-segment spectra: complex f64 half spectra, float32 dB magnitudes, optional
    (unwrapped) phase and the (n, 8) statistics records of ira_spectrum_stats, all left in HBM.
    """
    starts, lens = spectrum_segments(eng, batch, sample_rate_hz, settings, what)
    # nothing but the dB / phase kernel reads these spectra: even-length Bluestein elements stay packed (engine.rfft_any)
    spec, off, packed = eng.rfft_any_packed(batch.x, batch.off + starts, lens, bool(settings.use_hann_window))
    mag, ph = eng.spectrum_mag_phase(spec, off, lens, float(settings.magnitude_floor_db), want_phase=want_phase,
                                     packed=packed)
    phase = eng.phase_unwrap(ph, off, lens, unwrap, degrees) if want_phase else None
    nyq = 0.5 * float(sample_rate_hz)
    f_lo = float(np.clip(settings.f_min_hz, 0.0, nyq))
    f_hi = float(np.clip(settings.f_max_hz, f_lo, nyq))
    steps = np.array([rfft_bin_step(int(n), sample_rate_hz) for n in lens], dtype=np.float64)
    smoothed = False
    bins_w = int(getattr(settings, "smoothing_log_bins", 0) or 0)
    if bins_w > 1:
        # optional, default-off: log-frequency smoothing of the dB curve on the device, in place (ira_log_smooth_db); the
        # statistics below then see the smoothed curve, exactly as the reference recomputes them (frequency_response.py:236-260)
        s_lo = float(np.clip(settings.f_min_hz, 1.0, nyq))
        s_hi = float(np.clip(settings.f_max_hz, s_lo, nyq))
        rng = [selected_bin_range(int(n) // 2 + 1, float(st), max(1.0, s_lo), max(max(1.0, s_lo), s_hi))
               for n, st in zip(lens, steps)]
        k_lo = np.array([r[0] for r in rng], dtype=np.int32)
        nsel = np.array([r[1] for r in rng], dtype=np.int32)
        if np.all(nsel > 0):
            smoothed = eng.log_smooth(mag, off, np.ones(len(rng), np.int32), k_lo, nsel, steps, bins_w,
                                      int(getattr(settings, "log_bins_per_octave", 96)), through_float32=False)
    stats = eng.spectrum_stats(mag, off, lens, steps, f_lo, f_hi, 1000.0)
    # with packed elements `spec` holds half-length transforms Z in some slots, not half spectra: nothing may read it as
    # "the spectra"  -- it travels as spec_packed, and `spec` is only there when every slot is a spectrum
    return dict(spec=spec if packed is None else None, spec_packed=spec if packed is not None else None, off=off, packed=packed, starts=starts, lens=lens, mag=mag, phase=phase, stats=stats, f_lo=f_lo,
                f_hi=f_hi, smoothed=smoothed)


def analyse_frequency_response_batch(
    channels: Sequence[np.ndarray],
    sample_rate_hz: int,
    channel_names: Sequence[str],
    settings: FrequencyResponseAnalysisSettings,
) -> List[ChannelFrequencyResponse]:
    for c in channels:
        if c.ndim != 1:
            raise ValueError("analyse_frequency_response_for_channel expects a 1D mono array.")
    eng = get_engine()
    batch = eng.upload(list(channels))
    return frequency_response_results(spectrum_device(eng, batch, sample_rate_hz, settings, "spectrum"),
                                      sample_rate_hz, channel_names, settings)


def frequency_response_results(dev, sample_rate_hz: int, channel_names, settings) -> List[ChannelFrequencyResponse]:
    starts, lens, off, mag, f_lo, f_hi = dev["starts"], dev["lens"], dev["off"], dev["mag"], dev["f_lo"], dev["f_hi"]
    nyq = 0.5 * float(sample_rate_hz)
    # host-side smoothing only when the device did not do it (grids beyond the kernel's LDS budget)
    smoothing = bool(settings.smoothing_log_bins and int(settings.smoothing_log_bins) > 1) and not dev.get("smoothed")
    stats = None if smoothing else dev["stats"].cpu().numpy()
    mag_host = mag.cpu().numpy()
    out = []
    for i, name in enumerate(channel_names):
        n = int(lens[i])
        bins = n // 2 + 1
        freq = np.fft.rfftfreq(n, d=1.0 / float(sample_rate_hz)).astype(np.float32)
        db = mag_host[off[i] : off[i] + bins].copy()
        if smoothing:
            s_lo = float(np.clip(settings.f_min_hz, 1.0, nyq))
            s_hi = float(np.clip(settings.f_max_hz, s_lo, nyq))
            db = smooth_log_frequency(freq, db, s_lo, s_hi, int(settings.smoothing_log_bins),
                                      int(settings.log_bins_per_octave))
            sel = (freq >= f_lo) & (freq <= f_hi)
            if not np.any(sel):
                raise ValueError("Selected frequency range is empty (check f_min_hz/f_max_hz).")
            lin = 10.0 ** (db[sel].astype(np.float64) / 20.0)
            peak_hz = float(freq[sel][int(np.argmax(db[sel]))])
            wsum = float(np.sum(lin))
            centroid = float(np.sum(freq[sel].astype(np.float64) * lin) / wsum) if wsum > 0.0 else float(freq[sel][0])
        else:
            st = stats[i]
            if st[0] < 1.0:
                raise ValueError("Selected frequency range is empty (check f_min_hz/f_max_hz).")
            peak_hz = float(st[2])
            centroid = float(st[3] / st[4]) if st[4] > 0.0 else float(st[5])
        out.append(ChannelFrequencyResponse(
            channel_name=name, sample_rate_hz=int(sample_rate_hz), analysis_start_sample_index=int(starts[i]),
            analysis_length_samples=n, frequency_hz=freq, magnitude_db=db.astype(np.float32),
            peak_frequency_hz=peak_hz, spectral_centroid_hz=centroid,
        ))
    return out


def frequency_response_summary_lines(dev, sample_rate_hz: int, channel_names, settings) -> List[str]:
    """One summarise_frequency_response_results_text line per channel from the (n, 8) statistics records alone; the
    spectra stay in HBM.  (With the optional log-frequency smoothing the statistics depend on the smoothed curve: the
    full path is used.)"""
    if settings.smoothing_log_bins and int(settings.smoothing_log_bins) > 1 and not dev.get("smoothed"):
        return [summarise_frequency_response_results_text([r])
                for r in frequency_response_results(dev, sample_rate_hz, channel_names, settings)]
    stats = dev["stats"].cpu().numpy()
    rows = []
    for i, name in enumerate(channel_names):
        st = stats[i]
        if st[0] < 1.0:
            raise ValueError("Selected frequency range is empty (check f_min_hz/f_max_hz).")
        peak_hz = float(st[2])
        centroid = float(st[3] / st[4]) if st[4] > 0.0 else float(st[5])
        rows.append(f"[{name}] start_sample={int(dev['starts'][i])}  len_samples={int(dev['lens'][i])}  "
                    f"peak={peak_hz:.1f}Hz  centroid={centroid:.1f}Hz")
    return rows


def analyse_frequency_response_for_channel(
    samples: np.ndarray,
    sample_rate_hz: int,
    channel_name: str,
    settings: FrequencyResponseAnalysisSettings,
) -> ChannelFrequencyResponse:
    return analyse_frequency_response_batch([samples], sample_rate_hz, [channel_name], settings)[0]


def analyse_frequency_response_from_wav_file(
    input_wav_file_path: str | Path,
    settings: Optional[FrequencyResponseAnalysisSettings] = None,
) -> List[ChannelFrequencyResponse]:
    settings = settings or FrequencyResponseAnalysisSettings()
    loaded, chans = wav_channels(input_wav_file_path, settings.use_mono_downmix_for_stereo)
    return analyse_frequency_response_batch([c for _, c in chans], loaded.sample_rate_hz, [n for n, _ in chans],
                                            settings)


def plot_frequency_response_from_wav_file(
    input_wav_file_path: str | Path,
    analysis_settings: Optional[FrequencyResponseAnalysisSettings] = None,
    plot_settings: Optional[FrequencyResponsePlotSettings] = None,
    output_basename: Optional[str | Path] = None,
    show_interactive: bool = True,
) -> List[ChannelFrequencyResponse]:
    analysis_settings = analysis_settings or FrequencyResponseAnalysisSettings()
    plot_settings = plot_settings or FrequencyResponsePlotSettings()
    results = analyse_frequency_response_from_wav_file(input_wav_file_path, analysis_settings)
    from . import plotting
    plotting.render_frequency_response(results, analysis_settings, plot_settings,
                                       f"Frequency response (spectrum) — {input_wav_file_path}",
                                       plotting.png_path(output_basename, "_fr"), show_interactive)
    return results


def summarise_frequency_response_results_text(channel_results: List[ChannelFrequencyResponse]) -> str:
    return "\n".join(
        f"[{r.channel_name}] start_sample={r.analysis_start_sample_index}  "
        f"len_samples={r.analysis_length_samples}  "
        f"peak={r.peak_frequency_hz:.1f}Hz  centroid={r.spectral_centroid_hz:.1f}Hz"
        for r in channel_results
    )
