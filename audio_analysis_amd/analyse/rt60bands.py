"""
RT60 by frequency band (three-band / octave / third-octave FFT filter bank) on the GPU.

Host-side mirror of the reference's analyse/rt60bands.py (dataclasses :44-104, band tables :183-264,
analyse_rt60_bands_for_channel :324-413, summary :627-666).  Device work per channel:
  one forward float64 rFFT of the full file (ira_rfft_any; the reference recomputes it for every band),
  masked inverse transforms two bands at a time (ira_band_irfft, masks evaluated on the device in float32
  like the reference), Schroeder EDC per band from the common start index (ira_edc_db) and the T30
  (+T20/EDT) line fits (ira_curve_fits).
The filtering is circular over the FULL file length, exactly like the reference (pre-ringing wraps to the
tail and shapes the low-band values; this is reproduced, not "fixed").
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from ..engine import get_engine
from ._common import wav_channels
from .decay import DecayAnalysisSettings, compute_schroeder_edc_db, fit_decay_slope_over_db_range  # noqa: F401
from .frequency_response import rfft_bin_step


@dataclass(frozen=True)
class Rt60BandsAnalysisSettings:
    band_mode: str = "three"
    low_upper_hz: float = 250.0
    mid_center_hz: float = 1000.0
    mid_width_octaves: float = 2.0
    high_lower_hz: float = 4000.0
    f_min_hz: float = 31.5
    f_max_hz: float = 16000.0
    transition_width_octaves: float = (1.0 / 6.0)
    include_t20: bool = False
    include_edt: bool = False
    decay_settings: DecayAnalysisSettings = DecayAnalysisSettings()


@dataclass(frozen=True)
class Rt60BandsPlotSettings:
    ylim_seconds: Optional[Tuple[float, float]] = None
    secondary_channel_alpha: float = 0.7
    legend_values: bool = True


@dataclass(frozen=True)
class BandDefinition:
    name: str
    centre_hz: float
    kind: str
    low_edge_hz: Optional[float] = None
    high_edge_hz: Optional[float] = None


@dataclass(frozen=True)
class Rt60BandMetrics:
    rt60_t30_seconds: Optional[float]
    rt60_t20_seconds: Optional[float]
    edt_seconds: Optional[float]


@dataclass(frozen=True)
class Rt60BandsChannelResult:
    channel_name: str
    sample_rate_hz: int
    band_definitions: List[BandDefinition]
    band_metrics_by_name: Dict[str, Rt60BandMetrics]


# ---------------------------------------------------------------------------------------------------
# band tables (host, float64 -- must reproduce the reference's Python arithmetic digit for digit)
# ---------------------------------------------------------------------------------------------------


def _three_bands(s: Rt60BandsAnalysisSettings, nyq: float) -> List[BandDefinition]:
    low_upper = float(np.clip(s.low_upper_hz, 20.0, nyq))
    centre = float(np.clip(s.mid_center_hz, 20.0, nyq))
    half_width = 0.5 * float(max(0.1, s.mid_width_octaves))
    ratio = float(2.0 ** float(half_width))
    mid_lo = float(np.clip(centre / ratio, 20.0, nyq))
    mid_hi = float(np.clip(centre * ratio, 20.0, nyq))
    high_lower = float(np.clip(s.high_lower_hz, 20.0, nyq))
    return [
        BandDefinition("Low", float(np.sqrt(20.0 * low_upper)), "lowpass", high_edge_hz=low_upper),
        BandDefinition("Mid", centre, "bandpass", low_edge_hz=mid_lo, high_edge_hz=mid_hi),
        BandDefinition("High", float(np.sqrt(max(20.0, high_lower) * nyq)), "highpass", low_edge_hz=high_lower),
    ]


def _fractional_octave_bands(s: Rt60BandsAnalysisSettings, nyq: float, per_octave: int) -> List[BandDefinition]:
    f_lo = float(max(20.0, min(s.f_min_hz, nyq)))
    f_hi = float(max(f_lo, min(s.f_max_hz, nyq)))
    n = float(per_octave)
    step = 2.0 ** (1.0 / n)
    edge = 2.0 ** (1.0 / (2.0 * n))
    k_first = int(np.floor(np.log(f_lo / 1000.0) / np.log(step)))
    k_last = int(np.ceil(np.log(f_hi / 1000.0) / np.log(step)))
    table: List[BandDefinition] = []
    for k in range(k_first, k_last + 1):
        fc = 1000.0 * (step ** float(k))
        if not (f_lo <= fc <= f_hi):
            continue
        lo = float(np.clip(fc / edge, 20.0, nyq))
        hi = float(np.clip(fc * edge, 20.0, nyq))
        if hi > lo:
            table.append(BandDefinition(f"{int(round(fc))}Hz", float(fc), "bandpass", lo, hi))
    return sorted(table, key=lambda b: b.centre_hz)


def _build_band_definitions(settings: Rt60BandsAnalysisSettings, sample_rate_hz: int) -> List[BandDefinition]:
    nyq = 0.5 * float(sample_rate_hz)
    mode = str(settings.band_mode).lower()
    if mode == "three":
        return _three_bands(settings, nyq)
    if mode == "octave":
        return _fractional_octave_bands(settings, nyq, 1)
    if mode == "third":
        return _fractional_octave_bands(settings, nyq, 3)
    raise ValueError(f"Unknown band_mode: {settings.band_mode}")


def _lowpass_ramp(pass_hz: float, transition_oct: float, nyq: float) -> Tuple[float, float]:
    """(ramp start = pass edge, ramp end = stop edge) of the reference's low-pass mask."""
    p = float(np.clip(pass_hz, 1.0, nyq))
    stop = float(min(nyq, p * float(2.0 ** float(transition_oct))))
    if stop <= p:
        stop = min(nyq, p + 1.0)
    return p, float(stop)


def _highpass_ramp(pass_hz: float, transition_oct: float, nyq: float) -> Tuple[float, float]:
    """(ramp start = stop edge, ramp end = pass edge) of the reference's high-pass mask."""
    p = float(np.clip(pass_hz, 1.0, nyq))
    stop = float(max(1.0, p / float(2.0 ** float(transition_oct))))
    if p <= stop:
        stop = max(1.0, p - 1.0)
    return float(stop), p


def band_mask_record(band: BandDefinition, transition_oct: float, nyq: float) -> np.ndarray:
    """8-double descriptor consumed by ira_band_irfft: [kind, hp_x0, hp_x1, lp_x0, lp_x1, 0, 0, 0]."""
    rec = np.zeros(8, dtype=np.float64)
    if band.kind == "lowpass":
        rec[0] = 1.0
        rec[3], rec[4] = _lowpass_ramp(band.high_edge_hz, transition_oct, nyq)
    elif band.kind == "highpass":
        rec[0] = 2.0
        rec[1], rec[2] = _highpass_ramp(band.low_edge_hz, transition_oct, nyq)
    elif band.kind == "bandpass":
        lo = float(np.clip(band.low_edge_hz, 1.0, nyq))
        hi = float(np.clip(band.high_edge_hz, 1.0, nyq))
        if hi > lo:
            rec[0] = 3.0
            rec[1], rec[2] = _highpass_ramp(lo, transition_oct, nyq)
            rec[3], rec[4] = _lowpass_ramp(hi, transition_oct, nyq)
    else:
        raise ValueError(f"Unknown band kind: {band.kind}")
    return rec


# ---------------------------------------------------------------------------------------------------
# analysis
# ---------------------------------------------------------------------------------------------------


def analyse_rt60_bands_batch(
    channels: Sequence[np.ndarray],
    sample_rate_hz: int,
    channel_names: Sequence[str],
    settings: Rt60BandsAnalysisSettings,
) -> List[Rt60BandsChannelResult]:
    eng = get_engine()
    chans = [c.astype(np.float32, copy=False) for c in channels]
    batch = eng.upload(chans)
    bands, values, have = rt60_bands_device(eng, batch, sample_rate_hz, settings)
    return rt60_bands_results(bands, values, have, sample_rate_hz, channel_names)


def rt60_bands_results(bands, values, have, sample_rate_hz: int, channel_names) -> List[Rt60BandsChannelResult]:
    def opt(v):
        return None if np.isnan(v) else float(v)

    out = []
    for c, name in enumerate(channel_names):
        metrics: Dict[str, Rt60BandMetrics] = {}
        for b, band in enumerate(bands):
            if not have[c, b]:
                metrics[band.name] = Rt60BandMetrics(None, None, None)
            else:
                metrics[band.name] = Rt60BandMetrics(opt(values[c, b, 0]), opt(values[c, b, 1]), opt(values[c, b, 2]))
        out.append(Rt60BandsChannelResult(channel_name=name, sample_rate_hz=sample_rate_hz,
                                          band_definitions=list(bands), band_metrics_by_name=metrics))
    return out


def rt60_bands_device(eng, batch, sample_rate_hz: int, settings: Rt60BandsAnalysisSettings, defer: bool = False):
    """
    Filter bank + per-band decay fits for a device-resident batch.
    Returns (bands, values (nch, nbands, 3) float64 [t30, t20, edt; NaN = no fit], have (nch, nbands) bool).
    With defer=True the device->host copy of the fit records is postponed: the second element is then a
    zero-argument callable producing `values` (lets a pipeline enqueue more work before synchronising).
    """
    t = eng.torch
    dec = settings.decay_settings
    smooth = int(dec.edc_smoothing_window_samples or 0)
    nch = batch.count
    n_all = batch.length
    if np.any(n_all < 8):
        raise ValueError("Not enough samples for rt60bands analysis.")
    peaks = eng.peaks(batch) if dec.trim_to_peak else np.zeros(nch, dtype=np.int64)
    skip = np.zeros(nch, dtype=np.int64)
    if dec.ignore_leading_seconds > 0.0:
        raw = int(round(dec.ignore_leading_seconds * float(sample_rate_hz)))
        skip = np.clip(raw, 0, n_all)
    start = np.minimum(n_all, peaks + skip)

    bands = _build_band_definitions(settings, sample_rate_hz)
    nyq = 0.5 * float(sample_rate_hz)
    records = np.stack([band_mask_record(b, settings.transition_width_octaves, nyq) for b in bands]) \
        if bands else np.zeros((0, 8))
    nb = len(bands)

    fits_spec = [("t30", dec.t30_range_db)]
    if settings.include_t20:
        fits_spec.append(("t20", dec.t20_range_db))
    if settings.include_edt:
        fits_spec.append(("edt", dec.edt_range_db))
    ranges = []
    for _, rng in fits_spec:
        hi, lo = float(rng[0]), float(rng[1])
        if lo > hi:
            raise ValueError("range_db should be (higher_db, lower_db), e.g. (-5, -25).")
        ranges.append((hi, max(lo, float(dec.fit_lower_limit_db))))

    # forward spectra of the full files
    spec, spec_off = eng.rfft_any(batch.x, batch.off, n_all, use_hann=False)

    # band signals: y[c][b] has n_c float32 samples, channel after channel, band after band
    # (array arithmetic instead of nested Python loops over 256 channels x bands: the loops were ~2 ms of the host's ~6 ms
    # per report step, which is what bounds the bundle configuration -- its kernels take less time than its host code)
    n64 = n_all.astype(np.int64)
    per_entry = np.repeat(n64, nb)
    y_off = (np.cumsum(per_entry) - per_entry).reshape(nch, nb) if nb else np.zeros((nch, 1), dtype=np.int64)
    pos = int(per_entry.sum())
    y = eng.empty(pos, t.float32)
    if nb:
        # one entry per (channel, band); the engine pairs entries of equal length two per inverse transform
        fv_of = {int(v): rfft_bin_step(int(v), sample_rate_hz) for v in np.unique(n64)}
        fv = np.array([fv_of[int(v)] for v in n64], dtype=np.float64)
        # (the inverses also leave the energies of every band signal's EDC tiles: the fused fits below then read a band
        # signal once less -- not for the smoothed-curve path, which materialises the curve through ira_edc_db)
        band_tiles = eng.band_irfft(spec, np.repeat(np.asarray(spec_off, dtype=np.int64), nb),
                                    np.repeat(n64, nb).astype(np.int32), np.tile(records, (nch, 1)), np.repeat(fv, nb), y,
                                    y_off.reshape(-1), want_tiles=smooth <= 1)
    else:
        band_tiles = None

    # Schroeder EDC + fits on every (channel, band) tail with at least 8 samples
    tail_c = n64 - start.astype(np.int64)
    keep = np.nonzero(tail_c >= 8)[0]
    seg_c = np.repeat(keep, nb)
    seg_b = np.tile(np.arange(nb, dtype=np.int64), keep.size)
    seg_off = (y_off[seg_c, seg_b] + start.astype(np.int64)[seg_c]) if seg_c.size else np.zeros(0, np.int64)
    seg_len = tail_c[seg_c]
    values = np.full((nch, max(nb, 1), 3), np.nan)
    have = np.zeros((nch, max(nb, 1)), dtype=bool)
    if seg_c.size:
        seg_off_a, seg_len_a = np.asarray(seg_off, np.int64), np.asarray(seg_len, np.int64)
        if np.any(seg_len_a < 4):
            raise ValueError("Not enough samples after trimming/ignoring to compute EDC.")
        if smooth > 1:
            # default-off dB smoothing (decay.py:161-164 through rt60bands.py:356-360): the smoothed curve is what the fits
            # see, so it is materialised: EDC (unfloored f64) -> box smoothing + floor -> crossings / fits on the curve
            if np.any(seg_len_a < smooth):
                raise ValueError("edc_smoothing_window_samples exceeds the analysed length")
            _, e_off, raw64 = eng.edc_db(y, seg_off_a, seg_len_a, dec.edc_epsilon, dec.edc_floor_db, want_f64=True)
            edc = eng.edc_box_smooth(raw64, e_off, seg_len_a, smooth, dec.edc_floor_db)
            fit_dev, _ = eng.curve_fits(edc, e_off, seg_len_a, 1.0, float(sample_rate_hz), ranges, 8)
        else:
            # fused EDC -> crossings -> fits (ira_edc_fits): a band's EDC curve only feeds its fits (reference
            # rt60bands.py:272-321), so it is never written
            seg_tiles = None
            if band_tiles is not None:
                entry = seg_c * nb + seg_b                     # the (channel, band) entry of every segment; a segment ends
                seg_tiles = (band_tiles[0], band_tiles[1][entry], band_tiles[2][entry], band_tiles[3][entry])   # where its signal ends
            fit_dev, _, _, _ = eng.edc_fits(y, seg_off_a, seg_len_a, dec.edc_epsilon, dec.edc_floor_db, 1.0,
                                            float(sample_rate_hz), ranges, 8, tiles=seg_tiles)
        ci, bi = seg_c, seg_b
        have[ci, bi] = True

        fut = eng.fetch(fit_dev) if defer else None

        def finish():
            fit = fut.get() if fut is not None else fit_dev.cpu().numpy()
            for j, (key, _) in enumerate(fits_spec):
                col = {"t30": 0, "t20": 1, "edt": 2}[key]
                ok = fit[:, j, 0] == 1.0
                values[ci[ok], bi[ok], col] = fit[ok, j, 6]
            return values

        if defer:
            return bands, finish, have
        finish()

    return bands, values, have


def analyse_rt60_bands_for_channel(
    samples: np.ndarray,
    sample_rate_hz: int,
    channel_name: str,
    settings: Rt60BandsAnalysisSettings,
) -> Rt60BandsChannelResult:
    return analyse_rt60_bands_batch([samples], sample_rate_hz, [channel_name], settings)[0]


def analyse_rt60_bands_from_wav_file(
    input_wav_file_path: str | Path,
    settings: Optional[Rt60BandsAnalysisSettings] = None,
) -> List[Rt60BandsChannelResult]:
    settings = settings or Rt60BandsAnalysisSettings()
    loaded, chans = wav_channels(input_wav_file_path, settings.decay_settings.use_mono_downmix_for_stereo)
    return analyse_rt60_bands_batch([c for _, c in chans], loaded.sample_rate_hz, [n for n, _ in chans], settings)


def plot_rt60_bands_from_wav_file(
    input_wav_file_path: str | Path,
    settings: Optional[Rt60BandsAnalysisSettings] = None,
    plot_settings: Optional[Rt60BandsPlotSettings] = None,
    output_basename: Optional[str | Path] = None,
    show_interactive: bool = True,
) -> List[Rt60BandsChannelResult]:
    settings = settings or Rt60BandsAnalysisSettings()
    plot_settings = plot_settings or Rt60BandsPlotSettings()
    if plot_settings.legend_values and str(settings.band_mode).lower() in ("octave", "third"):
        plot_settings = replace(plot_settings, legend_values=False)
    results = analyse_rt60_bands_from_wav_file(input_wav_file_path, settings)
    from . import plotting
    plotting.render_rt60_bands(results, settings, plot_settings, f"RT60 bands — {input_wav_file_path}",
                               plotting.png_path(output_basename, "_rt60bands"), show_interactive)
    return results


def summarise_rt60_bands_results_text(
    channel_results: List[Rt60BandsChannelResult],
    include_t20: bool,
    include_edt: bool,
) -> str:
    cols = ["T30"] + (["T20"] if include_t20 else []) + (["EDT"] if include_edt else [])
    pick = {"T30": "rt60_t30_seconds", "T20": "rt60_t20_seconds", "EDT": "edt_seconds"}
    lines: List[str] = []
    for ch in channel_results:
        lines.append(f"[{ch.channel_name}]")
        lines.append("  ".join(["Band"] + [f"{m}_RT60(s)" for m in cols]))
        for band in ch.band_definitions:
            bm = ch.band_metrics_by_name.get(band.name)
            cells = [band.name]
            for m in cols:
                v = None if bm is None else getattr(bm, pick[m])
                cells.append("NA" if v is None else f"{float(v):.3f}")
            lines.append("  ".join(cells))
        lines.append("")
    return "\n".join(lines)
