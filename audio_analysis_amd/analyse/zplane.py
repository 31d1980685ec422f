"""
Z-plane pole (and optional zero) estimation from an impulse response, on the GPU.

Host-side mirror of the reference's analyse/zplane.py: settings/result dataclasses (:45-80), the helper
functions _fit_ar_least_squares (:83-120), _derive_fir_numerator_from_ar (:123-142),
_roots_from_poly_descending (:145-158), _rt60_from_pole_radius (:161-173), the plot/analysis entry
plot_zplane_from_wav_file (:176-285) and summarise_zplane_results_text (:288-302).
The reference has no analysis-only function (numerics live inside the plot function); analyse_zplane_batch
is the batched numeric body here and the plot function calls it.

Device work: ira_ar_fit (float64 MFMA Gram of the implicit Hankel matrix + Cholesky solve), ira_poly_roots
(Aberth-Ehrlich), ira_fir_numerator.  The reference's lstsq is SVD based; the normal-equation solution agrees
with it to about cond(A)^2 * 1e-16 (see DESIGN.md for the stated tolerance).  Root ORDER differs from
numpy.roots (which leaves it unspecified): compare sorted sets.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np

from ..engine import get_engine
from .io import get_analysis_channels, load_wav_file


@dataclass(frozen=True)
class ZPlaneAnalysisSettings:
    use_mono_downmix_for_stereo: bool = False
    trim_to_peak: bool = True
    ignore_leading_seconds: float = 0.0
    analysis_duration_seconds: Optional[float] = None
    model: str = "ar"
    ar_order: int = 256
    derive_zeros: bool = False
    zero_order: int = 64
    normalise_segment: bool = True
    ridge_lambda: float = 0.0


@dataclass(frozen=True)
class ZPlanePlotSettings:
    secondary_channel_alpha: float = 0.7
    show_unit_circle: bool = True
    show_axes: bool = True
    limit_radius: float = 1.2
    annotate_stats: bool = True


@dataclass(frozen=True)
class ChannelZPlaneResult:
    channel_name: str
    sample_rate_hz: int
    poles: np.ndarray
    zeros: Optional[np.ndarray]


def _to_complex(roots_dev, counts_dev) -> List[np.ndarray]:
    """roots/counts: device tensors, or HostFutures from Engine.fetch (deferred path)."""
    r = roots_dev.get() if hasattr(roots_dev, "get") else roots_dev.cpu().numpy()
    c = counts_dev.get() if hasattr(counts_dev, "get") else counts_dev.cpu().numpy()
    z = np.empty(r.shape[:2], dtype=np.complex128)                     # one pass over the record, then views per element
    z.real = r[:, :, 0]
    z.imag = r[:, :, 1]
    return [z[i, : c[i]] for i in range(r.shape[0])]


def _fit_ar_least_squares(x: np.ndarray, order: int, ridge_lambda: float = 0.0) -> np.ndarray:
    """AR coefficients [1, a1..ap] of x[n] + sum a_k x[n-k] = e[n] (covariance method)."""
    x = np.asarray(x, dtype=np.float64)
    p = int(order)
    if p < 1:
        return np.array([1.0], dtype=np.float64)
    if x.size <= p:
        p = max(1, x.size - 1)
    if x.size <= p:
        raise ValueError("AR fit needs at least two samples.")
    eng = get_engine()
    xd = eng.to_dev(x)
    co, info = eng.ar_fit(xd, np.zeros(1, np.int64), np.array([x.size], np.int32), None, p,
                          float(ridge_lambda) if ridge_lambda and ridge_lambda > 0.0 else 0.0, x_is_f64=True)
    _raise_if_not_finite(info.cpu().numpy()[:, 0])
    return co.cpu().numpy()[0].copy()


AR_STATUS_NOT_FINITE = 3.0      # ira_ar_minnorm: the Gram matrix holds NaN / infinity


def _raise_if_not_finite(status: np.ndarray) -> None:
    """numpy.linalg.lstsq raises LinAlgError on NaN / infinite input (reference zplane.py:117); so does the drop-in API."""
    if np.any(np.asarray(status) == AR_STATUS_NOT_FINITE):
        raise np.linalg.LinAlgError("SVD did not converge in Linear Least Squares")


def _derive_fir_numerator_from_ar(a: np.ndarray, h: np.ndarray, zero_order: int) -> np.ndarray:
    """b[n] = sum_k a[k] h[n-k] for n = 0..Q: a short numerator matching the first samples of h."""
    q = int(max(0, zero_order))
    a = np.asarray(a, dtype=np.float64)
    h = np.asarray(h, dtype=np.float64)[: q + 1]
    full = np.convolve(a, h)      # tiny (<= Q+1 by p+1) host product of results already on the host
    out = np.zeros(q + 1, dtype=np.float64)
    out[: min(q + 1, full.size)] = full[: q + 1]
    return out


def _roots_from_poly_descending(poly: np.ndarray) -> np.ndarray:
    """Roots of a real polynomial given in descending powers (trailing |c| < 1e-14 dropped first)."""
    poly = np.asarray(poly, dtype=np.float64)
    if poly.size <= 1:
        return np.array([], dtype=np.complex128)
    eng = get_engine()
    roots, cnt = eng.poly_roots(eng.to_dev(poly.reshape(1, -1)), 1, int(poly.size), 1e-14)
    return _to_complex(roots, cnt)[0]


def _rt60_from_pole_radius(r: float, sample_rate_hz: int) -> float:
    r = float(r)
    if r <= 0.0 or r >= 1.0:
        return float("inf")
    return np.log(1000.0) * ((-1.0 / np.log(r)) / float(sample_rate_hz))


def analyse_zplane_batch(
    channels: Sequence[np.ndarray],
    sample_rate_hz: int,
    channel_names: Sequence[str],
    settings: ZPlaneAnalysisSettings,
) -> List[ChannelZPlaneResult]:
    """Numeric body of the reference's plot function for a whole batch of channels."""
    eng = get_engine()
    batch = eng.upload(list(channels))
    poles, zeros, status, _ = zplane_device(eng, batch, sample_rate_hz, settings, with_status=True)
    _raise_if_not_finite(status)
    return [ChannelZPlaneResult(channel_name=name, sample_rate_hz=sample_rate_hz, poles=poles[i],
                                zeros=zeros[i] if settings.derive_zeros else None)
            for i, name in enumerate(channel_names)]


def zplane_device(eng, batch, sample_rate_hz: int, settings: ZPlaneAnalysisSettings, defer: bool = False,
                  with_status: bool = False):
    """AR fit + roots for a device-resident batch; returns host lists (poles, zeros) of complex128 arrays
    (or, with defer=True, a zero-argument callable producing them after all launches have been enqueued).
    with_status adds the per-channel solver status (info[0] of ira_ar_solve and its followers: 0 solved, 2 refined, 5 solved
    by the double-double normal equations, 4 minimum norm over a rank-deficient Gram matrix, 3 not finite -- the reference's
    lstsq raises LinAlgError there) and the condition estimate of the float64 Gram matrix (info[3]; for status 4 the rank)."""
    nch = batch.count
    peaks = eng.peaks(batch) if settings.trim_to_peak else np.zeros(nch, dtype=np.int64)
    skip = int(round(float(settings.ignore_leading_seconds) * sample_rate_hz))
    start = np.clip(peaks + skip, 0, batch.length)
    if settings.analysis_duration_seconds is None:
        seg_len = batch.length - start
    else:
        want = max(1, int(round(float(settings.analysis_duration_seconds) * sample_rate_hz)))
        seg_len = np.minimum(want, batch.length - start)
    if np.any(seg_len < 2):
        raise ValueError("Not enough samples after trimming/selection for the AR fit.")
    seg_off = batch.off + start
    divisor = None
    if settings.normalise_segment:
        if settings.trim_to_peak and skip == 0 and batch.peak_abs is not None:
            pk = batch.peak_abs.astype(np.float64)      # the segment starts AT the global peak: its max is that sample
        else:
            pk = eng.segment_peaks(batch.x, seg_off, seg_len)
        divisor = np.where(pk > 0.0, pk, 1.0)

    order = int(settings.ar_order)
    poles: List[Optional[np.ndarray]] = [None] * nch
    zeros: List[Optional[np.ndarray]] = [None] * nch
    if order < 1:
        for i in range(nch):
            poles[i] = np.array([], dtype=np.complex128)
        eff = np.zeros(nch, dtype=np.int64)
    else:
        eff = np.where(seg_len <= order, np.maximum(1, seg_len - 1), order).astype(np.int64)
    ridge = float(settings.ridge_lambda) if settings.ridge_lambda and settings.ridge_lambda > 0.0 else 0.0
    pending = []
    for p in sorted(set(eff.tolist())):
        if p < 1:
            continue
        idx = np.nonzero(eff == p)[0]
        div = None if divisor is None else divisor[idx]
        co, info = eng.ar_fit(batch.x, seg_off[idx], seg_len[idx], div, int(p), ridge)
        roots, cnt = eng.poly_roots(co, int(idx.size), int(p) + 1, 1e-14)
        zr = zc = None
        if settings.derive_zeros:
            q = int(max(0, settings.zero_order))
            b = eng.fir_numerator(co, int(p), batch.x, seg_off[idx], seg_len[idx], div, q)
            zr, zc = eng.poly_roots(b, int(idx.size), q + 1, 1e-14)
        if defer:
            roots, cnt, info = eng.fetch(roots), eng.fetch(cnt), eng.fetch(info)
            if zr is not None:
                zr, zc = eng.fetch(zr), eng.fetch(zc)
        pending.append((idx, roots, cnt, zr, zc, info))

    status = np.zeros(nch, dtype=np.float64)
    cond = np.full(nch, np.nan, dtype=np.float64)

    def finish():
        for idx, roots, cnt, zr, zc, info in pending:
            for k, r in zip(idx, _to_complex(roots, cnt)):
                poles[k] = r
            if zr is not None:
                for k, r in zip(idx, _to_complex(zr, zc)):
                    zeros[k] = r
            st = info.get() if hasattr(info, "get") else info.cpu().numpy()
            status[idx] = st[:, 0]
            cond[idx] = st[:, 3]
            for k in idx[st[:, 0] == AR_STATUS_NOT_FINITE]:
                poles[k] = np.array([], dtype=np.complex128)       # no fit: the reference raises for this channel
                if zeros[k] is not None:
                    zeros[k] = np.array([], dtype=np.complex128)
        return (poles, zeros, status, cond) if with_status else (poles, zeros)

    return finish if defer else finish()


def plot_zplane_from_wav_file(
    input_wav_file_path: str,
    settings: ZPlaneAnalysisSettings,
    plot_settings: ZPlanePlotSettings,
    output_basename: Optional[str | Path] = None,
    show_interactive: bool = True,
) -> List[ChannelZPlaneResult]:
    loaded = load_wav_file(input_wav_file_path, expected_channel_mode="mono_or_stereo",
                           allow_mono_and_upmix_to_stereo=False)
    chans = get_analysis_channels(loaded, use_mono_downmix_for_stereo=settings.use_mono_downmix_for_stereo)
    results = analyse_zplane_batch([c for _, c in chans], loaded.sample_rate_hz, [n for n, _ in chans], settings)
    from . import plotting
    for r in results:
        path = None
        if output_basename is not None:
            path = Path(str(Path(output_basename).with_suffix("")) + f"_zplane_{r.channel_name}.png")
        plotting.render_zplane(r, settings, plot_settings, f"Z-plane pole cloud ({r.channel_name})", path,
                               show_interactive, _rt60_from_pole_radius)
    return results


def summarise_zplane_results_text(results: List[ChannelZPlaneResult]) -> str:
    rows: List[str] = []
    for r in results:
        if r.poles.size == 0:
            rows.append(f"- {r.channel_name}: no poles (fit failed or order=0)")
            continue
        radius = np.abs(r.poles)
        rows.append(
            f"- {r.channel_name}: poles={r.poles.size}, max|p|={float(np.max(radius)):.6f}, "
            f"median|p|={float(np.median(radius)):.6f}, unstable(|p|>=1)={int(np.sum(radius >= 1.0))}"
        )
    return "Z-plane summary:\n" + "\n".join(rows) if rows else "No z-plane results."
