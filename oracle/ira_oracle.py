"""
oracle/ira_oracle.py -- CPU restatement of the reference's analyse/ hot path (NumPy).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.
The product path (audio_analysis_amd/) never imports anything from oracle/ and fails
loudly when the HIP library is missing.

What it is: a from-scratch NumPy re-derivation of every function SURVEY.md section 8(a)
lists, written from the behavioural description of the reference
(kianmcevoy/audio_analysis, analyse/*.py).  Each function cites the reference file:line it
follows.  Arithmetic that the reference delegates to third-party code (numpy.fft = pocketfft,
numpy.linalg.lstsq = LAPACK gelsd, numpy.roots = LAPACK geev) is delegated to the same NumPy
entry points here, so that the oracle reproduces the reference bit-for-bit under the same
NumPy build.

Parity pinning: tests/golden/*.npz were generated in the build container by importing the
reference itself (tests/golden/make_goldens.py, numpy 2.2.6 / scipy 1.15.3) on seeded inputs;
tests/test_oracle_vs_golden.py checks this file against every one of them (bit-exact for
indices and for the float arrays, since both sides run the same NumPy).  The reference ships no
tests or golden vectors of its own.

Results are returned as plain dicts (no dataclasses) so nothing here mirrors the reference's
type layout; the product's host layer (audio_analysis_amd/analyse/) owns the drop-in dataclasses.
"""

from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

SR = 48_000

# ----------------------------------------------------------------------------------------
# a1  WAV sample conversion + channel policy            (reference analyse/io.py:46-113, 66-95)
# ----------------------------------------------------------------------------------------


def pcm_to_float32(raw: np.ndarray) -> np.ndarray:
    """io.py:98-113 (+ :46-64): int16/32768, int32/2^31, float passthrough; then clip to [-1,1]."""
    if np.issubdtype(raw.dtype, np.floating):
        f = raw.astype(np.float32, copy=False)
    elif raw.dtype == np.int16:
        f = raw.astype(np.float32) / 32768.0
    elif raw.dtype == np.int32:
        f = raw.astype(np.float32) / 2147483648.0
    elif np.issubdtype(raw.dtype, np.integer):
        raise ValueError(f"Unsupported integer PCM dtype: {raw.dtype}")
    else:
        raise ValueError(f"Unsupported WAV dtype: {raw.dtype}")
    return np.clip(f, -1.0, 1.0).astype(np.float32)


def analysis_channels(samples_nc: np.ndarray, mono_downmix: bool = False) -> List[Tuple[str, np.ndarray]]:
    """io.py:66-95: (N,C) float32 -> [("mono",x)] | [("left",L),("right",R)] | [("mono",0.5*(L+R))]."""
    c = samples_nc.shape[1]
    if c == 1:
        return [("mono", samples_nc[:, 0].astype(np.float32, copy=False))]
    if c == 2:
        l = samples_nc[:, 0].astype(np.float32, copy=False)
        r = samples_nc[:, 1].astype(np.float32, copy=False)
        if mono_downmix:
            return [("mono", 0.5 * (l + r))]
        return [("left", l), ("right", r)]
    raise ValueError(f"Unsupported channel count: {c}")


# ----------------------------------------------------------------------------------------
# section 8f rank 2: the bundle on-disk format      (reference include/analysis/recorder.hpp:49-126)
# ----------------------------------------------------------------------------------------


def recorder_float_to_pcm16(x: np.ndarray) -> np.ndarray:
    """recorder.hpp:49-53: clamp to [-1, 1] in float32, multiply by 32767.0f in float32, truncate toward zero."""
    c = np.clip(np.asarray(x, dtype=np.float32), np.float32(-1.0), np.float32(1.0))
    return np.trunc(c * np.float32(32767.0)).astype(np.int16)


def recorder_wav_bytes(stereo_interleaved: np.ndarray, sample_rate: int = SR) -> bytes:
    """recorder.hpp:55-90: 44-byte RIFF/WAVE header (PCM, 2 channels, 16 bit, block align 4) + interleaved int16."""
    import struct
    v = np.asarray(stereo_interleaved, dtype=np.float32).reshape(-1)
    frames = v.size // 2
    data_bytes = frames * 4
    hdr = (b"RIFF" + struct.pack("<I", 36 + data_bytes) + b"WAVEfmt "
           + struct.pack("<IHHIIHH", 16, 1, 2, sample_rate, sample_rate * 4, 4, 16)
           + b"data" + struct.pack("<I", data_bytes))
    return hdr + recorder_float_to_pcm16(v[: frames * 2]).astype("<i2").tobytes()


def recorder_meta_json(sample_rate: int, length_samples: int, tap_names: Sequence[str]) -> str:
    """recorder.hpp:112-125: taps in std::map (byte-wise sorted) order."""
    names = ", ".join(f'"{n}"' for n in sorted(tap_names, key=lambda t: t.encode()))
    return ("{\n" + f'  "sample_rate_hz": {sample_rate},\n' + f'  "length_samples": {length_samples},\n'
            + f'  "taps": [{names}]\n' + "}\n")


def wav_pcm16_payload(blob: bytes) -> Tuple[int, np.ndarray]:
    """What io.py:200 (scipy.io.wavfile.read, third-party, version recorded in goldens.json) returns for a 16-bit PCM
    file: (sample rate, int16 array (frames, channels) or (frames,) for mono).  RIFF chunks are word aligned."""
    import struct
    if blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError("not a RIFF/WAVE file")
    pos, fmt = 12, None
    while pos + 8 <= len(blob):
        tag, size = blob[pos : pos + 4], struct.unpack("<I", blob[pos + 4 : pos + 8])[0]
        body = blob[pos + 8 : pos + 8 + size]
        if tag == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
        elif tag == b"data":
            if fmt is None or fmt[0] != 1 or fmt[5] != 16:
                raise ValueError("not 16-bit PCM")
            ch = fmt[1]
            a = np.frombuffer(body[: (len(body) // (2 * ch)) * 2 * ch], dtype="<i2")
            return int(fmt[2]), (a.reshape(-1, ch) if ch > 1 else a)
        pos += 8 + size + (size & 1)
    raise ValueError("no data chunk")


# ----------------------------------------------------------------------------------------
# a2  time-selection prologue (duplicated in every reference module)
# ----------------------------------------------------------------------------------------


def peak_index(x: np.ndarray) -> int:
    """argmax(|x|), first maximum wins (decay.py:136, spectrogram.py:181, zplane.py:197 ...)."""
    return int(np.argmax(np.abs(x)))


def select_segment(
    n: int,
    peak: int,
    sr: int,
    trim_to_peak: bool = True,
    ignore_leading_seconds: float = 0.0,
    duration_seconds: Optional[float] = None,
) -> Tuple[int, int]:
    """
    (start, length) of the analysed slice.
    decay.py:135-144, spectrogram.py:180-194, waterfall.py:358-372, modalcloud.py:298-312,
    frequency_response.py:185-199, filterplot.py:124-138.
    """
    start = 0
    length = n
    if trim_to_peak:
        start = peak
        length = n - peak
    if ignore_leading_seconds > 0.0:
        ig = int(round(float(ignore_leading_seconds) * float(sr)))
        ig = max(0, min(ig, length))
        start += ig
        length -= ig
    if duration_seconds is not None:
        d = int(round(float(duration_seconds) * float(sr)))
        d = max(0, min(d, length))
        length = d
    return start, length


# ----------------------------------------------------------------------------------------
# a3-a6  Schroeder EDC, crossings, line fits                       (reference analyse/decay.py)
# ----------------------------------------------------------------------------------------


def schroeder_edc_db(
    x: np.ndarray,
    sr: int,
    trim_to_peak: bool = True,
    ignore_leading_seconds: float = 0.0,
    floor_db: float = -120.0,
    eps: float = 1e-20,
    smoothing_window: int = 0,
) -> Tuple[np.ndarray, np.ndarray, int]:
    """decay.py:115-170.  Returns (time_seconds f32, edc_db f32, start_index)."""
    if x.ndim != 1:
        raise ValueError("compute_schroeder_edc_db expects a 1D mono array.")
    pk = peak_index(x) if trim_to_peak else 0
    start, length = select_segment(x.size, pk, sr, trim_to_peak, ignore_leading_seconds, None)
    seg = x[start : start + length].astype(np.float64)
    if seg.size < 4:
        raise ValueError("Not enough samples after trimming/ignoring to compute EDC.")
    e = seg * seg
    edc = np.cumsum(e[::-1])[::-1]                      # decay.py:151 (sequential f64, from the end)
    edc = np.maximum(edc, float(eps))                   # :154
    edc = edc / edc[0]                                  # :157
    db = 10.0 * np.log10(edc)                           # :158
    if smoothing_window and smoothing_window > 1:       # :161-164
        w = int(smoothing_window)
        db = np.convolve(db, np.ones(w, dtype=np.float64) / float(w), mode="same")
    db = np.maximum(db, float(floor_db)).astype(np.float32)   # :167
    t = (np.arange(db.size, dtype=np.float32) / float(sr)).astype(np.float32)  # :169
    return t, db, start


def crossing_time(t: np.ndarray, y: np.ndarray, target_db: float) -> Optional[float]:
    """decay.py:173-199 (same body at modalcloud.py:215-235): first y<=target, linear interpolation."""
    below = y <= target_db                              # float32 compare (NEP 50 weak python float)
    if not np.any(below):
        return None
    i = int(np.argmax(below))
    if i == 0:
        return float(t[0])
    t0, t1 = float(t[i - 1]), float(t[i])
    y0, y1 = float(y[i - 1]), float(y[i])
    if y1 == y0:
        return t1
    frac = (float(target_db) - y0) / (y1 - y0)
    frac = float(np.clip(frac, 0.0, 1.0))
    return t0 + frac * (t1 - t0)


def fit_decay(
    t: np.ndarray,
    y: np.ndarray,
    range_db: Tuple[float, float],
    lower_limit_db: float,
    min_points: int = 8,
) -> Optional[Dict[str, float]]:
    """decay.py:202-260 (min_points=8) and modalcloud.py:238-281 (min_points=10)."""
    hi, lo = float(range_db[0]), float(range_db[1])
    if lo > hi:
        raise ValueError("range_db should be (higher_db, lower_db), e.g. (-5, -25).")
    lo_eff = max(lo, float(lower_limit_db))
    ts = crossing_time(t, y, hi)
    te = crossing_time(t, y, lo_eff)
    if ts is None or te is None or te <= ts:
        return None
    m = (t >= ts) & (t <= te)                           # float32 compares
    npts = int(np.sum(m))
    if npts < int(min_points):
        return None
    tt = t[m].astype(np.float64)
    yy = y[m].astype(np.float64)
    a = np.column_stack([tt, np.ones_like(tt)])
    coef, _, _, _ = np.linalg.lstsq(a, yy, rcond=None)
    slope, icpt = float(coef[0]), float(coef[1])
    if slope >= 0.0:
        return None
    pred = slope * tt + icpt
    ss_res = float(np.sum((yy - pred) ** 2))
    ss_tot = float(np.sum((yy - np.mean(yy)) ** 2))
    r2 = 1.0 - (ss_res / ss_tot) if ss_tot > 0.0 else 0.0
    return dict(
        range_hi=hi, range_lo=lo, start_t=float(ts), end_t=float(te), slope=slope,
        intercept=icpt, r2=float(r2), rt60=float(-60.0 / slope), npts=npts,
    )


DECAY_DEFAULTS = dict(
    trim_to_peak=True, ignore_leading_seconds=0.0, edc_floor_db=-120.0, edc_epsilon=1e-20,
    fit_lower_limit_db=-80.0, t20_range_db=(-5.0, -25.0), t30_range_db=(-5.0, -35.0),
    compute_edt=False, edt_range_db=(0.0, -10.0), edc_smoothing_window_samples=0,
)


def analyse_decay(x: np.ndarray, sr: int = SR, **kw) -> Dict:
    """decay.py:268-329."""
    s = dict(DECAY_DEFAULTS); s.update(kw)
    t, db, start = schroeder_edc_db(
        x, sr, s["trim_to_peak"], s["ignore_leading_seconds"], s["edc_floor_db"], s["edc_epsilon"],
        s["edc_smoothing_window_samples"],
    )
    t0 = crossing_time(t, db, 0.0)
    t10 = crossing_time(t, db, -10.0)
    early = float(t10 - t0) if (t0 is not None and t10 is not None and t10 >= t0) else None
    fits: Dict[str, Dict] = {}
    order = (["EDT"] if s["compute_edt"] else []) + ["T20", "T30"]
    for name in order:
        rng = {"EDT": s["edt_range_db"], "T20": s["t20_range_db"], "T30": s["t30_range_db"]}[name]
        f = fit_decay(t, db, rng, s["fit_lower_limit_db"], 8)
        if f is not None:
            fits[name] = f
    return dict(start=start, time_seconds=t, edc_db=db, early_10db=early, fits=fits)


# ----------------------------------------------------------------------------------------
# a7-a10  RT60 by band                                         (reference analyse/rt60bands.py)
# ----------------------------------------------------------------------------------------

RT60_DEFAULTS = dict(
    band_mode="three", low_upper_hz=250.0, mid_center_hz=1000.0, mid_width_octaves=2.0,
    high_lower_hz=4000.0, f_min_hz=31.5, f_max_hz=16000.0, transition_width_octaves=1.0 / 6.0,
    include_t20=False, include_edt=False,
)


def band_definitions(sr: int = SR, **kw) -> List[Dict]:
    """rt60bands.py:183-264.  Each band: name, centre_hz, kind, low_edge_hz, high_edge_hz."""
    s = dict(RT60_DEFAULTS); s.update(kw)
    nyq = 0.5 * float(sr)
    mode = str(s["band_mode"]).lower()
    if mode == "three":                                 # :183-205
        low_upper = float(np.clip(s["low_upper_hz"], 20.0, nyq))
        mid_c = float(np.clip(s["mid_center_hz"], 20.0, nyq))
        half = 0.5 * float(max(0.1, s["mid_width_octaves"]))
        mid_lo = float(np.clip(mid_c / float(2.0 ** half), 20.0, nyq))
        mid_hi = float(np.clip(mid_c * float(2.0 ** half), 20.0, nyq))
        high_lower = float(np.clip(s["high_lower_hz"], 20.0, nyq))
        return [
            dict(name="Low", centre_hz=float(np.sqrt(20.0 * low_upper)), kind="lowpass",
                 low_edge_hz=None, high_edge_hz=low_upper),
            dict(name="Mid", centre_hz=mid_c, kind="bandpass", low_edge_hz=mid_lo, high_edge_hz=mid_hi),
            dict(name="High", centre_hz=float(np.sqrt(max(20.0, high_lower) * nyq)), kind="highpass",
                 low_edge_hz=high_lower, high_edge_hz=None),
        ]
    if mode in ("octave", "third"):                     # :208-253
        per_oct = 1.0 if mode == "octave" else 3.0
        f_min = float(max(20.0, min(s["f_min_hz"], nyq)))
        f_max = float(max(f_min, min(s["f_max_hz"], nyq)))
        step = 2.0 ** (1.0 / per_oct)
        half_band = 2.0 ** (1.0 / (2.0 * per_oct))
        k_lo = int(np.floor(np.log(f_min / 1000.0) / np.log(step)))
        k_hi = int(np.ceil(np.log(f_max / 1000.0) / np.log(step)))
        out = []
        for k in range(k_lo, k_hi + 1):
            fc = 1000.0 * (step ** float(k))
            if fc < f_min or fc > f_max:
                continue
            lo = float(np.clip(fc / half_band, 20.0, nyq))
            hi = float(np.clip(fc * half_band, 20.0, nyq))
            if hi <= lo:
                continue
            out.append(dict(name=f"{int(round(fc))}Hz", centre_hz=float(fc), kind="bandpass",
                            low_edge_hz=lo, high_edge_hz=hi))
        out.sort(key=lambda b: b["centre_hz"])
        return out
    raise ValueError(f"Unknown band_mode: {s['band_mode']}")


def _ramp(f: np.ndarray, x0: float, x1: float) -> np.ndarray:
    """rt60bands.py:116-124: half-cosine 0->1 between x0 and x1, float32 arithmetic on a float32 axis."""
    if x1 <= x0:
        return (f >= x1).astype(np.float32)
    u = np.clip((f - x0) / (x1 - x0), 0.0, 1.0)
    return (0.5 - 0.5 * np.cos(np.pi * u)).astype(np.float32)


def lowpass_mask(f: np.ndarray, pass_hz: float, trans_oct: float, nyq: float) -> np.ndarray:
    """rt60bands.py:127-137."""
    pass_hz = float(np.clip(pass_hz, 1.0, nyq))
    stop_hz = float(min(nyq, pass_hz * float(2.0 ** float(trans_oct))))
    if stop_hz <= pass_hz:
        stop_hz = min(nyq, pass_hz + 1.0)
    m = 1.0 - _ramp(f, pass_hz, stop_hz)
    m[f <= pass_hz] = 1.0
    m[f >= stop_hz] = 0.0
    return m.astype(np.float32)


def highpass_mask(f: np.ndarray, pass_hz: float, trans_oct: float, nyq: float) -> np.ndarray:
    """rt60bands.py:140-150."""
    pass_hz = float(np.clip(pass_hz, 1.0, nyq))
    stop_hz = float(max(1.0, pass_hz / float(2.0 ** float(trans_oct))))
    if pass_hz <= stop_hz:
        stop_hz = max(1.0, pass_hz - 1.0)
    m = _ramp(f, stop_hz, pass_hz)
    m[f <= stop_hz] = 0.0
    m[f >= pass_hz] = 1.0
    return m.astype(np.float32)


def band_mask(f: np.ndarray, band: Dict, trans_oct: float, nyq: float) -> np.ndarray:
    """rt60bands.py:153-167 and the dispatch at :362-389."""
    kind = band["kind"]
    if kind == "lowpass":
        return lowpass_mask(f, band["high_edge_hz"], trans_oct, nyq)
    if kind == "highpass":
        return highpass_mask(f, band["low_edge_hz"], trans_oct, nyq)
    if kind == "bandpass":
        lo = float(np.clip(band["low_edge_hz"], 1.0, nyq))
        hi = float(np.clip(band["high_edge_hz"], 1.0, nyq))
        if hi <= lo:
            return np.zeros_like(f, dtype=np.float32)
        return (highpass_mask(f, lo, trans_oct, nyq) * lowpass_mask(f, hi, trans_oct, nyq)).astype(np.float32)
    raise ValueError(f"Unknown band kind: {kind}")


def analyse_rt60_bands(x: np.ndarray, sr: int = SR, decay: Optional[Dict] = None, **kw) -> Dict:
    """
    rt60bands.py:324-413.  The forward rFFT is taken once here (the reference recomputes it per
    band at :170-175 with identical inputs, so the spectrum is bit-identical).
    """
    s = dict(RT60_DEFAULTS); s.update(kw)
    d = dict(DECAY_DEFAULTS); d.update(decay or {})
    full = x.astype(np.float32, copy=False)
    pk = peak_index(full) if d["trim_to_peak"] else 0
    ig = 0
    if d["ignore_leading_seconds"] > 0.0:
        ig = int(round(d["ignore_leading_seconds"] * float(sr)))
        ig = max(0, min(ig, full.size))
    start = min(full.size, pk + ig)
    n = int(full.size)
    if n < 8:
        raise ValueError("Not enough samples for rt60bands analysis.")
    nyq = 0.5 * float(sr)
    freqs = np.fft.rfftfreq(n, d=1.0 / float(sr)).astype(np.float32)
    bands = band_definitions(sr, **s)
    spec = np.fft.rfft(full.astype(np.float64, copy=False))
    metrics: Dict[str, Dict] = {}
    for b in bands:
        m = band_mask(freqs, b, s["transition_width_octaves"], nyq)
        y = np.fft.irfft(spec * m.astype(np.float64), n=n).astype(np.float32)    # :170-175
        y = y[start:]
        if y.size < 8:
            metrics[b["name"]] = dict(t30=None, t20=None, edt=None)
            continue
        t, db, _ = schroeder_edc_db(y, sr, False, 0.0, d["edc_floor_db"], d["edc_epsilon"],
                                    d["edc_smoothing_window_samples"])
        f30 = fit_decay(t, db, d["t30_range_db"], d["fit_lower_limit_db"], 8)
        f20 = fit_decay(t, db, d["t20_range_db"], d["fit_lower_limit_db"], 8) if s["include_t20"] else None
        fed = fit_decay(t, db, d["edt_range_db"], d["fit_lower_limit_db"], 8) if s["include_edt"] else None
        metrics[b["name"]] = dict(
            t30=None if f30 is None else f30["rt60"],
            t20=None if f20 is None else f20["rt60"],
            edt=None if fed is None else fed["rt60"],
        )
    return dict(start=start, bands=bands, metrics=metrics)


# ----------------------------------------------------------------------------------------
# a11-a12  STFT magnitude in dB              (reference spectrogram.py:107-160 and its two copies)
# ----------------------------------------------------------------------------------------


def stft_mag_db(
    x: np.ndarray, sr: int, n_fft: int, hop: int, use_hann: bool = True, floor_db: float = -120.0,
    frame_indices: Optional[np.ndarray] = None,
) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """
    spectrogram.py:107-160 == waterfall.py:188-230 == modalcloud.py:121-158.
    Valid framing, symmetric Hann (np.hanning), |rfft| -> max(., 10^(floor/20)) -> 20 log10 -> f32.
    The per-frame Python loop of the reference is replaced by one strided batched rfft; pocketfft
    transforms every row independently so the bits are the same.
    `frame_indices` (optional) restricts the computed columns (used by the waterfall shortcut).
    """
    if x.ndim != 1:
        raise ValueError("_compute_stft_magnitude_db expects a 1D mono array.")
    if n_fft <= 0 or hop <= 0:
        raise ValueError("n_fft and hop_length must be positive.")
    if x.size < n_fft:
        raise ValueError("Not enough samples for STFT (need at least n_fft).")
    xx = x.astype(np.float64, copy=False)
    t_frames = 1 + (xx.size - n_fft) // hop
    w = np.hanning(n_fft).astype(np.float64) if use_hann else np.ones(n_fft, dtype=np.float64)
    freq = np.fft.rfftfreq(n_fft, d=1.0 / float(sr)).astype(np.float32)
    frames = np.lib.stride_tricks.sliding_window_view(xx, n_fft)[::hop][:t_frames]
    if frame_indices is not None:
        frames = frames[np.asarray(frame_indices, dtype=np.int64)]
    mag_floor = 10.0 ** (float(floor_db) / 20.0)
    out = np.empty((freq.size, frames.shape[0]), dtype=np.float32)
    blk = 256
    for i in range(0, frames.shape[0], blk):
        sp = np.fft.rfft(frames[i : i + blk] * w, axis=1)
        mg = np.maximum(np.abs(sp), mag_floor)
        out[:, i : i + blk] = (20.0 * np.log10(mg)).astype(np.float32).T
    tt = (np.arange(t_frames, dtype=np.float32) * float(hop) / float(sr)).astype(np.float32)
    return tt, freq, out


STFT_DEFAULTS = dict(
    trim_to_peak=True, ignore_leading_seconds=0.0, analysis_duration_seconds=None,
    n_fft=4096, hop_length=512, use_hann_window=True, floor_db=-120.0,
)


def _stft_segment(x: np.ndarray, sr: int, s: Dict, what: str) -> Tuple[np.ndarray, int]:
    """spectrogram.py:177-198 (f64 slice -> f32 copy), same in waterfall.py:355-376, modalcloud.py:295-316."""
    pk = peak_index(x) if s["trim_to_peak"] else 0
    start, length = select_segment(x.size, pk, sr, s["trim_to_peak"], s["ignore_leading_seconds"],
                                   s["analysis_duration_seconds"])
    seg = x.astype(np.float64, copy=False)[start : start + length].astype(np.float32)
    if seg.size < s["n_fft"]:
        raise ValueError(f"Not enough samples after trimming/selection for {what} (need at least n_fft).")
    return seg, start


def analyse_spectrogram(x: np.ndarray, sr: int = SR, **kw) -> Dict:
    """spectrogram.py:168-217."""
    if x.ndim != 1:
        raise ValueError("analyse_spectrogram_for_channel expects a 1D mono array.")
    s = dict(STFT_DEFAULTS); s.update(kw)
    seg, start = _stft_segment(x, sr, s, "spectrogram")
    t, f, m = stft_mag_db(seg, sr, int(s["n_fft"]), int(s["hop_length"]), bool(s["use_hann_window"]),
                          float(s["floor_db"]))
    return dict(start=start, length=int(seg.size), time_seconds=t, frequency_hz=f, magnitude_db=m)


# ----------------------------------------------------------------------------------------
# a13-a14  Waterfall                                            (reference analyse/waterfall.py)
# ----------------------------------------------------------------------------------------

WATERFALL_DEFAULTS = dict(
    STFT_DEFAULTS, f_min_hz=20.0, f_max_hz=20000.0, slice_mode="auto", num_slices=18,
    slice_spacing_seconds=0.05, start_time_seconds=0.0, end_time_seconds=None,
    db_reference="global_max", smoothing_log_bins=0, log_bins_per_octave=96, dynamic_range_db=80.0,
)


def frame_times(t_frames: int, hop: int, sr: int) -> np.ndarray:
    """Frame-start times, float32 arithmetic (spectrogram.py:158)."""
    return (np.arange(t_frames, dtype=np.float32) * float(hop) / float(sr)).astype(np.float32)


def select_slice_frames(ft: np.ndarray, **kw) -> np.ndarray:
    """waterfall.py:233-286 -> ordered unique int32 frame indices."""
    s = dict(WATERFALL_DEFAULTS); s.update(kw)
    if ft.size == 0:
        return np.zeros((0,), dtype=np.int32)
    t0 = float(max(0.0, s["start_time_seconds"]))
    t1 = float(s["end_time_seconds"]) if s["end_time_seconds"] is not None else float(ft[-1])
    if t1 <= t0:
        t1 = float(ft[-1])
    inr = (ft >= t0) & (ft <= t1)
    if not np.any(inr):
        return np.zeros((0,), dtype=np.int32)
    lo = int(np.argmax(inr))
    hi = int(np.max(np.nonzero(inr)))
    mode = str(s["slice_mode"]).lower()
    if mode == "uniform_frames":
        return np.unique(np.linspace(lo, hi, int(max(1, s["num_slices"])), dtype=np.int32))
    if mode == "uniform_time":
        targets = np.arange(t0, t1 + 1e-9, float(max(1e-4, s["slice_spacing_seconds"])), dtype=np.float64)
        fallback = [lo, hi]
    else:
        targets = np.linspace(t0, t1, int(max(2, s["num_slices"])), dtype=np.float64)
        fallback = []
    picked = []
    for tv in targets:
        j = int(np.argmin(np.abs(ft - float(tv))))      # float32 difference, first minimum
        if lo <= j <= hi:
            picked.append(j)
    if not picked:
        picked = fallback
    return np.unique(np.array(picked, dtype=np.int32))


def smooth_db_log_frequency(freq: np.ndarray, mag_db: np.ndarray, f_min: float, f_max: float,
                            window_bins: int, bins_per_oct: int, waterfall_variant: bool) -> np.ndarray:
    """
    frequency_response.py:117-169 (waterfall_variant=False) / waterfall.py:140-185 (True).
    Optional (default off in both modules); host-side post-processing in the product too.
    The two reference copies differ only in an intermediate float32 round trip of the gridded curve.
    """
    if window_bins <= 1:
        return mag_db.astype(np.float32, copy=False) if waterfall_variant else mag_db
    f = freq.astype(np.float64)
    m = mag_db.astype(np.float64)
    lo = float(max(1.0, f_min)); hi = float(max(lo, f_max))
    sel = (f >= lo) & (f <= hi)
    if not np.any(sel):
        return mag_db.astype(np.float32, copy=False) if waterfall_variant else mag_db
    fs, ms = f[sel], m[sel]
    l0, l1 = float(np.log2(fs[0])), float(np.log2(fs[-1]))
    bpo = int(max(16, bins_per_oct))
    nb = int(max(8, np.ceil((l1 - l0) * bpo))) + 1
    grid = 2.0 ** np.linspace(l0, l1, nb, dtype=np.float64)
    g = np.interp(grid, fs, ms)
    ker = np.ones(int(window_bins), dtype=np.float64) / float(window_bins)
    if waterfall_variant:
        g = np.convolve(g.astype(np.float32).astype(np.float64), ker, mode="same").astype(np.float32).astype(np.float64)
    else:
        g = np.convolve(g, ker, mode="same")
    back = np.interp(fs, grid, g)
    out = mag_db.astype(np.float32, copy=True)
    out[sel] = back.astype(np.float32)
    return out


def analyse_waterfall(x: np.ndarray, sr: int = SR, **kw) -> Dict:
    """
    waterfall.py:349-410 with :289-341.  Only the selected frames' spectra are computed (the
    reference computes all T frames at :378-385 and then indexes <=18 of them at :309; frames are
    independent so the selected columns are bit-identical).
    """
    s = dict(WATERFALL_DEFAULTS); s.update(kw)
    seg, start = _stft_segment(x, sr, s, "waterfall")
    n_fft, hop = int(s["n_fft"]), int(s["hop_length"])
    t_frames = 1 + (seg.size - n_fft) // hop
    ft = frame_times(t_frames, hop, sr)
    idx = select_slice_frames(ft, **s)
    if idx.size < 2:
        raise ValueError("Not enough slices selected for waterfall (increase duration or num_slices).")
    _, freq, mag = stft_mag_db(seg, sr, n_fft, hop, bool(s["use_hann_window"]), float(s["floor_db"]),
                               frame_indices=idx)
    nyq = float(freq[-1]) if freq.size else 0.0
    f_lo = float(np.clip(s["f_min_hz"], 1.0, nyq))
    f_hi = float(np.clip(s["f_max_hz"], f_lo, nyq))
    fm = (freq >= f_lo) & (freq <= f_hi)
    if not np.any(fm):
        raise ValueError("Waterfall frequency selection is empty (check f_min_hz/f_max_hz).")
    f_sel = freq[fm].astype(np.float32)
    sl = mag[fm].T.astype(np.float32)                   # (S, Fsel)
    if s["smoothing_log_bins"] and int(s["smoothing_log_bins"]) > 1:
        sl = np.stack([
            smooth_db_log_frequency(f_sel, row, f_lo, f_hi, int(s["smoothing_log_bins"]),
                                    int(s["log_bins_per_octave"]), True) for row in sl
        ], axis=0).astype(np.float32)
    if str(s["db_reference"]).lower() == "slice_max":
        rel = sl - np.max(sl, axis=1, keepdims=True)
    else:
        rel = sl - float(np.max(sl))
    dyn = float(max(10.0, s["dynamic_range_db"]))
    rel = np.clip(rel, -dyn, 0.0).astype(np.float32)
    return dict(start=start, length=int(seg.size), frame_indices=idx.astype(np.int32),
                slice_times_seconds=ft[idx].astype(np.float32), frequency_hz=f_sel, slice_rel_db=rel)


# ----------------------------------------------------------------------------------------
# a15-a16  Modal cloud                                         (reference analyse/modalcloud.py)
# ----------------------------------------------------------------------------------------

MODAL_DEFAULTS = dict(
    STFT_DEFAULTS, n_fft=8192, f_min_hz=20.0, f_max_hz=20000.0, log_bins_per_octave=24, min_bins=24,
    fit_lower_limit_db=-80.0, t30_range_db=(-5.0, -35.0), t20_range_db=(-5.0, -25.0),
    edt_range_db=(0.0, -10.0), metric="t30", min_fit_points=10, min_peak_db_above_floor=20.0,
)


def log_bin_edges(f_min: float, f_max: float, bins_per_oct: int, min_bins: int) -> np.ndarray:
    """modalcloud.py:166-173 -> float32 edges (B+1,)."""
    lo = float(max(1.0, f_min))
    hi = float(max(lo * 1.001, f_max))
    octs = float(math.log2(hi / lo))
    nb = int(max(min_bins, math.ceil(octs * float(max(4, bins_per_oct)))))
    return (lo * (2.0 ** np.linspace(0.0, octs, nb + 1, dtype=np.float64))).astype(np.float32)


def log_bin_membership(freq_sel: np.ndarray, edges_f32: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """
    Integer restatement of the mask at modalcloud.py:197-200: for log bin b the rFFT rows with
    lo <= f < hi (float32 compares).  Returns (centres f32, first_row[b], row_count[b]).
    freq_sel is monotone so each mask is a contiguous run.
    """
    e = edges_f32.astype(np.float64)
    centres = np.sqrt(e[:-1] * e[1:]).astype(np.float32)
    first = np.zeros(centres.size, dtype=np.int64)
    count = np.zeros(centres.size, dtype=np.int64)
    for b in range(centres.size):
        m = (freq_sel >= float(e[b])) & (freq_sel < float(e[b + 1]))
        if np.any(m):
            nz = np.nonzero(m)[0]
            first[b] = int(nz[0]); count[b] = int(nz.size)
    return centres, first, count


def aggregate_log_bins(freq_sel: np.ndarray, mag_db_sel: np.ndarray, edges_f32: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """modalcloud.py:176-207: mean of 10^(dB/20) (f64) per log bin -> 20 log10(max(.,1e-30)) -> f32; empty -> NaN."""
    centres, first, count = log_bin_membership(freq_sel, edges_f32)
    lin = 10.0 ** (mag_db_sel.astype(np.float64) / 20.0)
    out = np.full((centres.size, mag_db_sel.shape[1]), np.nan, dtype=np.float32)
    for b in range(centres.size):
        if count[b] == 0:
            continue
        m = np.mean(lin[first[b] : first[b] + count[b], :], axis=0)
        out[b, :] = (20.0 * np.log10(np.maximum(m, 1e-30))).astype(np.float32)
    return centres, out


def analyse_modal_cloud(x: np.ndarray, sr: int = SR, **kw) -> Dict:
    """modalcloud.py:289-391 -> points (centre_hz, rt60, r2) sorted by frequency."""
    s = dict(MODAL_DEFAULTS); s.update(kw)
    seg, start = _stft_segment(x, sr, s, "modal cloud")
    t, freq, mag = stft_mag_db(seg, sr, int(s["n_fft"]), int(s["hop_length"]), bool(s["use_hann_window"]),
                               float(s["floor_db"]))
    nyq = 0.5 * float(sr)
    f_lo = float(np.clip(s["f_min_hz"], 1.0, nyq))
    f_hi = float(np.clip(s["f_max_hz"], f_lo, nyq))
    fm = (freq >= f_lo) & (freq <= f_hi)
    edges = log_bin_edges(f_lo, f_hi, int(s["log_bins_per_octave"]), int(s["min_bins"]))
    centres, curves = aggregate_log_bins(freq[fm], mag[fm, :], edges)
    metric = str(s["metric"]).lower()
    if metric == "t20":
        rng = s["t20_range_db"]
    elif metric == "edt":
        rng = s["edt_range_db"]
    else:
        metric = "t30"; rng = s["t30_range_db"]
    pts = []
    for b in range(centres.size):
        c = curves[b, :]
        if not np.all(np.isfinite(c)):
            continue
        pk = float(np.max(c))
        if (pk - float(s["floor_db"])) < float(s["min_peak_db_above_floor"]):
            continue
        rel = (c - pk).astype(np.float32)
        f = fit_decay(t, rel, rng, float(s["fit_lower_limit_db"]), int(s["min_fit_points"]))
        if f is None:
            continue
        pts.append((float(centres[b]), float(f["rt60"]), float(f["r2"])))
    pts.sort(key=lambda p: p[0])
    return dict(start=start, length=int(seg.size), metric=metric, points=pts, centres=centres, curves=curves)


# ----------------------------------------------------------------------------------------
# a17-a18  Whole-segment spectrum: frequency response, filter response
# ----------------------------------------------------------------------------------------

FR_DEFAULTS = dict(
    trim_to_peak=True, ignore_leading_seconds=0.0, analysis_duration_seconds=None, use_hann_window=True,
    magnitude_floor_db=-120.0, f_min_hz=20.0, f_max_hz=20000.0, smoothing_log_bins=0,
    log_bins_per_octave=96, phase_mode="degrees", unwrap_phase=True,
)


def _spectrum(x: np.ndarray, sr: int, s: Dict, what: str):
    pk = peak_index(x) if s["trim_to_peak"] else 0
    start, length = select_segment(x.size, pk, sr, s["trim_to_peak"], s["ignore_leading_seconds"],
                                   s["analysis_duration_seconds"])
    seg = x.astype(np.float64, copy=False)[start : start + length]
    if seg.size < 32:
        raise ValueError(f"Not enough samples after trimming/selection to analyse {what}.")
    if s["use_hann_window"]:
        seg = seg * np.hanning(seg.size).astype(np.float64)
    spec = np.fft.rfft(seg)
    mag = np.maximum(np.abs(spec).astype(np.float64), 10.0 ** (float(s["magnitude_floor_db"]) / 20.0))
    mag_db = (20.0 * np.log10(mag)).astype(np.float32)
    freq = np.fft.rfftfreq(length, d=1.0 / float(sr)).astype(np.float32)
    return start, length, spec, mag_db, freq


def analyse_frequency_response(x: np.ndarray, sr: int = SR, **kw) -> Dict:
    """frequency_response.py:173-271."""
    if x.ndim != 1:
        raise ValueError("analyse_frequency_response_for_channel expects a 1D mono array.")
    s = dict(FR_DEFAULTS); s.update(kw)
    start, length, _, mag_db, freq = _spectrum(x, sr, s, "spectrum")
    nyq = 0.5 * float(sr)
    if s["smoothing_log_bins"] and int(s["smoothing_log_bins"]) > 1:
        lo_s = float(np.clip(s["f_min_hz"], 1.0, nyq)); hi_s = float(np.clip(s["f_max_hz"], lo_s, nyq))
        mag_db = smooth_db_log_frequency(freq, mag_db, lo_s, hi_s, int(s["smoothing_log_bins"]),
                                         int(s["log_bins_per_octave"]), False)
    lo = float(np.clip(s["f_min_hz"], 0.0, nyq)); hi = float(np.clip(s["f_max_hz"], lo, nyq))
    m = (freq >= lo) & (freq <= hi)
    if not np.any(m):
        raise ValueError("Selected frequency range is empty (check f_min_hz/f_max_hz).")
    fs, ds = freq[m], mag_db[m]
    lin = 10.0 ** (ds.astype(np.float64) / 20.0)
    peak_hz = float(fs[int(np.argmax(ds))])
    wsum = float(np.sum(lin))
    cen = float(np.sum(fs.astype(np.float64) * lin) / wsum) if wsum > 0.0 else float(fs[0])
    return dict(start=start, length=length, frequency_hz=freq.astype(np.float32),
                magnitude_db=mag_db.astype(np.float32), peak_hz=peak_hz, centroid_hz=cen)


def analyse_filter_response(x: np.ndarray, sr: int = SR, **kw) -> Dict:
    """filterplot.py:112-203."""
    if x.ndim != 1:
        raise ValueError("analyse_filter_response_for_channel expects a 1D mono array.")
    s = dict(FR_DEFAULTS); s.update(kw)
    start, length, spec, mag_db, freq = _spectrum(x, sr, s, "filter response")
    ph = np.angle(spec).astype(np.float64)
    if s["unwrap_phase"]:
        ph = np.unwrap(ph)
    phase = np.rad2deg(ph).astype(np.float32) if s["phase_mode"] == "degrees" else ph.astype(np.float32)
    nyq = 0.5 * float(sr)
    lo = float(np.clip(s["f_min_hz"], 0.0, nyq)); hi = float(np.clip(s["f_max_hz"], lo, nyq))
    m = (freq >= lo) & (freq <= hi)
    if not np.any(m):
        raise ValueError("Selected frequency range is empty.")
    peak_hz = float(freq[m][int(np.argmax(mag_db[m]))])
    i1k = int(np.argmin(np.abs(freq - 1000.0)))
    return dict(start=start, length=length, frequency_hz=freq, magnitude_db=mag_db, phase=phase,
                peak_hz=peak_hz, mag_1k_db=float(mag_db[i1k]), idx_1k=i1k)


# ----------------------------------------------------------------------------------------
# a19-a22  Z-plane AR pole fit                                     (reference analyse/zplane.py)
# ----------------------------------------------------------------------------------------

ZPLANE_DEFAULTS = dict(
    trim_to_peak=True, ignore_leading_seconds=0.0, analysis_duration_seconds=None, ar_order=256,
    derive_zeros=False, zero_order=64, normalise_segment=True, ridge_lambda=0.0,
)


def zplane_segment(x: np.ndarray, sr: int, s: Dict) -> Tuple[np.ndarray, int]:
    """zplane.py:195-213: index = peak + round(ignore*sr), clamped; optional duration; divide by peak."""
    start = peak_index(x) if s["trim_to_peak"] else 0
    start += int(round(float(s["ignore_leading_seconds"]) * sr))
    start = max(0, min(start, len(x)))
    if s["analysis_duration_seconds"] is None:
        seg = x[start:]
    else:
        ln = int(round(float(s["analysis_duration_seconds"]) * sr))
        seg = x[start : start + max(1, ln)]
    seg = seg.astype(np.float64, copy=False)
    if s["normalise_segment"]:
        pk = float(np.max(np.abs(seg))) if seg.size else 1.0
        if pk > 0.0:
            seg = seg / pk
    return seg, start


def ar_design(x: np.ndarray, p: int) -> Tuple[np.ndarray, np.ndarray]:
    """zplane.py:100-108: rows n=p..N-1, A[n,k-1]=x[n-k], y=-x[n]."""
    n = x.size
    a = np.lib.stride_tricks.sliding_window_view(x, p + 1)[:, ::-1]   # row r = x[r+p], x[r+p-1], ..., x[r]
    return np.ascontiguousarray(a[:, 1:]), -a[:, 0]


def fit_ar(x: np.ndarray, order: int, ridge_lambda: float = 0.0) -> np.ndarray:
    """zplane.py:83-120: covariance-method least squares (gelsd) or ridge normal equations."""
    x = np.asarray(x, dtype=np.float64)
    p = int(order)
    if p < 1:
        return np.array([1.0], dtype=np.float64)
    if x.size <= p:
        p = max(1, x.size - 1)
    a, y = ar_design(x, p)
    if ridge_lambda and ridge_lambda > 0.0:
        g = a.T @ a
        r = a.T @ y
        g.flat[:: p + 1] += float(ridge_lambda)
        rest = np.linalg.solve(g, r)
    else:
        rest, *_ = np.linalg.lstsq(a, y, rcond=None)
    return np.concatenate(([1.0], rest))


def poly_roots(c: np.ndarray) -> np.ndarray:
    """zplane.py:145-158: strip trailing |c|<1e-14, then numpy.roots (companion eigenvalues)."""
    c = np.asarray(c, dtype=np.float64)
    while c.size > 1 and abs(c[-1]) < 1e-14:
        c = c[:-1]
    if c.size <= 1:
        return np.array([], dtype=np.complex128)
    return np.roots(c)


def fir_numerator(a: np.ndarray, h: np.ndarray, zero_order: int) -> np.ndarray:
    """zplane.py:123-142: b[n] = sum_k a[k] h[n-k], n = 0..Q (a truncated convolution)."""
    q = int(max(0, zero_order))
    h = np.asarray(h, dtype=np.float64)
    b = np.zeros(q + 1, dtype=np.float64)
    for n in range(q + 1):
        acc = 0.0
        for k in range(0, len(a)):
            if 0 <= n - k < h.size:
                acc += a[k] * h[n - k]
        b[n] = acc
    return b


def rt60_from_radius(r: float, sr: int) -> float:
    """zplane.py:161-173."""
    r = float(r)
    if r <= 0.0 or r >= 1.0:
        return float("inf")
    return float(np.log(1000.0) * ((-1.0 / np.log(r)) / float(sr)))


def analyse_zplane(x: np.ndarray, sr: int = SR, **kw) -> Dict:
    """The numeric body of zplane.py:192-232 plus the statistics of :260-276 / :294-299."""
    s = dict(ZPLANE_DEFAULTS); s.update(kw)
    seg, start = zplane_segment(x, sr, s)
    a = fit_ar(seg, int(s["ar_order"]), float(s["ridge_lambda"]))
    poles = poly_roots(a)
    zeros = poly_roots(fir_numerator(a, seg, int(s["zero_order"]))) if s["derive_zeros"] else None
    out = dict(start=start, a=a, poles=poles, zeros=zeros)
    if poles.size:
        rad = np.abs(poles)
        out.update(max_radius=float(np.max(rad)), median_radius=float(np.median(rad)),
                   unstable=int(np.sum(rad >= 1.0)))
    return out


# ----------------------------------------------------------------------------------------
# section 8f  group delay                      (reference group_delay.py:89-137, :159-171, :209-220)
# ----------------------------------------------------------------------------------------
GD_DEFAULTS = dict(
    trim_to_peak=True, ignore_leading_seconds=0.0, analysis_duration_seconds=None, use_hann_window=True,
    fft_size=None, f_min_hz=20.0, f_max_hz=20000.0, unwrap_phase=True, smoothing_bins=0,
)


def group_delay_segment(n: int, peak: int, sr: int, s: Dict) -> Tuple[int, int]:
    """(start, length) exactly as group_delay.py:159-171 selects it (note: NOT the select_segment rule -- the
    ignore time is added to the start without being clipped to the remaining length first, and a duration of 0
    still takes one sample)."""
    start = peak if s["trim_to_peak"] else 0
    start += int(round(float(s["ignore_leading_seconds"]) * sr))
    start = max(0, min(start, n))
    if s["analysis_duration_seconds"] is None:
        return start, n - start
    want = max(1, int(round(float(s["analysis_duration_seconds"]) * sr)))
    return start, max(0, min(want, n - start))


def group_delay_fft_size(seg_len: int, fft_size: Optional[int]) -> int:
    """group_delay.py:101-107: next power of two >= segment length, capped at 2^20, unless given."""
    if fft_size is not None:
        return int(fft_size)
    p = 1 << (max(1, int(seg_len)) - 1).bit_length()
    return min(p, 1 << 20)


def analyse_group_delay(x: np.ndarray, sr: int = SR, **kw) -> Dict:
    """group_delay.py:89-137: windowed segment -> rfft(n_fft) -> angle -> unwrap -> -d(phase)/dw (np.gradient on the
    rad/sample axis) -> optional moving average ('same' convolution) -> frequency mask."""
    s = dict(GD_DEFAULTS); s.update(kw)
    start, length = group_delay_segment(x.size, peak_index(x), sr, s)
    seg = x[start : start + length].astype(np.float64)
    if s["use_hann_window"]:
        seg = seg * np.hanning(seg.size)
    n_fft = group_delay_fft_size(seg.size, s["fft_size"])
    H = np.fft.rfft(seg, n=n_fft)
    freq = np.fft.rfftfreq(n_fft, d=1.0 / float(sr))
    ph = np.angle(H)
    if s["unwrap_phase"]:
        ph = np.unwrap(ph)
    w = 2.0 * np.pi * (freq / float(sr))
    gd = -np.gradient(ph, w)
    sb = int(s["smoothing_bins"] or 0)
    if sb > 1:
        gd = np.convolve(gd, np.ones(sb, dtype=np.float64) / float(sb), mode="same")
    m = (freq >= float(s["f_min_hz"])) & (freq <= float(s["f_max_hz"]))
    return dict(start=start, length=length, n_fft=n_fft, freq=freq[m], gd=gd[m])


def group_delay_summary(names: List[str], gds: List[np.ndarray]) -> str:
    """group_delay.py:209-220."""
    lines = []
    for name, gd in zip(names, gds):
        if gd.size == 0:
            continue
        lines.append(f"- {name}: gd median={float(np.median(gd)):.3f} samples, "
                     f"p10={float(np.percentile(gd, 10)):.3f}, p90={float(np.percentile(gd, 90)):.3f}")
    if not lines:
        return "No group delay results."
    return "Group delay summary:\n" + "\n".join(lines)


# ----------------------------------------------------------------------------------------
# section 8f  diffusion                        (reference diffusion.py:92-226, :234-376, :457-476)
# ----------------------------------------------------------------------------------------
DIFF_DEFAULTS = dict(
    trim_to_peak=True, ignore_leading_seconds=0.0, window_seconds=0.050, hop_seconds=0.010,
    max_lag_milliseconds=10.0, echo_density_threshold_rms=1.0, echo_density_normalise_to_gaussian=True,
)


def diffusion_trim(x: np.ndarray, sr: int, trim_to_peak: bool, ignore_leading_seconds: float) -> Tuple[np.ndarray, int]:
    """diffusion.py:92-117: float64 view -> argmax|x| -> optional ignore -> back to float32."""
    v = x.astype(np.float64, copy=False)
    start = 0
    if trim_to_peak:
        pk = int(np.argmax(np.abs(v)))
        start += pk
        v = v[pk:]
    if ignore_leading_seconds > 0.0:
        ig = int(round(ignore_leading_seconds * float(sr)))
        ig = max(0, min(ig, v.size))
        start += ig
        v = v[ig:]
    return v.astype(np.float32), start


def diffusion_geometry(n: int, sr: int, s: Dict) -> Tuple[int, int, int, int]:
    """(win, hop, frames, max_lag), diffusion.py:246-258."""
    win = max(16, int(round(s["window_seconds"] * float(sr))))
    hop = max(1, int(round(s["hop_seconds"] * float(sr))))
    frames = 0 if n < win else 1 + (n - win) // hop
    max_lag = max(1, int(round((s["max_lag_milliseconds"] / 1000.0) * float(sr))))
    return win, hop, frames, max_lag


def gaussian_exceedance(k: float) -> float:
    """diffusion.py:126-136: P(|x| > k sigma) for a Gaussian."""
    import math
    phi = 0.5 * (1.0 + math.erf(float(k) / np.sqrt(2.0)))
    return 2.0 * (1.0 - phi)


def window_max_abs_autocorr(w: np.ndarray, max_lag: int) -> float:
    """diffusion.py:139-159.  float32 arithmetic throughout (float32 mean, float32 dots), like the reference."""
    if w.size < 4:
        return float("nan")
    w0 = w - float(np.mean(w))
    den = float(np.dot(w0, w0))
    if den <= 1e-20:
        return float("nan")
    best = 0.0
    for lag in range(1, min(max_lag, w0.size - 2) + 1):
        best = max(best, abs(float(np.dot(w0[:-lag], w0[lag:]) / den)))
    return best


def window_echo_density(w: np.ndarray, thr_rms: float, normalise: bool) -> float:
    """diffusion.py:205-226."""
    if w.size < 4:
        return float("nan")
    w0 = w - float(np.mean(w))
    rms = float(np.sqrt(np.mean(w0 * w0)))
    if rms <= 1e-20:
        return float("nan")
    frac = float(np.mean(np.abs(w0) > float(thr_rms) * rms))
    if not normalise:
        return frac
    ex = gaussian_exceedance(thr_rms)
    return float("nan") if ex <= 1e-12 else frac / ex


def window_corr0(a: np.ndarray, b: np.ndarray) -> float:
    """diffusion.py:162-174."""
    if a.size != b.size or a.size < 4:
        return float("nan")
    a0 = a - float(np.mean(a)); b0 = b - float(np.mean(b))
    aa = float(np.dot(a0, a0)); bb = float(np.dot(b0, b0))
    if aa <= 1e-20 or bb <= 1e-20:
        return float("nan")
    return float(np.dot(a0, b0) / np.sqrt(aa * bb))


def window_iacc_max(a: np.ndarray, b: np.ndarray, max_lag: int) -> float:
    """diffusion.py:177-202: max |normalised cross-correlation| over lags -max_lag..+max_lag."""
    if a.size != b.size or a.size < 4:
        return float("nan")
    a0 = a - float(np.mean(a)); b0 = b - float(np.mean(b))
    den = np.sqrt(float(np.dot(a0, a0)) * float(np.dot(b0, b0)))
    if den <= 1e-20:
        return float("nan")
    L = min(max_lag, a0.size - 2)
    best = abs(float(np.dot(a0, b0) / den))
    for lag in range(1, L + 1):
        best = max(best, abs(float(np.dot(a0[:-lag], b0[lag:]) / den)))
        best = max(best, abs(float(np.dot(a0[lag:], b0[:-lag]) / den)))
    return best


def analyse_diffusion(x: np.ndarray, sr: int = SR, **kw) -> Dict:
    """diffusion.py:234-291: per-window max|autocorr| and echo density of one channel (float32 series)."""
    s = dict(DIFF_DEFAULTS); s.update(kw)
    v, start = diffusion_trim(x, sr, bool(s["trim_to_peak"]), float(s["ignore_leading_seconds"]))
    win, hop, frames, max_lag = diffusion_geometry(v.size, sr, s)
    if frames <= 0:
        raise ValueError("Not enough samples for diffusion analysis windows.")
    t = np.zeros(frames, dtype=np.float32); ac = np.zeros(frames, dtype=np.float32); ed = np.zeros(frames, dtype=np.float32)
    for i in range(frames):
        w = v[i * hop : i * hop + win]
        t[i] = (i * hop + win * 0.5) / float(sr)
        ac[i] = window_max_abs_autocorr(w, max_lag)
        ed[i] = window_echo_density(w, s["echo_density_threshold_rms"], bool(s["echo_density_normalise_to_gaussian"]))
    return dict(start=start, time=t, ac=ac, ed=ed, win=win, hop=hop, max_lag=max_lag)


def diffusion_stereo(left: np.ndarray, right: np.ndarray, sr: int = SR, **kw) -> Dict:
    """diffusion.py:323-376: both channels trimmed at the peak of their MEAN; corr0 and IACC per window."""
    s = dict(DIFF_DEFAULTS); s.update(kw)
    comb = (left.astype(np.float64) + right.astype(np.float64)) * 0.5
    ct, start = diffusion_trim(comb.astype(np.float32), sr, bool(s["trim_to_peak"]), float(s["ignore_leading_seconds"]))
    l = left.astype(np.float32)[start : start + ct.size]
    r = right.astype(np.float32)[start : start + ct.size]
    win, hop, frames, max_lag = diffusion_geometry(ct.size, sr, s)
    c0 = np.zeros(max(frames, 0), dtype=np.float32); ia = np.zeros(max(frames, 0), dtype=np.float32)
    for i in range(frames):
        wl, wr = l[i * hop : i * hop + win], r[i * hop : i * hop + win]
        c0[i] = window_corr0(wl, wr)
        ia[i] = window_iacc_max(wl, wr, max_lag)
    return dict(start=start, corr0=c0, iacc=ia)


def diffusion_summary(names: List[str], series: List[Dict], stereo: Optional[Dict] = None) -> str:
    """diffusion.py:457-476."""
    lines = []
    for name, d in zip(names, series):
        lines.append(f"[{name}]")
        lines.append(f"  median_max_abs_autocorr={float(np.nanmedian(d['ac'])):.3f}")
        lines.append(f"  median_echo_density={float(np.nanmedian(d['ed'])):.3f}")
        if stereo is not None:
            lines.append(f"  median_corr0={float(np.nanmedian(stereo['corr0'])):.3f}")
            lines.append(f"  median_iacc_max={float(np.nanmedian(stereo['iacc'])):.3f}")
    return "\n".join(lines)


# ----------------------------------------------------------------------------------------
# section 8f rank 4: sweep deconvolution                  (reference analyse/deconvolve.py:85-193)
# ----------------------------------------------------------------------------------------


def next_power_of_two(n: int) -> int:
    """deconvolve.py:85-88."""
    return 1 if n <= 1 else 1 << (int(n - 1).bit_length())


def sweep_downmix(samples_nc: np.ndarray) -> np.ndarray:
    """deconvolve.py:91-97: float64 mean over channels -> float32."""
    return np.mean(samples_nc.astype(np.float64, copy=False), axis=1).astype(np.float32)


def deconvolve(recorded: np.ndarray, sweep: np.ndarray, regularization_relative: float = 1e-10,
               normalise_peak: bool = True, target_peak: float = 0.95, remove_dc: bool = True,
               output_length_mode: str = "recorded") -> np.ndarray:
    """deconvolve.py:124-193: H = Y conj(X) / (|X|^2 + eps), eps = reg * max(1e-30, max |X|^2); per channel irfft ->
    float32, truncate, remove the float32 mean, then one peak normalisation over all channels.  (N_out, C) float32."""
    rec = pcm_to_float32(np.asarray(recorded))
    rec = rec.reshape(-1, 1) if rec.ndim == 1 else rec
    sw = np.asarray(sweep, dtype=np.float32)
    if rec.shape[0] < 8 or sw.size < 8:
        raise ValueError("Recorded and sweep must both contain at least a few samples.")
    n_rec = int(rec.shape[0])
    n_fft = next_power_of_two(max(n_rec, int(sw.size)))
    X = np.fft.rfft(sw.astype(np.float64), n=n_fft)
    power = np.abs(X) ** 2
    eps = float(regularization_relative) * max(1e-30, float(np.max(power)))
    denom = power + eps
    Xc = np.conj(X)
    chans = []
    for c in range(rec.shape[1]):
        Y = np.fft.rfft(rec[:, c].astype(np.float64), n=n_fft)
        h = np.fft.irfft((Y * Xc) / denom, n=n_fft).astype(np.float32)
        if output_length_mode == "recorded":
            h = h[:n_rec]
        elif output_length_mode != "full_fft":
            raise ValueError(f"Unknown output_length_mode: {output_length_mode}")
        if remove_dc and h.size > 0:
            h = (h - float(np.mean(h))).astype(np.float32)
        chans.append(h)
    ir = np.stack(chans, axis=1).astype(np.float32)
    if normalise_peak:
        peak = float(np.max(np.abs(ir))) if ir.size else 0.0
        if peak > 0.0:
            ir = (ir * (float(target_peak) / peak)).astype(np.float32)
    return ir
