"""
Device engine: owns torch-ROCm tensors (device memory only), twiddle/window tables and the stream, and
calls the hand-written HIP kernels in libira.so through ctypes.  PyTorch is plumbing here -- no torch op
computes anything on the data path.

One Engine per process (= per GPU).  All entry points are batched over channels ("segments"); the
single-channel drop-in functions in audio_analysis_amd.analyse.* call them with a batch of one.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import IraError, check

FIT_DOUBLES = 8
EDC_SCRATCH_DOUBLES = 2048
EDC_TILE = 4096


def _ptr(t) -> int:
    return 0 if t is None else int(t.data_ptr())


@dataclass
class ChannelBatch:
    """A ragged batch of mono channels resident in HBM: flat float32 samples + per-channel offset/length."""
    x: "object"                  # torch.float32 (total,) on device
    off: np.ndarray              # int64 (B,) host
    length: np.ndarray           # int64 (B,) host
    off_dev: "object"
    len_dev: "object"
    peak: Optional[np.ndarray] = None       # int64 (B,) host, filled by Engine.peaks()
    peak_abs: Optional[np.ndarray] = None   # float32 (B,)

    @property
    def count(self) -> int:
        return int(self.off.size)


_ENGINE = None


def get_engine() -> "Engine":
    """Process-wide engine on the current CUDA/HIP device.  Raises if there is no GPU or no libira.so."""
    global _ENGINE
    if _ENGINE is None:
        _ENGINE = Engine()
    return _ENGINE


class Engine:
    def __init__(self, device: Optional[str] = None):
        import torch

        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise IraError("audio_analysis_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback for the product path.")
        self.torch = torch
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._tables: Dict[Tuple, object] = {}

    # ------------------------------------------------------------------ plumbing
    @property
    def stream(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def sync(self) -> None:
        self.torch.cuda.current_stream(self.device).synchronize()

    def to_dev(self, a: np.ndarray):
        return self.torch.from_numpy(np.ascontiguousarray(a)).to(self.device, non_blocking=False)

    def empty(self, n: int, dtype):
        return self.torch.empty(int(max(n, 1)), dtype=dtype, device=self.device)

    def upload(self, channels: Sequence[np.ndarray]) -> ChannelBatch:
        """Host float32 channels -> one flat device buffer (H2D)."""
        lens = np.array([int(c.size) for c in channels], dtype=np.int64)
        off = np.zeros(len(channels), dtype=np.int64)
        if len(channels) > 1:
            off[1:] = np.cumsum(lens[:-1])
        flat = np.empty(int(lens.sum()), dtype=np.float32)
        for c, o, n in zip(channels, off, lens):
            if c.ndim != 1:
                raise ValueError("expects 1D mono arrays")
            flat[o : o + n] = c.astype(np.float32, copy=False)
        return self.wrap(self.to_dev(flat), off, lens)

    def wrap(self, x_dev, off: np.ndarray, lens: np.ndarray) -> ChannelBatch:
        off = np.ascontiguousarray(off, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        return ChannelBatch(x=x_dev, off=off, length=lens, off_dev=self.to_dev(off), len_dev=self.to_dev(lens))

    # ------------------------------------------------------------------ tables (host NumPy -> device, cached)
    def window(self, n: int, use_hann: bool, precision: int):
        key = ("win", n, bool(use_hann), precision)
        if key not in self._tables:
            w = np.hanning(n).astype(np.float64) if use_hann else np.ones(n, dtype=np.float64)
            self._tables[key] = self.to_dev(w.astype(np.float32) if precision == 32 else w)
        return self._tables[key]

    def twiddle(self, n: int, precision: int):
        """exp(-2 pi i k / n), k < n/2, interleaved (re, im)."""
        key = ("tw", n, precision)
        if key not in self._tables:
            k = np.arange(n // 2, dtype=np.float64)
            ang = -2.0 * np.pi * k / float(n)
            t = np.stack([np.cos(ang), np.sin(ang)], axis=1)
            self._tables[key] = self.to_dev(t.astype(np.float32) if precision == 32 else t)
        return self._tables[key]

    # ------------------------------------------------------------------ a2
    def peaks(self, b: ChannelBatch) -> np.ndarray:
        """argmax|x| per channel (first max wins), synchronises once and caches on the batch."""
        if b.peak is None:
            t = self.torch
            pk = self.empty(b.count, t.int64)
            pa = self.empty(b.count, t.float32)
            check(self.lib.ira_peak_index(_ptr(b.x), _ptr(b.off_dev), _ptr(b.len_dev), b.count, _ptr(pk), _ptr(pa),
                                          self.stream), "ira_peak_index")
            b.peak = pk.cpu().numpy()[: b.count].copy()
            b.peak_abs = pa.cpu().numpy()[: b.count].copy()
        return b.peak

    # ------------------------------------------------------------------ a3
    def edc_db(self, x_dev, seg_off: np.ndarray, seg_len: np.ndarray, eps: float, floor_db: float,
               want_f64: bool = False):
        """Schroeder EDC in dB for segments of x_dev.  Returns (edc flat f32 device, edc_off host int64[, f64])."""
        t = self.torch
        n = int(seg_off.size)
        if np.any(seg_len > EDC_SCRATCH_DOUBLES * EDC_TILE):
            raise ValueError("segment too long for the EDC kernel (> 8.3 M samples)")
        edc_off = np.zeros(n, dtype=np.int64)
        if n > 1:
            edc_off[1:] = np.cumsum(seg_len[:-1])
        out = self.empty(int(seg_len.sum()), t.float32)
        out64 = self.empty(int(seg_len.sum()), t.float64) if want_f64 else None
        scratch = self.empty(n * EDC_SCRATCH_DOUBLES, t.float64)
        # NOTE: device temporaries must stay referenced until the call is enqueued (the caching allocator
        # would otherwise hand the same block to the next to_dev()).
        d_off, d_len, d_eoff = self.to_dev(seg_off), self.to_dev(seg_len), self.to_dev(edc_off)
        check(self.lib.ira_edc_db(_ptr(x_dev), _ptr(d_off), _ptr(d_len), n, float(eps), float(floor_db), _ptr(out),
                                  _ptr(out64), _ptr(d_eoff), _ptr(scratch), self.stream), "ira_edc_db")
        if want_f64:
            return out, edc_off, out64
        return out, edc_off

    # ------------------------------------------------------------------ a4/a5/a16
    def curve_fits(self, y_dev, off: np.ndarray, lens: np.ndarray, t_mul: float, t_div: float,
                   ranges: Sequence[Tuple[float, float]], min_points: int, cross: Sequence[float] = (),
                   rel_to_peak: bool = False, floor_db: float = -120.0, min_peak_above_floor: float = 0.0,
                   t_axis_dev=None):
        """Returns (fits (ncurves, nranges, 8) float64 device, cross (ncurves, ncross) float64 device)."""
        t = self.torch
        n = int(off.size)
        nr, nc = len(ranges), len(cross)
        fit = self.empty(n * max(nr, 1) * FIT_DOUBLES, t.float64)
        cr = self.empty(n * max(nc, 1), t.float64)
        flat = [v for r in ranges for v in r]
        d_off, d_len = self.to_dev(off), self.to_dev(lens)
        check(self.lib.ira_curve_fits(_ptr(y_dev), _ptr(d_off), _ptr(d_len), n,
                                      int(lens.max()) if n else 0, float(t_mul), float(t_div), _ptr(t_axis_dev),
                                      _lib.dbl_array(flat), nr,
                                      int(min_points), _lib.dbl_array(list(cross)), nc, 1 if rel_to_peak else 0,
                                      float(floor_db), float(min_peak_above_floor), _ptr(fit), _ptr(cr), self.stream),
              "ira_curve_fits")
        return fit[: n * nr * FIT_DOUBLES].view(n, nr, FIT_DOUBLES) if nr else None, \
            (cr[: n * nc].view(n, nc) if nc else None)

    # ------------------------------------------------------------------ a11
    def stft_mag_db(self, x_dev, seg_off: np.ndarray, nframes: np.ndarray, n_fft: int, hop: int, use_hann: bool,
                    floor_db: float, precision: int = 32, frame_sel: Optional[List[np.ndarray]] = None):
        """
        STFT magnitude (dB) of segments starting at seg_off with nframes[s] valid frames each.
        Returns (out flat f32 device, out_off host int64); out[s] is a C-contiguous (n_fft/2+1, T_s) matrix.
        frame_sel: optional per-segment arrays of frame indices (then T_s = len(frame_sel[s])).
        """
        t = self.torch
        n = int(seg_off.size)
        f = n_fft // 2 + 1
        if frame_sel is not None:
            cols = np.array([int(s.size) for s in frame_sel], dtype=np.int32)
            sel_off = np.zeros(n, dtype=np.int64)
            if n > 1:
                sel_off[1:] = np.cumsum(cols[:-1])
            sel = self.to_dev(np.concatenate(frame_sel).astype(np.int32)) if cols.sum() else self.empty(1, t.int32)
            sel_off_dev = self.to_dev(sel_off)
        else:
            cols = np.ascontiguousarray(nframes, dtype=np.int32)
            sel = None
            sel_off_dev = None
        out_off = np.zeros(n, dtype=np.int64)
        sizes = cols.astype(np.int64) * f
        if n > 1:
            out_off[1:] = np.cumsum(sizes[:-1])
        out = self.empty(int(sizes.sum()), t.float32)
        d_off, d_cols, d_ooff = self.to_dev(seg_off), self.to_dev(cols), self.to_dev(out_off)
        check(self.lib.ira_stft_mag_db(_ptr(x_dev), _ptr(d_off), _ptr(d_cols), n,
                                       int(cols.max()) if n else 0, int(n_fft), int(hop),
                                       _ptr(self.window(n_fft, use_hann, precision)),
                                       _ptr(self.twiddle(n_fft, precision)), int(precision), float(floor_db),
                                       _ptr(out), _ptr(d_ooff), _ptr(sel), _ptr(sel_off_dev),
                                       self.stream), "ira_stft_mag_db")
        return out, out_off, cols
