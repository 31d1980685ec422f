"""
Device engine: owns torch-ROCm tensors (device memory only), twiddle/window tables and the stream, and
calls the hand-written HIP kernels in libira.so through ctypes.  PyTorch is plumbing here -- no torch op
computes anything on the data path.

One Engine per process (= per GPU).  All entry points are batched over channels ("segments"); the
single-channel drop-in functions in audio_analysis_amd.analyse.* call them with a batch of one.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import IraError, check

FIT_DOUBLES = 8
EDC_SCRATCH_DOUBLES = 4096
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
    ready: Optional[object] = None          # event recorded after the batch's arrays were enqueued for upload
    _peak_pending: Optional[tuple] = None   # Engine.peaks_begin(): pinned results + event, picked up by Engine.peaks()

    @property
    def count(self) -> int:
        return int(self.off.size)


class _TimedLib:
    """
    Transparent proxy over the ctypes library: when the engine's `events` list is set, every ira_* launch is
    bracketed by two HIP events recorded on the SAME stream the kernels are enqueued on, so per-call device
    time can be read back after a synchronise (bench.py's roofline uses this).
    """
    _PLAIN = {"ira_error_string", "ira_abi_version", "ira_ar_partial_doubles", "ira_ar_exact_doubles"}

    def __init__(self, lib, eng):
        self._lib, self._eng, self._cache = lib, eng, {}

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name in self._PLAIN or not name.startswith("ira_"):
            return fn
        if name not in self._cache:
            eng = self._eng

            def timed(*args, _fn=fn, _name=name):
                rec = eng.events
                if rec is None:
                    return _fn(*args)
                t = eng.torch
                stream = t.cuda.current_stream(eng.device)
                e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
                e0.record(stream)
                rc = _fn(*args)
                e1.record(stream)
                rec.append((_name + eng.event_tag, e0, e1))
                return rc

            self._cache[name] = timed
        return self._cache[name]


class HostFuture:
    """Result of Engine.fetch(): a device tensor on its way into pinned host memory.  get() is valid once the
    stream it was enqueued on has reached the copy (Engine.sync(), or the event of the step it belongs to).
    The first get() (or the future's destruction) hands its slice of the fetch arena back to the engine."""

    def __init__(self, pinned, shape, owner=None, segment=None):
        self._pinned, self._shape, self._owner, self._segment = pinned, shape, owner, segment

    def _release(self):
        if self._owner is not None:
            self._owner._fetch_release(self._segment)
            self._owner = None

    def get(self) -> np.ndarray:
        out = self._pinned.numpy().reshape(self._shape).copy()
        self._release()
        return out

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass


def _TORCH_DTYPES(t):
    return {"<i8": t.int64, "<i4": t.int32, "<f8": t.float64, "<f4": t.float32, "|u1": t.uint8, "<i2": t.int16}


_ENGINE = None


def get_engine() -> "Engine":
    """Process-wide engine on the current CUDA/HIP device.  Raises if there is no GPU or no libira.so."""
    global _ENGINE
    if _ENGINE is None:
        _ENGINE = Engine()
    return _ENGINE


def conv_size(need: int, three_pow2: bool = True) -> int:
    """Smallest Bluestein convolution size the library takes (ira_fft_split: 2^k, or 3 * 2^k when allowed) that holds
    `need` distinct lags; at least 16."""
    need = max(int(need), 16)
    m = 1 << int(need - 1).bit_length()
    if three_pow2 and m >= 128 and 3 * (m >> 2) >= need:
        m = 3 * (m >> 2)
    return m


class Engine:
    def __init__(self, device: Optional[str] = None):
        import torch

        self.lib = _TimedLib(_lib.load(), self)
        self.events = None          # list of (name, start_event, end_event) while timing is on
        self.event_tag = ""
        if not torch.cuda.is_available():
            raise IraError("audio_analysis_amd needs an AMD GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback for the product path.")
        self.torch = torch
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self._tables: Dict[Tuple, object] = {}
        self._filter_pools: Dict[tuple, dict] = {}      # (M, stream) -> pool, see _filter_pool
        self._filter_tick = 0
        self._ring = None
        # round 5: the band inverses can leave the energies of their signals' EDC tiles for ira_edc_fits, which then reads a
        # band signal once less.  Measured (profiles/r05_tile_energies.txt, config 3): ira_edc_fits 6.66 -> 5.40 ms per step,
        # but the second pass pays 1.6 ms for forming the partials (a serial tail of every tile): 28.2 -> 28.6 ms of device
        # time.  OFF by default; IRA_BAND_TILE_ENERGIES=1 / this attribute is the A/B switch.
        self.band_tile_energies = os.environ.get("IRA_BAND_TILE_ENERGIES", "0") == "1"
        if os.environ.get("IRA_WORKSPACE_MB"):            # tuning knob: long-FFT jobs per launch (see workspace_budget_bytes)
            self.workspace_budget_bytes = int(float(os.environ["IRA_WORKSPACE_MB"]) * (1 << 20))

    # ------------------------------------------------------------------ plumbing
    @property
    def stream(self) -> int:
        return int(self.torch.cuda.current_stream(self.device).cuda_stream)

    def collect_events(self):
        """Synchronise and return {call name: [milliseconds, ...]} for the launches recorded so far."""
        self.sync()
        out = {}
        for name, e0, e1 in (self.events or []):
            out.setdefault(name, []).append(float(e0.elapsed_time(e1)))
        if self.events is not None:
            self.events = []
        return out

    def sync(self) -> None:
        self.torch.cuda.current_stream(self.device).synchronize()

    # Small host arrays (offsets, lengths, job tables) go through a pinned staging ring and an ASYNC copy: a
    # pageable .to(device) blocks the host until the stream has drained, which would serialise host and GPU.
    _RING_BYTES = 32 << 20
    _ASYNC_LIMIT = 1 << 20

    def to_dev(self, a: np.ndarray):
        t = self.torch
        a = np.ascontiguousarray(a)
        nbytes = a.nbytes
        if nbytes == 0 or nbytes > self._ASYNC_LIMIT:
            return t.from_numpy(a).to(self.device, non_blocking=False)
        if self._ring is None:
            self._ring = t.empty(self._RING_BYTES, dtype=t.uint8).pin_memory()
            self._ring_np = self._ring.numpy()
            self._ring_pos = 0
        pos = (self._ring_pos + 63) & ~63
        if pos + nbytes > self._RING_BYTES:
            t.cuda.synchronize(self.device)  # every copy issued so far (on any stream) has left the ring before it wraps
            pos = 0
        self._ring_np[pos : pos + nbytes] = a.view(np.uint8).reshape(-1)
        self._ring_pos = pos + nbytes
        src = self._ring[pos : pos + nbytes].view(_TORCH_DTYPES(t)[a.dtype.str]).view(a.shape)
        # (round 5: the same copy as a bare hipMemcpyAsync (ctypes) into a device ring that mirrors this one -- no tensor
        # allocation, no dispatcher -- was built and measured: the same ~85 us of host time per call, which is the runtime's
        # own cost of a small asynchronous copy; 11.9-12.1 vs 11.9-12.3 ms per bundle step.  Removed.)
        return src.to(self.device, non_blocking=True)

    def job_tables(self, *arrays):
        """to_dev_pack under the name that says what the arrays are: the job tables of ONE call (offsets, lengths, indices ...
        read only by the launches the calling method enqueues before it returns)."""
        return self.to_dev_pack(*arrays)

    def to_dev_pack(self, *arrays):
        """Several small host arrays -> ONE write into the pinned staging ring and ONE asynchronous H2D copy; returns the
        device views in order (None entries stay None).  A call's job tables (offsets, lengths, indices ...) used to be
        one copy each: 65 copy-engine operations of ~5 us per report step (profiles/r02_*: __amd_rocclr_copyBuffer), each
        serialised with the kernels of its stream."""
        t = self.torch
        items = [(i, np.ascontiguousarray(a)) for i, a in enumerate(arrays) if a is not None]
        total = sum(((a.nbytes + 15) & ~15) for _, a in items)
        if not items or total == 0 or total > self._ASYNC_LIMIT or any(a.nbytes == 0 for _, a in items):
            return [None if a is None else self.to_dev(a) for a in arrays]
        blob = np.zeros(total, dtype=np.uint8)
        spans, pos = [], 0
        for i, a in items:
            blob[pos : pos + a.nbytes] = a.view(np.uint8).reshape(-1)
            spans.append((i, pos, a))
            pos += (a.nbytes + 15) & ~15
        dev = self.to_dev(blob)
        out = [None] * len(arrays)
        for i, pos, a in spans:
            out[i] = dev[pos : pos + a.nbytes].view(_TORCH_DTYPES(t)[a.dtype.str]).view(a.shape)
        return out

    def empty(self, n: int, dtype):
        return self.torch.empty(int(max(n, 1)), dtype=dtype, device=self.device)

    # Small results (fit records, statistics, roots) leave through a pinned arena with ASYNC copies, so that a step's
    # device->host traffic is enqueued with its kernels and read after ONE wait -- and so that the next step can be
    # enqueued behind it (audio_analysis_amd.pipeline.FullReport.submit / finish).  The arena is cut into segments that
    # are filled front to back; a segment is reused only when every future carved from it has been read (or dropped):
    # an unread result is never overwritten, however many steps are in flight or however large their records are.
    # When no segment is free the copy gets a private pinned allocation (slower, never wrong).
    _FETCH_BYTES = 96 << 20
    _FETCH_SEGMENTS = 6

    def _fetch_release(self, segment: int) -> None:
        self._fetch_out[segment] -= 1

    def fetch(self, tensor) -> HostFuture:
        t = self.torch
        tensor = tensor.contiguous()
        nbytes = int(tensor.numel()) * tensor.element_size()
        seg_bytes = self._FETCH_BYTES // self._FETCH_SEGMENTS
        if getattr(self, "_fetch_arena", None) is None:
            self._fetch_arena = t.empty(self._FETCH_BYTES, dtype=t.uint8).pin_memory()
            self._fetch_seg, self._fetch_pos = 0, 0
            self._fetch_out = [0] * self._FETCH_SEGMENTS
        pos = (self._fetch_pos + 63) & ~63
        seg = self._fetch_seg
        if nbytes <= seg_bytes and pos + nbytes > seg_bytes:
            # current segment is full: move on to the next one if everything carved from it has been consumed
            nxt = (seg + 1) % self._FETCH_SEGMENTS
            if self._fetch_out[nxt] == 0:
                seg, pos = nxt, 0
                self._fetch_seg = nxt
        if nbytes > seg_bytes or pos + nbytes > seg_bytes:
            host = t.empty(tensor.shape, dtype=tensor.dtype).pin_memory()
            host.copy_(tensor, non_blocking=True)
            return HostFuture(host, tuple(tensor.shape))
        if pos == 0 and self._fetch_out[seg] != 0:           # only reachable on the very first wrap with stale futures
            host = t.empty(tensor.shape, dtype=tensor.dtype).pin_memory()
            host.copy_(tensor, non_blocking=True)
            return HostFuture(host, tuple(tensor.shape))
        self._fetch_pos = pos + nbytes
        self._fetch_out[seg] += 1
        base = seg * seg_bytes + pos
        host = self._fetch_arena[base : base + nbytes].view(tensor.dtype).view(tensor.shape)
        host.copy_(tensor, non_blocking=True)
        return HostFuture(host, tuple(tensor.shape), self, seg)

    _side = None

    def side_stream(self):
        """High-priority stream for the tiny peak-pick launch whose result the HOST needs before it can lay out a
        step: it overtakes whatever the main stream still has queued from the previous step."""
        if self._side is None:
            self._side = self.torch.cuda.Stream(device=self.device, priority=-1)
        return self._side

    _lanes = None
    num_lanes = max(1, min(4, int(os.environ.get("IRA_STREAMS", "2"))))
    # which report blocks go to which lane (pipeline.FullReport.submit; None = its measured default).  IRA_LANE_DEAL, e.g.
    # "0,1|4,3,5,2", is the A/B switch of tools: 0 bands, 1 spectrum, 2 zplane, 3 decay, 4 modal, 5 stft
    lane_deal = ([[int(v) for v in part.split(",")] for part in os.environ["IRA_LANE_DEAL"].split("|")]
                 if os.environ.get("IRA_LANE_DEAL") else None)

    def block_streams(self):
        """
        Compute streams ("lanes") for independent report blocks (pipeline.FullReport).  The many small-grid,
        latency-bound kernels of one lane (curve fits, Cholesky solves, root finders, unwrap scans) run beside another
        lane's wide kernels instead of leaving most CUs idle, and the launch gaps of one lane are covered by the others.
        Every lane has its own chirp-filter plan pool (_filters); tables are uploaded synchronously (_table_to_dev).
        IRA_STREAMS=1 keeps everything on the caller's stream (A/B switch); default 2 (round 3, 256 x 10 s per step: 1 lane
        12.4 k, 2 lanes 13.4 k, 3 lanes 11.6-12.3 k IRs/s; round 1 measured 3 lanes best at 64 per step, when the small
        latency-bound kernels were a fifth of the step).
        """
        if self.num_lanes <= 1:
            return None
        if self._lanes is None or len(self._lanes) != self.num_lanes:
            # IRA_LANE_PRIO (A/B only), e.g. "-1,0": HIP stream priorities of the lanes (lower = served first)
            prio = [int(v) for v in os.environ.get("IRA_LANE_PRIO", "").split(",") if v.strip()]
            self._lanes = tuple(self.torch.cuda.Stream(device=self.device, priority=(prio[i] if i < len(prio) else 0))
                                for i in range(self.num_lanes))
        return self._lanes

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

    def subset(self, b: ChannelBatch, idx: np.ndarray) -> ChannelBatch:
        """The channels idx of a batch as a batch of their own: same sample buffer, their offsets / lengths / peaks.
        (Its ready event is recorded behind the new offset/length uploads; the samples themselves are already resident:
        whoever knows the peaks has waited for the batch's upload.)"""
        idx = np.asarray(idx, dtype=np.int64)
        sub = self.wrap(b.x, b.off[idx], b.length[idx])
        if b.peak is not None:
            sub.peak = b.peak[idx].copy()
            sub.peak_abs = b.peak_abs[idx].copy() if b.peak_abs is not None else None
        return sub

    def wrap(self, x_dev, off: np.ndarray, lens: np.ndarray) -> ChannelBatch:
        off = np.ascontiguousarray(off, dtype=np.int64)
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        d_off, d_len = self.to_dev_pack(off, lens)
        b = ChannelBatch(x=x_dev, off=off, length=lens, off_dev=d_off, len_dev=d_len)
        b.ready = self.torch.cuda.Event()
        b.ready.record(self.torch.cuda.current_stream(self.device))
        return b

    # ------------------------------------------------------------------ tables (host NumPy -> device, cached)
    def window(self, n: int, use_hann: bool, precision: int):
        key = ("win", n, bool(use_hann), precision)
        if key not in self._tables:
            w = np.hanning(n).astype(np.float64) if use_hann else np.ones(n, dtype=np.float64)
            self._tables[key] = self._table_to_dev(w.astype(np.float32) if precision == 32 else w)
        return self._tables[key]

    def _table_to_dev(self, a: np.ndarray):
        """Plan data (windows, twiddles): uploaded once and COMPLETELY before use -- tables are shared by every stream."""
        tab = self.to_dev(a)
        self.sync()
        return tab

    def twiddle(self, n: int, precision: int):
        """exp(-2 pi i k / n), k < n/2, interleaved (re, im)."""
        key = ("tw", n, precision)
        if key not in self._tables:
            k = np.arange(n // 2, dtype=np.float64)
            ang = -2.0 * np.pi * k / float(n)
            t = np.stack([np.cos(ang), np.sin(ang)], axis=1)
            self._tables[key] = self._table_to_dev(t.astype(np.float32) if precision == 32 else t)
        return self._tables[key]

    # ------------------------------------------------------------------ a2
    def peaks_begin(self, b: ChannelBatch) -> None:
        """Enqueue the peak pick of a batch on the high-priority side stream (behind the batch's upload only) and its
        result's copy into pinned memory, WITHOUT waiting: the host can do other work while a freshly uploaded batch is
        still on its way.  peaks() picks the result up."""
        if b.peak is not None or getattr(b, "_peak_pending", None) is not None:
            return
        t = self.torch
        side = self.side_stream()
        if b.ready is not None:
            side.wait_event(b.ready)
        with t.cuda.stream(side):
            pk = self.empty(b.count, t.int64)
            pa = self.empty(b.count, t.float32)
            check(self.lib.ira_peak_index(_ptr(b.x), _ptr(b.off_dev), _ptr(b.len_dev), b.count,
                                          int(b.length.max()) if b.count else 0, _ptr(pk), _ptr(pa), self.stream),
                  "ira_peak_index")
            hk = t.empty(pk.shape, dtype=pk.dtype, pin_memory=True)
            ha = t.empty(pa.shape, dtype=pa.dtype, pin_memory=True)
            hk.copy_(pk, non_blocking=True)
            ha.copy_(pa, non_blocking=True)
            ev = t.cuda.Event()
            ev.record(side)
        b._peak_pending = (hk, ha, ev, pk, pa)

    def peaks(self, b: ChannelBatch) -> np.ndarray:
        """argmax|x| per channel (first max wins), synchronises once and caches on the batch."""
        pend = getattr(b, "_peak_pending", None)
        if b.peak is None and pend is not None:
            hk, ha, ev, _, _ = pend
            ev.synchronize()
            b.peak = hk.numpy()[: b.count].copy()
            b.peak_abs = ha.numpy()[: b.count].copy()
            b._peak_pending = None
        if b.peak is None:
            t = self.torch
            pk = self.empty(b.count, t.int64)
            pa = self.empty(b.count, t.float32)
            check(self.lib.ira_peak_index(_ptr(b.x), _ptr(b.off_dev), _ptr(b.len_dev), b.count,
                                          int(b.length.max()) if b.count else 0, _ptr(pk), _ptr(pa), self.stream),
                  "ira_peak_index")
            b.peak = pk.cpu().numpy()[: b.count].copy()
            b.peak_abs = pa.cpu().numpy()[: b.count].copy()
        return b.peak

    # ------------------------------------------------------------------ a3
    def edc_db(self, x_dev, seg_off: np.ndarray, seg_len: np.ndarray, eps: float, floor_db: float,
               want_f64: bool = False):
        """Schroeder EDC in dB for segments of x_dev.  Returns (edc flat f32 device, edc_off host int64[, f64])."""
        t = self.torch
        n = int(seg_off.size)
        if np.any(seg_len > 2047 * EDC_TILE):
            raise ValueError("segment too long for the EDC kernel (> 8.3 M samples)")
        edc_off = np.zeros(n, dtype=np.int64)
        if n > 1:
            edc_off[1:] = np.cumsum(seg_len[:-1])
        out = self.empty(int(seg_len.sum()), t.float32)
        out64 = self.empty(int(seg_len.sum()), t.float64) if want_f64 else None
        scratch = self.empty(n * EDC_SCRATCH_DOUBLES, t.float64)
        # NOTE: device temporaries must stay referenced until the call is enqueued (the caching allocator
        # would otherwise hand the same block to the next to_dev()).
        d_off, d_len, d_eoff = self.job_tables(seg_off, seg_len, edc_off)
        check(self.lib.ira_edc_db(_ptr(x_dev), _ptr(d_off), _ptr(d_len), n, int(seg_len.max()), float(eps),
                                  float(floor_db), _ptr(out),
                                  _ptr(out64), _ptr(d_eoff), _ptr(scratch), self.stream), "ira_edc_db")
        if want_f64:
            return out, edc_off, out64
        return out, edc_off

    def edc_box_smooth(self, edc64_dev, off: np.ndarray, lens: np.ndarray, window: int, floor_db: float):
        """Box smoothing of unfloored float64 dB curves + floor + float32 cast (ira_edc_box_smooth)."""
        t = self.torch
        lens = np.ascontiguousarray(lens, dtype=np.int64)
        out = self.empty(int(lens.sum()), t.float32)
        d_off, d_len = self.job_tables(np.ascontiguousarray(off, np.int64), lens)
        check(self.lib.ira_edc_box_smooth(_ptr(edc64_dev), _ptr(d_off), _ptr(d_len), int(lens.size), int(lens.max()),
                                          int(window), float(floor_db), _ptr(out), self.stream), "ira_edc_box_smooth")
        return out

    # ------------------------------------------------------------------ a3-a6 fused
    def edc_fits(self, x_dev, seg_off: np.ndarray, seg_len: np.ndarray, eps: float, floor_db: float,
                 t_mul: float, t_div: float, ranges: Sequence[Tuple[float, float]], min_points: int,
                 cross: Sequence[float] = (), want_edc: bool = False, tiles=None):
        """Schroeder EDC crossings + decay-line fits straight from the samples (ira_edc_fits).
        Returns (fits (n, nranges, 8) f64 device | None, cross (n, ncross) f64 device | None, edc f32 device | None,
        edc_off host int64): the EDC curve is only written when want_edc.
        tiles = (part device, part_off int64, part_wgs int32, part_tiles int32 per segment): partial tile energies the producer of
        the segments left behind (band_irfft(want_tiles=True)); every such segment must END where its signal ends."""
        t = self.torch
        n = int(seg_off.size)
        seg_len = np.ascontiguousarray(seg_len, dtype=np.int64)
        if np.any(seg_len > 2047 * EDC_TILE):
            raise ValueError("segment too long for the EDC kernel (> 8.3 M samples)")
        nr, nc = len(ranges), len(cross)
        edc_off = np.zeros(n, dtype=np.int64)
        if n > 1:
            edc_off[1:] = np.cumsum(seg_len[:-1])
        fit = self.empty(n * max(nr, 1) * FIT_DOUBLES, t.float64)
        cr = self.empty(n * max(nc, 1), t.float64)
        out = self.empty(int(seg_len.sum()), t.float32) if want_edc else None
        scratch = self.empty(n * EDC_SCRATCH_DOUBLES, t.float64)
        flat = [v for r in ranges for v in r]
        part, p_off, p_wgs, p_tiles = tiles if tiles is not None else (None, None, None, None)
        d_off, d_len, d_eoff, d_poff, d_pwgs, d_ptiles = self.job_tables(
            np.ascontiguousarray(seg_off, np.int64), seg_len, edc_off if want_edc else None,
            None if part is None else np.ascontiguousarray(p_off, np.int64),
            None if part is None else np.ascontiguousarray(p_wgs, np.int32),
            None if part is None else np.ascontiguousarray(p_tiles, np.int32))
        check(self.lib.ira_edc_fits(_ptr(x_dev), _ptr(d_off), _ptr(d_len), n, int(seg_len.max()) if n else 0,
                                    float(eps), float(floor_db), float(t_mul), float(t_div), _lib.dbl_array(flat), nr,
                                    int(min_points), _lib.dbl_array(list(cross)), nc, _ptr(fit), _ptr(cr), _ptr(out),
                                    _ptr(d_eoff), _ptr(scratch), _ptr(part), _ptr(d_poff), _ptr(d_pwgs), _ptr(d_ptiles),
                                    self.stream),
              "ira_edc_fits")
        return (fit[: n * nr * FIT_DOUBLES].view(n, nr, FIT_DOUBLES) if nr else None,
                cr[: n * nc].view(n, nc) if nc else None, out, edc_off)

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
        d_off, d_len = self.job_tables(off, lens)
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
                    floor_db: float, precision: int = 32, frame_sel: Optional[List[np.ndarray]] = None,
                    frame_major: bool = False):
        """
        STFT magnitude (dB) of segments starting at seg_off with nframes[s] valid frames each.
        Returns (out flat f32 device, out_off host int64); out[s] is a C-contiguous (n_fft/2+1, T_s) matrix.
        frame_sel: optional per-segment arrays of frame indices (then T_s = len(frame_sel[s])).
        frame_major=True asks for the transposed (T_s, n_fft/2+1) layout (ira_stft_mag_db_tf; see stft_frame_major_ok).
        """
        t = self.torch
        n = int(seg_off.size)
        f = n_fft // 2 + 1
        if self.stft_generic_needed(n_fft):
            return self._stft_generic(x_dev, seg_off, nframes, int(n_fft), int(hop), use_hann, floor_db, frame_sel,
                                      frame_major)
        if frame_sel is not None:
            cols = np.array([int(s.size) for s in frame_sel], dtype=np.int32)
            sel_off = np.zeros(n, dtype=np.int64)
            if n > 1:
                sel_off[1:] = np.cumsum(cols[:-1])
            if cols.sum():
                sel, sel_off_dev = self.job_tables(np.concatenate(frame_sel).astype(np.int32), sel_off)
            else:
                sel, sel_off_dev = self.empty(1, t.int32), self.to_dev(sel_off)
        else:
            cols = np.ascontiguousarray(nframes, dtype=np.int32)
            sel = None
            sel_off_dev = None
        out_off = np.zeros(n, dtype=np.int64)
        sizes = cols.astype(np.int64) * f
        if n > 1:
            out_off[1:] = np.cumsum(sizes[:-1])
        out = self.empty(int(sizes.sum()), t.float32)
        d_off, d_cols, d_ooff = self.job_tables(seg_off, cols, out_off)
        self.event_tag = f"[f{precision},n{n_fft}{',sel' if frame_sel is not None else ''}]"
        fn = self.lib.ira_stft_mag_db_tf if frame_major else self.lib.ira_stft_mag_db
        check(fn(_ptr(x_dev), _ptr(d_off), _ptr(d_cols), n, int(cols.max()) if n else 0, int(n_fft), int(hop),
                 _ptr(self.window(n_fft, use_hann, precision)), _ptr(self.twiddle(n_fft, precision)), int(precision),
                 float(floor_db), _ptr(out), _ptr(d_ooff), _ptr(sel), _ptr(sel_off_dev), self.stream),
              "ira_stft_mag_db_tf" if frame_major else "ira_stft_mag_db")
        self.event_tag = ""
        return out, out_off, cols

    def stft_logbin(self, x_dev, seg_off: np.ndarray, nframes: np.ndarray, n_fft: int, hop: int, use_hann: bool,
                    floor_db: float, k_base: int, first: np.ndarray, count: np.ndarray):
        """Fused float64 STFT + log-bin aggregation (ira_stft_logbin; n_fft 8192): (curves device, curves_off host)."""
        t = self.torch
        n = int(seg_off.size)
        nbins = int(first.size)
        cols = np.ascontiguousarray(nframes, dtype=np.int32)
        sizes = cols.astype(np.int64) * nbins
        out_off = np.zeros(n, dtype=np.int64)
        if n > 1:
            out_off[1:] = np.cumsum(sizes[:-1])
        out = self.empty(int(sizes.sum()), t.float32)
        d_off, d_cols, d_ooff, d_f, d_c = self.job_tables(seg_off, cols, out_off, first.astype(np.int32),
                                                           count.astype(np.int32))
        self.event_tag = f"[f64,n{n_fft}]"
        check(self.lib.ira_stft_logbin(_ptr(x_dev), _ptr(d_off), _ptr(d_cols), n, int(cols.max()) if n else 0,
                                       int(n_fft), int(hop), _ptr(self.window(n_fft, use_hann, 64)),
                                       _ptr(self.twiddle(n_fft, 64)), 64, float(floor_db), int(k_base), _ptr(d_f),
                                       _ptr(d_c), nbins, _ptr(out), _ptr(d_ooff), self.stream), "ira_stft_logbin")
        self.event_tag = ""
        return out, out_off

    @staticmethod
    def stft_generic_needed(n_fft: int) -> bool:
        """Frame sizes the register / LDS STFT kernels do not take (they need a power of two in [64, 16384]): the reference
        accepts any positive n_fft (numpy.fft.rfft of arbitrary length, spectrogram.py:150), e.g. `--nfft 6000`."""
        n = int(n_fft)
        return n < 64 or n > 16384 or (n & (n - 1)) != 0

    def _stft_generic(self, x_dev, seg_off, nframes, n_fft: int, hop: int, use_hann: bool, floor_db: float, frame_sel,
                      frame_major: bool):
        """STFT of an arbitrary frame size through the arbitrary-length transforms: every frame is one element of
        ira_rfft_any / ira_rfft_smooth (float64, Hann window of length n_fft = numpy.hanning(n_fft), frames of equal length
        ride two per complex transform), then ira_spectrum_mag_phase's dB conversion (the same max(|X|, 10^(floor/20))
        -> 20 log10 -> float32 as a11).  The natural result is frame-major (T, F); the (F, T) layout of the reference is
        produced by a per-segment transposed copy (plumbing, only when a caller asks for it)."""
        t = self.torch
        n = int(seg_off.size)
        f = n_fft // 2 + 1
        if frame_sel is not None:
            frames = [np.asarray(s_, dtype=np.int64) for s_ in frame_sel]
        else:
            frames = [np.arange(int(c), dtype=np.int64) for c in nframes]
        cols = np.array([fr.size for fr in frames], dtype=np.int32)
        sizes = cols.astype(np.int64) * f
        out_off = np.zeros(n, dtype=np.int64)
        if n > 1:
            out_off[1:] = np.cumsum(sizes[:-1])
        total_frames = int(cols.sum())
        if total_frames == 0:
            return self.empty(1, t.float32), out_off, cols
        xoff = np.concatenate([int(seg_off[i]) + fr * int(hop) for i, fr in enumerate(frames)])
        lengths = np.full(total_frames, n_fft, dtype=np.int32)
        spec, spec_off = self.rfft_any(x_dev, xoff, lengths, bool(use_hann))
        mag, _ = self.spectrum_mag_phase(spec, spec_off, lengths, float(floor_db), want_phase=False)
        if frame_major:
            return mag, out_off, cols
        out = self.empty(int(sizes.sum()), t.float32)
        for i in range(n):
            c = int(cols[i])
            if c:
                o = int(out_off[i])
                out[o : o + c * f].view(f, c).copy_(mag[o : o + c * f].view(c, f).t())
        return out, out_off, cols

    @staticmethod
    def stft_logbin_ok(n_fft: int) -> bool:
        return int(n_fft) == 8192

    @staticmethod
    def stft_frame_major_ok(n_fft: int, precision: int) -> bool:
        """Configurations that deliver the frame-major (T, F) layout directly: ira_stft_mag_db_tf's, and every frame size
        that goes through the arbitrary-length path (_stft_generic)."""
        return (int(n_fft), int(precision)) in ((4096, 32), (8192, 64)) or Engine.stft_generic_needed(n_fft)

    # ------------------------------------------------------------------ a9/a17: arbitrary-length f64 DFTs
    workspace_budget_bytes = 48 << 30   # cap for the Bluestein work + filter arrays of one chunk

    # Convolution sizes the Bluestein kernels take: 2^k and 3 * 2^k.  three_pow2_sizes = False restricts the choice to
    # powers of two (round 2's behaviour; A/B).
    three_pow2_sizes = True

    def conv_size_for(self, need: int) -> int:
        """Smallest supported convolution size M >= need (a linear convolution of `need` distinct lags)."""
        return conv_size(need, self.three_pow2_sizes)

    def long_tables(self, m: int):
        key = ("long", int(m))
        if key not in self._tables:
            import ctypes
            l1, l2 = ctypes.c_int32(0), ctypes.c_int32(0)
            check(self.lib.ira_fft_split(int(m), ctypes.byref(l1), ctypes.byref(l2)), "ira_fft_split")
            n1, n2 = l1.value, l2.value

            def tab(count, period):
                ang = -2.0 * np.pi * np.arange(count, dtype=np.float64) / float(period)
                return self._table_to_dev(np.stack([np.cos(ang), np.sin(ang)], axis=1))

            self._tables[key] = (tab(n1, n1), tab(n2, n2), tab(n2, m))
        return self._tables[key]

    # Lengths of the form 2^a 3^b 5^c that split into two factors <= 1024 (480000, 2^19, ...) take the direct two-pass
    # mixed-radix transform (ira_rfft_smooth / ira_band_irfft_smooth) instead of Bluestein.  Set False for an A/B.
    smooth_ffts = True

    def smooth_split(self, n: int):
        """(n1, n2) if the library has a direct transform for length n, else None (cached)."""
        key = ("smooth?", int(n))
        if key not in self._tables:
            import ctypes
            a, b = ctypes.c_int32(0), ctypes.c_int32(0)
            rc = self.lib.ira_fft_smooth_split(int(n), ctypes.byref(a), ctypes.byref(b)) if n < (1 << 31) else -3
            self._tables[key] = (a.value, b.value) if rc == 0 else None
        return self._tables[key] if self.smooth_ffts else None

    def smooth_tables(self, n: int):
        key = ("smooth", int(n))
        if key not in self._tables:
            n1, n2 = self.smooth_split(n)

            def tab(count, period):
                ang = -2.0 * np.pi * np.arange(count, dtype=np.float64) / float(period)
                return self._table_to_dev(np.stack([np.cos(ang), np.sin(ang)], axis=1))

            self._tables[key] = (tab(n1, n1), tab(n2, n2), tab(n2, n))
        return self._tables[key]

    def _chunks_of(self, idx: np.ndarray, bytes_per_job: int):
        step = max(1, int(self.workspace_budget_bytes // max(1, bytes_per_job)))
        for i in range(0, idx.size, step):
            yield idx[i : i + step]

    def _chunks_by_size(self, need: np.ndarray):
        """Group element indices by the convolution size they need, then cut each group to the workspace budget."""
        ms = np.array([self.conv_size_for(int(v)) for v in need], dtype=np.int64)
        for m in sorted(set(ms.tolist())):
            idx = np.nonzero(ms == m)[0]
            per = 16 * m * 2                           # work + (worst case) one filter per element
            step = max(1, int(self.workspace_budget_bytes // per))
            for i in range(0, idx.size, step):
                yield int(m), idx[i : i + step]

    # Chirp-filter spectra depend only on (length, M): they are PLAN data, like twiddle tables, and are kept in LRU pools
    # of device slots -- one pool per (M, stream), all pools together within filter_cache_bytes -- so that repeated
    # lengths (every step of a batch job, every band pair of a file) do not rebuild them.  A pool starts at twice the
    # slots its first call needs and doubles when a call needs more; when the budget is full the pools used least
    # recently are dropped.  filter_cache_bytes = 0 disables the cache.
    filter_cache_bytes = 16 << 30

    def _filter_pool(self, m: int, need_slots: int):
        """The pool for size m on the current stream with room for need_slots filters of one call, or None."""
        t = self.torch
        slot_doubles = 2 * int(m)
        limit = int(self.filter_cache_bytes // (8 * slot_doubles))
        if limit < need_slots:
            return None
        key = (int(m), int(self.stream))
        self._filter_tick += 1
        pool = self._filter_pools.get(key)
        if pool is not None and pool["cap"] >= need_slots:
            pool["last"] = self._filter_tick
            return pool
        cap = max(2 * need_slots, 32, 2 * pool["cap"] if pool is not None else 0)
        cap = min(cap, limit)
        self._filter_pools.pop(key, None)                  # a pool that grows starts again (its filters are rebuilt on demand)
        total = lambda: sum(q["cap"] * 16 * k[0] for k, q in self._filter_pools.items())
        while self._filter_pools and total() + cap * 8 * slot_doubles > self.filter_cache_bytes:
            victim = min(self._filter_pools, key=lambda k: self._filter_pools[k]["last"])
            del self._filter_pools[victim]
        pool = dict(buf=self.empty(cap * slot_doubles, t.float64), slot_of={}, length_of=[None] * cap, tick=0,
                    used=[0] * cap, cap=cap, last=self._filter_tick)
        self._filter_pools[key] = pool
        return pool

    def forget_filters(self) -> None:
        """Forget which chirp-filter spectra the pools hold (their buffers stay): the next call of every length rebuilds
        its filter.  bench.py's `value_new_lengths` calls this before every step -- the rate of a job whose fr / filter
        segment lengths never repeat (they are data dependent: N - argmax|x|), against the pooled rate of a job that sees
        the same lengths again."""
        for pool in self._filter_pools.values():
            cap = pool["cap"]
            pool["slot_of"], pool["length_of"], pool["used"] = {}, [None] * cap, [0] * cap

    def _filters(self, lengths: np.ndarray, m: int):
        """Chirp-filter spectra for the distinct lengths in `lengths`; returns (bfilt device, bidx int32 host)."""
        t = self.torch
        uniq, inv = np.unique(lengths.astype(np.int32), return_inverse=True)
        t1, t2, tf = self.long_tables(m)
        slot_doubles = 2 * int(m)
        pool = self._filter_pool(m, int(uniq.size))
        if pool is None:                          # does not fit the budget: build a private array for this call
            bf = self.empty(int(uniq.size) * slot_doubles, t.float64)
            d_l = self.to_dev(uniq.astype(np.int32))
            check(self.lib.ira_bluestein_filter(_ptr(d_l), int(uniq.size), int(m), _ptr(t1), _ptr(t2), _ptr(tf),
                                                _ptr(bf), self.stream), "ira_bluestein_filter")
            return bf, inv.astype(np.int32)
        cap = pool["cap"]
        pool["tick"] += 1
        tick = pool["tick"]
        slots = np.empty(uniq.size, dtype=np.int32)
        missing = []
        for j, ln in enumerate(uniq.tolist()):
            s = pool["slot_of"].get(ln)
            if s is None:
                missing.append(j)
            else:
                slots[j] = s
                pool["used"][s] = tick
        if missing:
            # victims: least recently used slots not needed by this call
            order = sorted(range(cap), key=lambda s: pool["used"][s])
            victims = [s for s in order if pool["used"][s] != tick][: len(missing)]
            for j, s in zip(missing, victims):
                old_len = pool["length_of"][s]
                if old_len is not None:
                    del pool["slot_of"][old_len]
                ln = int(uniq[j])
                pool["slot_of"][ln] = s
                pool["length_of"][s] = ln
                pool["used"][s] = tick
                slots[j] = s
            # build the missing filters slot by slot runs (contiguous slots in one launch where possible)
            todo = sorted((int(slots[j]), int(uniq[j])) for j in missing)
            i = 0
            while i < len(todo):
                k = i
                while k + 1 < len(todo) and todo[k + 1][0] == todo[k][0] + 1:
                    k += 1
                run = todo[i : k + 1]
                d_l = self.to_dev(np.array([ln for _, ln in run], dtype=np.int32))
                dst = pool["buf"][run[0][0] * slot_doubles:]
                check(self.lib.ira_bluestein_filter(_ptr(d_l), len(run), int(m), _ptr(t1), _ptr(t2), _ptr(tf), _ptr(dst),
                                                    self.stream), "ira_bluestein_filter")
                i = k + 1
        return pool["buf"], slots[inv].astype(np.int32)

    # A channel's transforms are its own.  Round 2 let two real signals of equal length from DIFFERENT channels share one
    # complex transform (z = x1 + i*x2) and the odd band of one channel share an inverse with the odd band of another:
    # cheaper by half, but then the last bits of a channel's spectrum depend on which channel happened to be its partner,
    # i.e. on batch composition and shard boundaries (SURVEY.md section 8e asks for byte-identical records for every
    # world size).  Now a single real signal of EVEN length rides a HALF-length complex transform instead (x[2m] + i
    # x[2m+1]: the same saving), bands are paired only within one channel, and an odd band out takes the half-length
    # inverse.  pair_across_channels = True restores round 2's pairing (A/B; results then agree to ~1e-16 of the larger
    # signal, not bit for bit).  pair_real_ffts = False additionally forbids pairing the bands of one channel.
    pair_real_ffts = True
    pair_across_channels = False
    # A real signal of EVEN (non-smooth) length L is transformed as the complex sequence x[2m] + i*x[2m+1] of length
    # L/2 (half the Bluestein convolution size) and split with a twiddle.  Set False for an A/B.
    half_real_ffts = True
    # The untangling X[k] = E[k] + W^k O[k] of such a half-length transform happens where Z[k] and Z[n - k] already meet:
    # in the second pass of the direct transform (smooth lengths), and in the dB / phase kernel for the Bluestein spectra of
    # the fr / filter blocks (packed spectra).  False restores round 3's separate split passes (A/B).
    fuse_half_split = True
    # Band inverses of smooth lengths: jobs whose bands span few bins (third-octave bands below ~800 Hz of a 10 s file)
    # skip the first pass and the n-point work array (ira_band_irfft_smooth, job_info_dev).  False = both passes for every
    # job (A/B).
    sparse_bands = True
    last_band_info = ()
    last_band_info_half = ()

    @staticmethod
    def _pair_by_key(idx: np.ndarray, keys: np.ndarray):
        """Pairs of entries of idx whose keys are equal; leftovers are paired with -1.  Order-stable.
        (Array arithmetic: run lengths of the sorted keys, even positions of a run lead a pair -- the metrics pipeline calls
        this for 256 channels per step, and a Python loop per entry was part of what bounds the bundle configuration.)"""
        idx = np.asarray(idx, dtype=np.int64)
        if idx.size == 0:
            return np.zeros(0, dtype=np.int64), np.zeros(0, dtype=np.int64)
        order = idx[np.argsort(keys[idx], kind="stable")]
        sk = keys[order]
        new_run = np.r_[True, sk[1:] != sk[:-1]]
        run_start = np.flatnonzero(new_run)
        run_id = np.cumsum(new_run) - 1
        pos = np.arange(order.size) - run_start[run_id]                  # position inside the run
        run_len = np.diff(np.r_[run_start, order.size])[run_id]
        lead = (pos % 2) == 0
        has_partner = lead & (pos + 1 < run_len)
        first = order[lead]
        nxt = np.r_[order[1:], -1]
        second = np.where(has_partner, nxt, -1)[lead]
        return first.astype(np.int64), second.astype(np.int64)

    @staticmethod
    def _pair_bands(keys: np.ndarray, width: np.ndarray):
        """Like _pair_by_key over all entries, but a group of odd size leaves its entry of smallest `width` alone (the first
        of them on a tie) and pairs the others in order."""
        order = np.argsort(keys, kind="stable")
        sk = keys[order]
        starts = np.flatnonzero(np.r_[True, sk[1:] != sk[:-1]]) if order.size else np.zeros(0, dtype=np.int64)
        sizes = np.diff(np.r_[starts, order.size])
        if order.size and np.all(sizes == sizes[0]):
            # every group has the same size (every channel of a report has the same bands): array arithmetic, same output
            # order as the loop below (a group's pairs, then its lone entry)
            g, sz = int(starts.size), int(sizes[0])
            grp = order.reshape(g, sz)
            if sz % 2 == 0:
                return grp[:, 0::2].reshape(-1).astype(np.int64), grp[:, 1::2].reshape(-1).astype(np.int64)
            lone_pos = np.argmin(width[grp], axis=1)                     # the first minimum, like argmin over the group
            keep = np.ones((g, sz), dtype=bool)
            keep[np.arange(g), lone_pos] = False
            rest = grp[keep].reshape(g, sz - 1)
            lone = grp[np.arange(g), lone_pos][:, None]
            first = np.concatenate([rest[:, 0::2], lone], axis=1).reshape(-1)
            second = np.concatenate([rest[:, 1::2], np.full((g, 1), -1, dtype=order.dtype)], axis=1).reshape(-1)
            return first.astype(np.int64), second.astype(np.int64)
        first, second = [], []
        i = 0
        while i < order.size:
            j = i
            while j < order.size and keys[order[j]] == keys[order[i]]:
                j += 1
            grp = order[i:j]
            lone = -1
            if grp.size % 2:
                lone = int(grp[int(np.argmin(width[grp]))])
                grp = grp[grp != lone]
            first.extend(grp[0::2].tolist()); second.extend(grp[1::2].tolist())
            if lone >= 0:
                first.append(lone); second.append(-1)
            i = j
        return np.asarray(first, dtype=np.int64), np.asarray(second, dtype=np.int64)

    def rfft_any(self, x_dev, xoff: np.ndarray, lengths: np.ndarray, use_hann: bool,
                 data_len: Optional[np.ndarray] = None, win_len: Optional[np.ndarray] = None):
        """Half spectra of arbitrary-length segments: (spec, spec_off); see _rfft_any."""
        spec, spec_off, _ = self._rfft_any(x_dev, xoff, lengths, use_hann, data_len, win_len, False)
        return spec, spec_off

    def rfft_any_packed(self, x_dev, xoff: np.ndarray, lengths: np.ndarray, use_hann: bool):
        """The same for a caller whose only consumer is spectrum_mag_phase(packed=...): (spec, spec_off, packed) -- even-length
        elements that ride a half-length Bluestein transform stay PACKED (see _rfft_any); packed is None when none is."""
        return self._rfft_any(x_dev, xoff, lengths, use_hann, None, None, True)

    def _rfft_any(self, x_dev, xoff: np.ndarray, lengths: np.ndarray, use_hann: bool,
                  data_len: Optional[np.ndarray], win_len: Optional[np.ndarray], packed_ok: bool):
        """
        Half spectra (complex f64) of x[xoff[e] : xoff[e]+L[e]] (* hanning) for every element.
        data_len / win_len (optional, per element): numpy.fft.rfft(x[:d] * hanning(w)[:d], n=L) -- d samples are read
        (d < L zero-pads, d > L truncates) under the Hann window of length w (reference group_delay.py:95-109).
        Returns (spec float64 device viewed as (total_bins, 2), spec_off int64 host in complex elements, packed) -- always
        three values; the public wrappers are rfft_any (two) and rfft_any_packed (three).
        packed_ok: the caller's consumer is spectrum_mag_phase(packed=...) -- even-length elements that ride a half-length
        Bluestein transform may then stay PACKED: spec_off[e] holds the L/2 values Z = DFT(x[2m] + i x[2m+1]) instead of
        the L/2 + 1 bins, and the third return value marks them (int32 per element; None when nothing is packed).
        """
        t = self.torch
        n = int(xoff.size)
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        xoff = np.ascontiguousarray(xoff, dtype=np.int64)
        padded = data_len is not None or win_len is not None
        if padded:
            data_len = np.ascontiguousarray(lengths if data_len is None else data_len, dtype=np.int32)
            win_len = np.ascontiguousarray(lengths if win_len is None else win_len, dtype=np.int32)
        bins = lengths.astype(np.int64) // 2 + 1
        spec_off = np.zeros(n, dtype=np.int64)
        if n > 1:
            spec_off[1:] = np.cumsum(bins[:-1])
        spec = self.empty(int(bins.sum()) * 2, t.float64)
        # ---- smooth lengths: direct two-pass transform, one call per distinct length ------------------------------------
        # Even lengths whose half is smooth too ride a HALF-length complex transform each (interleave mode); the rest one
        # full-length transform each -- or, with pair_across_channels, two per transform as in round 2.
        rest = np.ones(n, dtype=bool)
        for L in np.unique(lengths):
            full_ok = self.smooth_split(int(L)) is not None
            half_ok = (int(L) % 2 == 0 and int(L) >= 128 and self.half_real_ffts and not self.pair_across_channels
                       and self.smooth_split(int(L) // 2) is not None)
            if not (full_ok or half_ok):
                continue
            grp = np.nonzero(lengths == L)[0]
            rest[grp] = False
            nt = int(L) // 2 if half_ok else int(L)                  # transform length
            t1, t2, tf = self.smooth_tables(nt)
            for idx in self._chunks_of(grp, 32 * nt):
                if half_ok:
                    j1, j2 = idx.astype(np.int64), np.full(idx.size, -1, dtype=np.int64)
                elif self.pair_real_ffts and self.pair_across_channels and idx.size > 1:
                    j1, j2 = self._pair_by_key(idx, lengths)
                else:
                    j1, j2 = idx.astype(np.int64), np.full(idx.size, -1, dtype=np.int64)
                work = self.empty(int(j1.size) * 2 * nt, t.float64)
                paired = j2 >= 0
                safe = np.maximum(j2, 0)
                a_x2 = a_so2 = a_zo = zpair = None
                # round 4: with an even n1 the untangling of the half-length transform is part of its second pass
                # (mirror-pair tiles, ira_rfft_smooth without zpair scratch); fuse_half_split = False is the A/B
                fused = half_ok and self.fuse_half_split and self.smooth_split(nt)[0] % 2 == 0
                if fused:
                    pass
                elif half_ok:
                    a_x2 = (xoff[j1] + 1).astype(np.int64)
                    a_so2 = spec_off[j1].astype(np.int64)
                    a_zo = (np.arange(j1.size, dtype=np.int64) * nt)
                    zpair = self.empty(int(j1.size) * 2 * nt, t.float64)
                elif paired.any():
                    a_x2 = np.where(paired, xoff[safe], -1).astype(np.int64)
                    a_so2 = np.where(paired, spec_off[safe], 0).astype(np.int64)
                    zoff = np.cumsum(np.where(paired, int(L), 0)) - np.where(paired, int(L), 0)
                    zpair = self.empty(int(paired.sum()) * 2 * int(L), t.float64)
                    a_zo = zoff.astype(np.int64)
                a_dl = a_wl = a_dl2 = a_wl2 = None
                if padded:
                    a_dl, a_wl = data_len[j1], win_len[j1]
                    if not half_ok:
                        a_dl2, a_wl2 = data_len[safe], win_len[safe]
                d_xo, d_so, d_x2, d_so2, d_zo, d_dl, d_wl, d_dl2, d_wl2 = self.job_tables(
                    xoff[j1], spec_off[j1], a_x2, a_so2, a_zo, a_dl, a_wl, a_dl2, a_wl2)
                check(self.lib.ira_rfft_smooth(_ptr(x_dev), _ptr(d_xo), nt, int(j1.size), 1 if use_hann else 0,
                                               _ptr(t1), _ptr(t2), _ptr(tf), _ptr(work), _ptr(spec), _ptr(d_so),
                                               _ptr(d_x2), _ptr(d_so2), _ptr(zpair), _ptr(d_zo), _ptr(d_dl), _ptr(d_wl),
                                               _ptr(d_dl2), _ptr(d_wl2), 1 if half_ok else 0, self.stream), "ira_rfft_smooth")
        packed = np.zeros(n, dtype=np.int32)
        if not rest.any():
            return spec, spec_off, None
        rest_idx = np.nonzero(rest)[0]
        # ---- everything else: Bluestein, grouped by the power-of-two convolution size ------------------------------------
        # Jobs: "half" = one real signal of EVEN length carried as x[2m] + i*x[2m+1] (a complex transform of L/2: half
        # the convolution size); "pair" = two signals of equal length as x1 + i*x2; "single".
        use_half = self.half_real_ffts and not padded
        lr = lengths[rest_idx]
        is_half = ((lr % 2 == 0) & (lr >= 8)) if use_half else np.zeros(rest_idx.size, dtype=bool)
        others = rest_idx[~is_half]
        if self.pair_real_ffts and self.pair_across_channels and others.size > 1:
            p1, p2 = self._pair_by_key(others, lengths)
        else:
            p1, p2 = others.astype(np.int64), np.full(others.size, -1, dtype=np.int64)
        e1 = np.concatenate([rest_idx[is_half].astype(np.int64), p1])
        e2 = np.concatenate([np.full(int(is_half.sum()), -1, dtype=np.int64), p2])
        half = np.concatenate([np.ones(int(is_half.sum()), dtype=bool), np.zeros(p1.size, dtype=bool)])
        jlen = np.where(half, lengths[e1] // 2, lengths[e1]).astype(np.int32)        # transform length of the job
        # lags the convolution must keep apart: 2 l - 1 when all l outputs of a complex transform are wanted (half and
        # paired jobs), l + l/2 for a single real signal (outputs k <= l/2 only)
        jl64 = jlen.astype(np.int64)
        need = np.where(half | (e2 >= 0), 2 * jl64 - 1, jl64 + jl64 // 2)
        for lm, sel in self._chunks_by_size(need):
            t1, t2, tf = self.long_tables(lm)
            j1, j2, jh, jl = e1[sel], e2[sel], half[sel], jlen[sel]
            bf, bidx = self._filters(jl, lm)
            work = self.empty(int(sel.size) * 2 * lm, t.float64)
            paired = j2 >= 0
            two = paired | jh                                   # jobs that carry a second "signal"
            safe = np.maximum(j2, 0)
            a_x2 = a_so2 = a_zo = zpair = None
            keep_packed = bool(packed_ok and self.fuse_half_split and jh.any() and not paired.any())
            if keep_packed:
                # the half-length transforms land in the spectrum array itself (L/2 values in the element's L/2 + 1 slots)
                # and stay packed: the dB / phase kernel untangles them as it reads them
                a_x2 = np.where(jh, xoff[j1] + 1, -1).astype(np.int64)
                a_so2 = np.zeros(j1.size, dtype=np.int64)
                a_zo = spec_off[j1].astype(np.int64)
                zpair = spec
                packed[j1[jh]] = 1
            elif two.any():
                a_x2 = np.where(jh, xoff[j1] + 1, np.where(paired, xoff[safe], -1)).astype(np.int64)
                a_so2 = np.where(paired, spec_off[safe], 0).astype(np.int64)
                zlen = np.where(two, jl.astype(np.int64), 0)
                a_zo = (np.cumsum(zlen) - zlen).astype(np.int64)
                zpair = self.empty(int(zlen.sum()) * 2, t.float64)
            a_dl = a_wl = a_dl2 = a_wl2 = a_il = None
            if padded:
                a_dl, a_wl, a_dl2, a_wl2 = data_len[j1], win_len[j1], data_len[safe], win_len[safe]
            elif jh.any():
                a_dl = jl
                a_wl = np.where(jh, lengths[j1], jl).astype(np.int32)       # the Hann window belongs to the REAL signal
                a_il = jh.astype(np.int32)
            d_xo, d_l, d_bi, d_so, d_x2, d_so2, d_zo, d_dl, d_wl, d_dl2, d_wl2, d_il = self.job_tables(
                xoff[j1], jl, bidx, spec_off[j1], a_x2, a_so2, a_zo, a_dl, a_wl, a_dl2, a_wl2, a_il)
            if not padded and a_dl is not None:
                d_dl2, d_wl2 = d_dl, d_wl
            check(self.lib.ira_rfft_any(_ptr(x_dev), _ptr(d_xo), _ptr(d_l), int(sel.size), 1 if use_hann else 0, lm,
                                        _ptr(t1), _ptr(t2), _ptr(tf), _ptr(bf), _ptr(d_bi), _ptr(work), _ptr(spec),
                                        _ptr(d_so), _ptr(d_x2), _ptr(d_so2), _ptr(zpair), _ptr(d_zo), int(jl.max()),
                                        _ptr(d_dl), _ptr(d_wl), _ptr(d_dl2), _ptr(d_wl2), _ptr(d_il),
                                        1 if keep_packed else 0, self.stream),
                  "ira_rfft_any")
        return spec, spec_off, (packed if (packed_ok and packed.any()) else None)

    def band_tile_layout(self, nt: int, half_out: bool):
        """(tiles, workgroups) of the partial tile energies ira_band_irfft_smooth leaves per signal (ira.h; cached)."""
        key = ("tiles", int(nt), bool(half_out))
        if key not in self._tables:
            import ctypes
            a, b = ctypes.c_int32(0), ctypes.c_int32(0)
            check(self.lib.ira_band_tile_layout(int(nt), 1 if half_out else 0, ctypes.byref(a), ctypes.byref(b)),
                  "ira_band_tile_layout")
            self._tables[key] = (a.value, b.value)
        return self._tables[key]

    def band_irfft(self, spec_dev, spec_off: np.ndarray, lengths: np.ndarray, band_params: np.ndarray,
                   freq_val: np.ndarray, y_dev, y_off: np.ndarray, want_tiles: bool = False):
        """
        Masked inverse transforms, ONE BAND PER ENTRY: entry j filters the half spectrum at spec_off[j] (length
        lengths[j], bin step freq_val[j]) with the 8-double mask record band_params[j] and writes lengths[j]
        float32 samples at y_off[j].  Two bands of the SAME spectrum are inverse-transformed together (y1 + i*y2; see
        ira_band_irfft in include/ira.h); a band left over takes a half-length inverse when its length allows it (smooth
        family, even length) and a full-length one otherwise.  (pair_across_channels = True: round 2's pairing of entries
        with the same length and bin step from different channels.)
        want_tiles (round 5): the smooth-length inverses also leave the partial energies of every band signal's EDC tiles;
        returns (part float64 device, part_off int64 per entry (-1: none, e.g. Bluestein lengths), part_wgs and part_tiles
        int32 per entry) for edc_fits(tiles=...), else None.
        """
        t = self.torch
        self.last_band_info = []          # the job_info records of this call's launches (tests, diagnostics) ...
        self.last_band_info_half = []     # ... and whether the launch was the half-length (single band) one
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        spec_off = np.ascontiguousarray(spec_off, dtype=np.int64)
        y_off = np.ascontiguousarray(y_off, dtype=np.int64)
        band_params = np.ascontiguousarray(band_params, dtype=np.float64).reshape(-1, 8)
        freq_val = np.ascontiguousarray(freq_val, dtype=np.float64)
        # pair key: same transform length AND same float64 bin step (AND same spectrum when cross-channel pairing is off)
        cols = [lengths.astype(np.float64), freq_val]
        if not self.pair_across_channels:
            cols.append(spec_off.astype(np.float64))
        _, key = np.unique(np.stack(cols, axis=1), axis=0, return_inverse=True)
        if self.pair_real_ffts and self.sparse_bands:
            # the band left over in a group of odd size is the NARROWEST one (round 4): alone it is a narrow job of the
            # half-length inverse and skips the first pass, paired with a wide neighbour it would not
            kind, width = band_params[:, 0], np.full(lengths.size, np.inf)
            width[kind == 1.0] = band_params[kind == 1.0, 4]
            width[kind == 3.0] = band_params[kind == 3.0, 4] - band_params[kind == 3.0, 1]
            width[(kind != 1.0) & (kind != 2.0) & (kind != 3.0)] = 0.0
            j1, j2 = self._pair_bands(key.reshape(-1), width)
        elif self.pair_real_ffts:
            j1, j2 = self._pair_by_key(np.arange(lengths.size), key.reshape(-1))
        else:
            j1, j2 = np.arange(lengths.size, dtype=np.int64), np.full(lengths.size, -1, dtype=np.int64)
        jl = lengths[j1]
        safe = np.maximum(j2, 0)
        has2 = j2 >= 0
        el_par = np.zeros((j1.size, 2, 8), dtype=np.float64)
        el_par[:, 0, :] = band_params[j1]
        el_par[has2, 1, :] = band_params[safe[has2]]
        el_so2 = np.where(has2, spec_off[safe], spec_off[j1]).astype(np.int64)
        el_y2 = np.where(has2, y_off[safe], -1).astype(np.int64)
        rest = np.ones(j1.size, dtype=bool)
        launches = []                                         # (jobs, half-length?, transform length) of the smooth family
        for L in np.unique(jl):
            full_ok = self.smooth_split(int(L)) is not None
            half_ok = (int(L) % 2 == 0 and int(L) >= 128 and self.half_real_ffts
                       and self.smooth_split(int(L) // 2) is not None)
            for halves in (False, True):
                # single bands take the half-length inverse when the length allows it, pairs the full-length one
                if halves and not half_ok:
                    continue
                if not halves and not full_ok:
                    continue
                grp = np.nonzero((jl == L) & rest & ((~has2) if halves else (has2 | (not half_ok))))[0]
                if grp.size == 0:
                    continue
                rest[grp] = False
                nt = int(L) // 2 if halves else int(L)
                for sel in self._chunks_of(grp, 16 * nt):
                    launches.append((sel, halves, nt))
        # partial tile energies: one block of 2 x tiles x workgroups doubles per job, launch after launch in ONE buffer
        part = part_off = part_wgs = part_tiles = None
        bases = [0] * len(launches)
        if want_tiles and self.band_tile_energies:
            part_off = np.full(lengths.size, -1, dtype=np.int64)
            part_wgs = np.zeros(lengths.size, dtype=np.int32)
            part_tiles = np.zeros(lengths.size, dtype=np.int32)
            total = 0
            for li, (sel, halves, nt) in enumerate(launches):
                tiles, wgs = self.band_tile_layout(nt, halves)
                bases[li] = total
                blk = tiles * wgs
                first = total + np.arange(sel.size, dtype=np.int64) * (2 * blk)
                part_off[j1[sel]] = first
                part_wgs[j1[sel]] = wgs
                part_tiles[j1[sel]] = tiles
                two = has2[sel]
                part_off[j2[sel][two]] = first[two] + blk
                part_wgs[j2[sel][two]] = wgs
                part_tiles[j2[sel][two]] = tiles
                total += int(sel.size) * 2 * blk
            part = self.empty(total, t.float64) if total else None
        if True:
            if True:
                for li, (sel, halves, nt) in enumerate(launches):
                    t1, t2, tf = self.smooth_tables(nt)
                    work = self.empty(int(sel.size) * 2 * nt, t.float64)
                    d_so, d_bp, d_fv, d_y1, d_y2, d_so2 = self.job_tables(
                        spec_off[j1][sel], np.ascontiguousarray(el_par[sel]), np.ascontiguousarray(freq_val[j1][sel]),
                        np.ascontiguousarray(y_off[j1][sel]), np.ascontiguousarray(el_y2[sel]),
                        None if halves else np.ascontiguousarray(el_so2[sel]))
                    # round 4: narrow bands skip the first pass (ira.h, job_info_dev); sparse_bands = False is the A/B
                    info = self.empty(4 * int(sel.size), t.int32) if self.sparse_bands else None
                    check(self.lib.ira_band_irfft_smooth(_ptr(spec_dev), _ptr(d_so), nt, int(sel.size), _ptr(d_bp),
                                                         _ptr(d_fv), _ptr(t1), _ptr(t2), _ptr(tf), _ptr(work), _ptr(y_dev),
                                                         _ptr(d_y1), _ptr(d_y2), _ptr(d_so2), 1 if halves else 0,
                                                         _ptr(info), (_ptr(part) + 8 * bases[li]) if part is not None else None,
                                                         self.stream), "ira_band_irfft_smooth")
                    self.last_band_info.append(info)
                    self.last_band_info_half.append(bool(halves))
        tiles_out = (part, part_off, part_wgs, part_tiles) if part is not None else None
        if not rest.any():
            return tiles_out
        rest_idx = np.nonzero(rest)[0]
        for lm, sub in self._chunks_by_size(2 * jl[rest_idx].astype(np.int64) - 1):
            sel = rest_idx[sub]
            t1, t2, tf = self.long_tables(lm)
            bf, bidx = self._filters(jl[sel], lm)
            work = self.empty(int(sel.size) * 2 * lm, t.float64)
            d_so, d_l = self.to_dev(spec_off[j1][sel]), self.to_dev(jl[sel])
            d_bp = self.to_dev(np.ascontiguousarray(el_par[sel]))
            d_fv = self.to_dev(np.ascontiguousarray(freq_val[j1][sel]))
            d_bi = self.to_dev(bidx)
            d_y1 = self.to_dev(np.ascontiguousarray(y_off[j1][sel]))
            d_y2 = self.to_dev(np.ascontiguousarray(el_y2[sel]))
            d_so2 = self.to_dev(np.ascontiguousarray(el_so2[sel]))
            check(self.lib.ira_band_irfft(_ptr(spec_dev), _ptr(d_so), _ptr(d_l), int(sel.size), _ptr(d_bp), _ptr(d_fv),
                                          lm, _ptr(t1), _ptr(t2), _ptr(tf), _ptr(bf), _ptr(d_bi), _ptr(work),
                                          _ptr(y_dev), _ptr(d_y1), _ptr(d_y2), _ptr(d_so2), self.stream),
                  "ira_band_irfft")
        return tiles_out

    def spectrum_mag_phase(self, spec_dev, spec_off: np.ndarray, lengths: np.ndarray, floor_db: float,
                           want_phase: bool, packed: Optional[np.ndarray] = None):
        """mag_db (f32) [+ wrapped phase f64] laid out at the same per-element bin offsets as the spectra.  packed (int32 per
        element, from rfft_any(packed_ok=True)): elements whose spectrum is still the packed half-length transform."""
        t = self.torch
        n = int(spec_off.size)
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        total = int((lengths.astype(np.int64) // 2 + 1).sum())
        mag = self.empty(total, t.float32)
        ph = self.empty(total, t.float64) if want_phase else None
        d_so, d_l, d_pk = self.job_tables(spec_off, lengths, None if packed is None else np.ascontiguousarray(packed, np.int32))
        check(self.lib.ira_spectrum_mag_phase(_ptr(spec_dev), _ptr(d_so), _ptr(d_l), n, int(lengths.max()),
                                              float(floor_db), _ptr(mag), _ptr(d_so), _ptr(ph), _ptr(d_so), _ptr(d_pk),
                                              self.stream), "ira_spectrum_mag_phase")
        return mag, ph

    def phase_unwrap(self, phase_dev, off: np.ndarray, lengths: np.ndarray, unwrap: bool, degrees: bool,
                     as_float64: bool = False):
        """float32 (optionally degrees) unwrapped phase, or with as_float64 the float64 radians."""
        t = self.torch
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        total = int((lengths.astype(np.int64) // 2 + 1).sum())
        out = self.empty(total, t.float64 if as_float64 else t.float32)
        d_o, d_l = self.job_tables(off, lengths)
        check(self.lib.ira_phase_unwrap(_ptr(phase_dev), _ptr(d_o), _ptr(d_l), int(off.size), 1 if unwrap else 0,
                                        1 if degrees else 0, 0 if as_float64 else _ptr(out), _ptr(d_o),
                                        _ptr(out) if as_float64 else 0, self.stream), "ira_phase_unwrap")
        return out

    def log_smooth(self, mag_dev, off: np.ndarray, stride: np.ndarray, k_lo: np.ndarray, nsel: np.ndarray,
                   fstep: np.ndarray, window: int, bins_per_octave: int, through_float32: bool) -> bool:
        """Log-frequency smoothing of dB curves in place (ira_log_smooth_db).  Grid geometry exactly as the reference
        computes it on the host: log2 of the first / last selected float32 frequency, count = max(8, ceil(span * bpo)) + 1
        with bpo >= 16.  Returns False (nothing done) when a grid is too large for the kernel."""
        n = int(np.asarray(off).size)
        k_lo = np.ascontiguousarray(k_lo, dtype=np.int32)
        nsel = np.ascontiguousarray(nsel, dtype=np.int32)
        fstep = np.ascontiguousarray(fstep, dtype=np.float64)
        f_first = (k_lo.astype(np.float64) * fstep).astype(np.float32).astype(np.float64)
        f_last = ((k_lo.astype(np.float64) + nsel - 1) * fstep).astype(np.float32).astype(np.float64)
        with np.errstate(divide="ignore"):
            a, b = np.log2(f_first), np.log2(f_last)
        bpo = int(max(16, bins_per_octave))
        count = (np.maximum(8, np.ceil((b - a) * bpo)).astype(np.int64) + 1).astype(np.int32)
        if n == 0:
            return True
        if int(count.max()) > 2048 or not np.all(np.isfinite(a)):
            return False
        d = self.job_tables(np.ascontiguousarray(off, np.int64), np.ascontiguousarray(stride, np.int32), k_lo, nsel, fstep,
                             a, b, count)
        check(self.lib.ira_log_smooth_db(_ptr(mag_dev), *[_ptr(x) for x in d], n, int(count.max()), int(window),
                                         1 if through_float32 else 0, self.stream), "ira_log_smooth_db")
        return True

    def order_stats(self, values_dev, off: np.ndarray, count: np.ndarray, ranks: np.ndarray):
        """sorted(segment)[rank] for every (segment, rank): float64 device tensor (nseg, nranks); ranks is (nseg, nranks)."""
        t = self.torch
        ranks = np.ascontiguousarray(ranks, dtype=np.int64)
        nseg, nr = ranks.shape
        out = self.empty(nseg * nr, t.float64)
        d_o, d_c = self.to_dev(np.ascontiguousarray(off, np.int64)), self.to_dev(np.ascontiguousarray(count, np.int32))
        d_r = self.to_dev(ranks.reshape(-1))
        check(self.lib.ira_order_stats(_ptr(values_dev), _ptr(d_o), _ptr(d_c), int(nseg), _ptr(d_r), int(nr), _ptr(out),
                                       self.stream), "ira_order_stats")
        return out[: nseg * nr].view(nseg, nr)

    def diffusion(self, x_dev, xoff: np.ndarray, nframes: np.ndarray, win: int, hop: int, max_lag: int,
                  thr_rms: float, gauss_expected: float):
        """Per-window max|autocorr| and echo density (float32 device, flat) + host offsets (see ira_diffusion)."""
        t = self.torch
        n = int(xoff.size)
        nframes = np.ascontiguousarray(nframes, dtype=np.int32)
        out_off = np.zeros(n, dtype=np.int64)
        if n > 1:
            out_off[1:] = np.cumsum(nframes[:-1].astype(np.int64))
        total = int(nframes.astype(np.int64).sum())
        ac, ed = self.empty(total, t.float32), self.empty(total, t.float32)
        d_xo, d_nf, d_oo = self.to_dev(np.ascontiguousarray(xoff, np.int64)), self.to_dev(nframes), self.to_dev(out_off)
        check(self.lib.ira_diffusion(_ptr(x_dev), _ptr(d_xo), _ptr(d_nf), n, int(nframes.max()), int(win), int(hop),
                                     int(max_lag), float(thr_rms), float(gauss_expected), _ptr(ac), _ptr(ed),
                                     _ptr(d_oo), self.stream), "ira_diffusion")
        return ac, ed, out_off

    def diffusion_stereo(self, x_dev, loff: np.ndarray, roff: np.ndarray, nframes: np.ndarray, win: int, hop: int,
                         max_lag: int):
        """Per-window corr0 and IACC max of channel pairs (see ira_diffusion_stereo)."""
        t = self.torch
        n = int(loff.size)
        nframes = np.ascontiguousarray(nframes, dtype=np.int32)
        out_off = np.zeros(n, dtype=np.int64)
        if n > 1:
            out_off[1:] = np.cumsum(nframes[:-1].astype(np.int64))
        total = int(nframes.astype(np.int64).sum())
        c0, ia = self.empty(total, t.float32), self.empty(total, t.float32)
        d_lo, d_ro = self.to_dev(np.ascontiguousarray(loff, np.int64)), self.to_dev(np.ascontiguousarray(roff, np.int64))
        d_nf, d_oo = self.to_dev(nframes), self.to_dev(out_off)
        check(self.lib.ira_diffusion_stereo(_ptr(x_dev), _ptr(d_lo), _ptr(d_ro), _ptr(d_nf), n, int(nframes.max()),
                                            int(win), int(hop), int(max_lag), _ptr(c0), _ptr(ia), _ptr(d_oo),
                                            self.stream), "ira_diffusion_stereo")
        return c0, ia, out_off

    def group_delay(self, phase64_dev, off: np.ndarray, n_fft: np.ndarray, bin_step: np.ndarray, sample_rate_hz: float):
        """-numpy.gradient(phase, w) on the rad/sample axis of an n_fft-point rFFT (see ira_group_delay)."""
        t = self.torch
        n = int(off.size)
        nbins = (np.ascontiguousarray(n_fft, dtype=np.int64) // 2 + 1).astype(np.int32)
        gd = self.empty(int(nbins.astype(np.int64).sum()), t.float64)
        # Which formula numpy.gradient takes (uniform spacing iff every diff of the axis is bit-identical) depends on the
        # transform length, the bin step and the sample rate alone: decided once per distinct triple on the host, with the
        # reference's own expression (group_delay.py:113-124), instead of a sweep over every bin of every channel per call.
        bin_step = np.ascontiguousarray(bin_step, np.float64)
        known = np.empty(n, dtype=np.int32)
        for i in range(n):
            key = (int(nbins[i]), float(bin_step[i]), float(sample_rate_hz))
            f = self._gd_nonuniform.get(key)
            if f is None:
                w = 2.0 * np.pi * ((np.arange(key[0], dtype=np.float64) * key[1]) / key[2])
                d = np.diff(w)
                f = int(d.size > 0 and not bool(np.all(d == d[0])))
                self._gd_nonuniform[key] = f
            known[i] = f
        d_o, d_n, d_v, flags = self.job_tables(np.ascontiguousarray(off, np.int64), nbins, bin_step, known)
        check(self.lib.ira_group_delay(_ptr(phase64_dev), _ptr(d_o), _ptr(d_n), n, int(nbins.max()), _ptr(d_v),
                                       float(sample_rate_hz), _ptr(flags), 1, _ptr(gd), self.stream), "ira_group_delay")
        return gd

    _gd_nonuniform: dict = {}

    def spectrum_stats(self, mag_dev, off: np.ndarray, lengths: np.ndarray, freq_val: np.ndarray, f_min: float,
                       f_max: float, probe_hz: float = 1000.0):
        t = self.torch
        n = int(off.size)
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        out = self.empty(n * 8, t.float64)
        d_o, d_l, d_fv = self.job_tables(off, lengths, np.ascontiguousarray(freq_val, np.float64))
        check(self.lib.ira_spectrum_stats(_ptr(mag_dev), _ptr(d_o), _ptr(d_l), n, _ptr(d_fv), float(f_min),
                                          float(f_max), float(probe_hz), _ptr(out), self.stream), "ira_spectrum_stats")
        return out[: n * 8].view(n, 8)

    # ------------------------------------------------------------------ a14 / a15
    def waterfall_rel(self, mag_dev, mag_off: np.ndarray, nslices: np.ndarray, k_lo: int, nsel: int,
                      slice_max: bool, dyn_db: float):
        """(S_e, nsel) relative-dB slices per element; returns (out f32 device, out_off host)."""
        t = self.torch
        n = int(mag_off.size)
        nslices = np.ascontiguousarray(nslices, dtype=np.int32)
        sizes = nslices.astype(np.int64) * int(nsel)
        out_off = np.zeros(n, dtype=np.int64)
        if n > 1:
            out_off[1:] = np.cumsum(sizes[:-1])
        out = self.empty(int(sizes.sum()), t.float32)
        d_mo, d_ns, d_oo = self.job_tables(mag_off, nslices, out_off)
        check(self.lib.ira_waterfall_rel(_ptr(mag_dev), _ptr(d_mo), _ptr(d_ns), n, int(k_lo), int(nsel),
                                         1 if slice_max else 0, float(dyn_db), _ptr(out), _ptr(d_oo), self.stream),
              "ira_waterfall_rel")
        return out, out_off

    def logbin_aggregate(self, mag_dev, mag_off: np.ndarray, nframes: np.ndarray, k_base: int, first: np.ndarray,
                         count: np.ndarray, frame_major_rows: int = 0):
        """(nbins, T_e) float32 log-bin curves per element; returns (out device, out_off host).
        frame_major_rows = F when mag holds the (T, F) matrices of stft_mag_db(..., frame_major=True)."""
        t = self.torch
        n = int(mag_off.size)
        nbins = int(first.size)
        nframes = np.ascontiguousarray(nframes, dtype=np.int32)
        sizes = nframes.astype(np.int64) * nbins
        out_off = np.zeros(n, dtype=np.int64)
        if n > 1:
            out_off[1:] = np.cumsum(sizes[:-1])
        out = self.empty(int(sizes.sum()), t.float32)
        d_mo, d_nf, d_oo = self.to_dev(mag_off), self.to_dev(nframes), self.to_dev(out_off)
        d_f, d_c = self.to_dev(first.astype(np.int32)), self.to_dev(count.astype(np.int32))
        check(self.lib.ira_logbin_aggregate(_ptr(mag_dev), _ptr(d_mo), _ptr(d_nf), n, int(nframes.max()), int(k_base),
                                            _ptr(d_f), _ptr(d_c), nbins, _ptr(out), _ptr(d_oo), int(frame_major_rows),
                                            self.stream), "ira_logbin_aggregate")
        return out, out_off

    # ------------------------------------------------------------------ a19-a21: AR pole fit
    def ar_fit(self, x_dev, xoff: np.ndarray, lengths: np.ndarray, divisor: Optional[np.ndarray], order: int,
               ridge: float = 0.0, x_is_f64: bool = False):
        """AR coefficients (nb, order+1) float64 device + info (nb, 4: status, pivots, cond estimate) for segments of x_dev."""
        t = self.torch
        n = int(xoff.size)
        lengths = np.ascontiguousarray(lengths, dtype=np.int32)
        max_len = int(lengths.max())
        per = int(self.lib.ira_ar_partial_doubles(int(order), max_len))
        if per <= 0:
            raise ValueError("AR order must satisfy 1 <= order <= 1024 < segment length")
        part = self.empty(n * per, t.float64)
        gs = self.empty(n * order * order, t.float64) if order > 128 else None
        coeffs = self.empty(n * (order + 1), t.float64)
        info = self.empty(n * 4, t.float64)
        d_xo, d_l, d_div = self.job_tables(np.ascontiguousarray(xoff, np.int64), lengths,
                                            np.ascontiguousarray(divisor, np.float64) if divisor is not None else None)
        flags = (1 if self.ar_dense_gram else 0) | (2 if self.ar_workgroup_solve else 0)   # IRA_AR_DENSE_GRAM | IRA_AR_WORKGROUP_SOLVE
        check(self.lib.ira_ar_gram(0 if x_is_f64 else _ptr(x_dev), _ptr(x_dev) if x_is_f64 else 0, _ptr(d_xo),
                                   _ptr(d_l), _ptr(d_div), n, max_len, int(order), _ptr(part), flags, self.stream),
              "ira_ar_gram")
        check(self.lib.ira_ar_solve(_ptr(part), _ptr(d_l), n, max_len, int(order), float(ridge), _ptr(gs),
                                    _ptr(coeffs), _ptr(info), flags, self.stream), "ira_ar_solve")
        if self.ar_exact_cond > 0.0 and not self.ar_dense_gram:
            # conditional on the device: only elements whose float64 Cholesky broke down or whose condition estimate says
            # cond(G) eps is no longer small are solved again, in double-double arithmetic (ira_ar_exact)
            ddp = self.empty(n * int(self.lib.ira_ar_exact_doubles(int(order), max_len, 0)), t.float64)
            dds = self.empty(n * int(self.lib.ira_ar_exact_doubles(int(order), max_len, 1)), t.float64)
            check(self.lib.ira_ar_exact(0 if x_is_f64 else _ptr(x_dev), _ptr(x_dev) if x_is_f64 else 0, _ptr(d_xo),
                                        _ptr(d_l), _ptr(d_div), n, max_len, int(order), float(ridge), _ptr(part), _ptr(ddp),
                                        _ptr(dds), _ptr(coeffs), _ptr(info), float(self.ar_exact_cond), self.stream),
                  "ira_ar_exact")
        if ridge == 0.0 and order <= 512 and self.ar_minnorm_cut > 0.0 and not self.ar_dense_gram:
            # conditional on the device: only elements whose Cholesky failed (rank-deficient Gram matrix) do any work
            scratch2 = self.empty(n * 2 * order * order, t.float64)
            check(self.lib.ira_ar_minnorm(_ptr(part), _ptr(d_l), n, max_len, int(order), _ptr(scratch2), _ptr(coeffs),
                                          _ptr(info), float(self.ar_minnorm_cut), self.stream), "ira_ar_minnorm")
        if ridge == 0.0 and self.ar_refine_steps > 0:
            # conditional on the device: only elements whose pivots show cond(G) > threshold do any work
            nchunks = -(-(max_len - int(order)) // 4096)
            grad = self.empty(n * nchunks * (order + 1), t.float64)
            check(self.lib.ira_ar_refine(0 if x_is_f64 else _ptr(x_dev), _ptr(x_dev) if x_is_f64 else 0, _ptr(d_xo),
                                         _ptr(d_l), _ptr(d_div), n, max_len, int(order), _ptr(part), _ptr(gs),
                                         _ptr(coeffs), _ptr(info), _ptr(grad), float(self.ar_refine_cond),
                                         int(self.ar_refine_steps), flags, self.stream), "ira_ar_refine")
        return coeffs[: n * (order + 1)].view(n, order + 1), info[: n * 4].view(n, 4)

    # Normal equations lose cond(A)^2 eps; above this cond(G) estimate the fit gets refinement steps (ira_ar_refine).
    # Rank-deficient Gram matrices (Cholesky pivot <= 0) get the minimum-norm solution lstsq returns (ira_ar_minnorm):
    # eigen-directions with lambda <= cut * lambda_max are dropped.  0 disables the fallback.
    ar_minnorm_cut = 1e-12
    # A/B: form the Gram matrix as a dense contraction on the FP64 matrix cores (IRA_AR_DENSE_GRAM) instead of the lag sums
    ar_dense_gram = False
    # A/B: the 256-thread solve kernel also for order <= 64 (default there since round 4: one wave per element, same bits)
    ar_workgroup_solve = False
    ar_refine_cond = 1e9          # on the estimate trace(G) ||G^-1|| (<= order * cond(G))
    ar_refine_steps = 2
    # Above this estimate (or when the float64 Cholesky breaks down) the normal equations are solved again in double-double
    # arithmetic (ira_ar_exact): refinement needs cond(G) eps << 1.  0 disables the path (A/B).
    ar_exact_cond = 1e13

    def poly_roots(self, coeffs_dev, npoly: int, ncoef: int, trail_eps: float = 1e-14):
        """Roots (npoly, ncoef-1, 2) float64 device + counts int32 device."""
        t = self.torch
        roots = self.empty(npoly * (ncoef - 1) * 2, t.float64)
        cnt = self.empty(npoly, t.int32)
        check(self.lib.ira_poly_roots(_ptr(coeffs_dev), int(npoly), int(ncoef), float(trail_eps), _ptr(roots),
                                      _ptr(cnt), self.stream), "ira_poly_roots")
        return roots[: npoly * (ncoef - 1) * 2].view(npoly, ncoef - 1, 2), cnt[:npoly]

    def fir_numerator(self, coeffs_dev, order: int, x_dev, xoff: np.ndarray, lengths: np.ndarray,
                      divisor: Optional[np.ndarray], zero_order: int):
        t = self.torch
        n = int(xoff.size)
        b = self.empty(n * (zero_order + 1), t.float64)
        d_xo = self.to_dev(np.ascontiguousarray(xoff, np.int64))
        d_l = self.to_dev(np.ascontiguousarray(lengths, np.int32))
        d_div = self.to_dev(np.ascontiguousarray(divisor, np.float64)) if divisor is not None else None
        check(self.lib.ira_fir_numerator(_ptr(coeffs_dev), int(order), _ptr(x_dev), _ptr(d_xo), _ptr(d_l), _ptr(d_div),
                                         n, int(zero_order), _ptr(b), self.stream), "ira_fir_numerator")
        return b[: n * (zero_order + 1)].view(n, zero_order + 1)

    # ------------------------------------------------------------------ section 8f rank 4: deconvolution
    def deconv_divide(self, yspec_dev, yspec_off: np.ndarray, xspec_dev, xspec_off: np.ndarray, n_fft: np.ndarray,
                      regularization_relative: float) -> None:
        """Y <- Y conj(X) / (|X|^2 + eps) in place (see ira_deconv_divide)."""
        t = self.torch
        n = int(yspec_off.size)
        n_fft = np.ascontiguousarray(n_fft, dtype=np.int32)
        pmax = self.empty(n, t.float64)
        # device copies stay referenced until the launch is enqueued (a temporary would hand its block back to the
        # allocator, and the next to_dev would overwrite it before the kernel reads it)
        d_yo, d_xo = self.to_dev(np.ascontiguousarray(yspec_off, np.int64)), self.to_dev(np.ascontiguousarray(xspec_off, np.int64))
        d_nf = self.to_dev(n_fft)
        check(self.lib.ira_deconv_divide(_ptr(yspec_dev), _ptr(d_yo), _ptr(xspec_dev), _ptr(d_xo), _ptr(d_nf), n,
                                         int(n_fft.max()), float(regularization_relative), _ptr(pmax), self.stream),
              "ira_deconv_divide")

    def deconv_finish(self, h_dev, h_off: np.ndarray, n_out: np.ndarray, group: np.ndarray, remove_dc: bool,
                      normalise_peak: bool, target_peak: float) -> None:
        """DC removal per channel and one peak normalisation per file, in place (see ira_deconv_finish)."""
        t = self.torch
        n = int(h_off.size)
        n_out = np.ascontiguousarray(n_out, dtype=np.int32)
        group = np.ascontiguousarray(group, dtype=np.int32)
        ngroups = int(group.max()) + 1 if n else 0
        mean, peak = self.empty(n, t.float32), self.empty(ngroups, t.int32)
        d_ho, d_no, d_gr = self.to_dev(np.ascontiguousarray(h_off, np.int64)), self.to_dev(n_out), self.to_dev(group)
        check(self.lib.ira_deconv_finish(_ptr(h_dev), _ptr(d_ho), _ptr(d_no), _ptr(d_gr), n, ngroups,
                                         int(n_out.max()) if n else 0, 1 if remove_dc else 0, 1 if normalise_peak else 0,
                                         float(target_peak), _ptr(mean), _ptr(peak), self.stream), "ira_deconv_finish")

    def segment_peaks(self, x_dev, off: np.ndarray, lens: np.ndarray):
        """max|x| (float32 values as float64) of arbitrary segments."""
        t = self.torch
        n = int(off.size)
        pk = self.empty(n, t.int64)
        pa = self.empty(n, t.float32)
        d_o, d_l = self.to_dev(np.ascontiguousarray(off, np.int64)), self.to_dev(np.ascontiguousarray(lens, np.int64))
        check(self.lib.ira_peak_index(_ptr(x_dev), _ptr(d_o), _ptr(d_l), n, int(np.max(lens)) if n else 0, _ptr(pk),
                                      _ptr(pa), self.stream), "ira_peak_index")
        return pa.cpu().numpy()[:n].astype(np.float64)
