"""
Native tap ingest: bundle tap files -> a device-resident ChannelBatch without the
Python WAV stack.

The reference reads every tap with scipy.io.wavfile.read, converts int16 -> float32 on the host
(analyse/io.py:46-64, :98-113), applies the channel policy (analyse/io.py:66-95) and repeats all of that once per
analysis module (ten times per report).  Here a tap written by the reference's C++ recorder
(include/analysis/recorder.hpp:55-90: 44-byte header, 16-bit PCM, stereo) is

  1. probed by libira's RIFF walker (ira_wav_probe; host),
  2. read as raw interleaved int16 straight into pinned memory (ira_wav_read_pcm16; host),
  3. uploaded as int16 (2 bytes per sample on the PCIe link instead of 4), and
  4. converted on the device (ira_pcm16_to_channels): x/32768 clipped to [-1, 1], split into planar channels or
     mixed down to 0.5*(L+R) in float32 -- bit-identical to the reference's conversion.

WAV encodings the native reader does not take (int32, float, 24-bit ...) go through analyse.io.load_wav_file (file
decoding only; every analysis still runs on the device).  Validation errors are the reference's
(analyse/io.py:156-178), raised per file with the file's path.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import check
from .engine import ChannelBatch, Engine

IRA_E_UNSUPPORTED = -3
IRA_E_IO = -4
IRA_E_FORMAT = -5


def _raise_like_scipy(rc: int, path: Path) -> None:
    """Errors of the reference's reader for the same files (scipy.io.wavfile.read behind analyse/io.py:200): a missing
    or unreadable file is an OSError (FileNotFoundError when it does not exist), a file that is not RIFF/WAVE a
    ValueError.  bundle.run_bundle_report's abort semantics (reference bundle.py:56-67) depend on the types.
    A data chunk shorter than its header says is NOT an error there: scipy warns "Reached EOF prematurely" and returns
    the samples the file holds, so ira_wav_probe reports that many frames and the tap is analysed (tests/golden/
    truncated_wav.json, written by the reference's reader)."""
    import errno
    import os
    if rc == IRA_E_IO:
        if not path.exists():
            raise FileNotFoundError(errno.ENOENT, os.strerror(errno.ENOENT), str(path))
        raise OSError(errno.EIO, "could not read WAV file (truncated or unreadable)", str(path))
    if rc == IRA_E_FORMAT:
        raise ValueError(f"File format of {path} not understood. Only 'RIFF' / 'WAVE' files are supported.")


@dataclass(frozen=True)
class TapInfo:
    path: Path
    sample_rate_hz: int
    channels: int
    frames: int
    data_offset: int
    native: bool                 # True: 16-bit PCM that libira reads itself


def probe_tap(path: str | Path) -> TapInfo:
    """RIFF/WAVE header of one tap (host only; does not need a GPU)."""
    lib = _lib.load()
    p = Path(path)
    rate, ch = C.c_int32(0), C.c_int32(0)
    frames, off = C.c_int64(0), C.c_int64(0)
    rc = lib.ira_wav_probe(str(p).encode(), C.addressof(rate), C.addressof(ch), C.addressof(frames), C.addressof(off))
    if rc == IRA_E_UNSUPPORTED:
        return TapInfo(p, int(rate.value), int(ch.value), int(frames.value), int(off.value), False)
    _raise_like_scipy(rc, p)
    check(rc, f"ira_wav_probe({p})")
    return TapInfo(p, int(rate.value), int(ch.value), int(frames.value), int(off.value), True)


def read_tap_pcm16(info: TapInfo, dst: Optional[np.ndarray] = None) -> np.ndarray:
    """Interleaved int16 payload of a native tap, shape (frames, channels) (host only)."""
    if not info.native:
        raise ValueError(f"{info.path} is not 16-bit PCM")
    lib = _lib.load()
    if dst is None:
        dst = np.empty((info.frames, info.channels), dtype=np.int16)
    if dst.dtype != np.int16 or dst.size != info.frames * info.channels or not dst.flags.c_contiguous:
        raise ValueError("dst must be a C-contiguous int16 array of frames*channels values")
    rc = lib.ira_wav_read_pcm16(str(info.path).encode(), info.data_offset, info.frames, info.channels, dst.ctypes.data)
    _raise_like_scipy(rc, info.path)
    check(rc, f"ira_wav_read_pcm16({info.path})")
    return dst


def _io_threads() -> int:
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 2)
    return max(2, min(16, n))


def _path_array(paths: Sequence[Path]):
    enc = [str(p).encode() for p in paths]
    return (C.c_char_p * len(enc))(*enc), enc          # (the byte strings must outlive the call)


def probe_taps(paths: Sequence[str | Path]) -> List[TapInfo]:
    """probe_tap for a whole group in ONE library call (ira_wav_probe_batch: the readers run on threads inside the call,
    the interpreter lock is released once).  Errors are raised for the first offending file in group order, with the
    types of the per-file path."""
    ps = [Path(p) for p in paths]
    n = len(ps)
    if n == 0:
        return []
    lib = _lib.load()
    arr, keep = _path_array(ps)
    status = np.zeros(n, dtype=np.int32); rate = np.zeros(n, dtype=np.int32); ch = np.zeros(n, dtype=np.int32)
    frames = np.zeros(n, dtype=np.int64); off = np.zeros(n, dtype=np.int64)
    check(lib.ira_wav_probe_batch(C.addressof(arr), n, _io_threads(), status.ctypes.data, rate.ctypes.data, ch.ctypes.data,
                                  frames.ctypes.data, off.ctypes.data), "ira_wav_probe_batch")
    del keep
    out = []
    for i, p in enumerate(ps):
        rc = int(status[i])
        if rc == IRA_E_UNSUPPORTED:
            out.append(TapInfo(p, int(rate[i]), int(ch[i]), int(frames[i]), int(off[i]), False))
            continue
        _raise_like_scipy(rc, p)
        check(rc, f"ira_wav_probe({p})")
        out.append(TapInfo(p, int(rate[i]), int(ch[i]), int(frames[i]), int(off[i]), True))
    return out


def read_taps_pcm16(infos: Sequence[TapInfo], dst: np.ndarray, dst_off: Sequence[int]) -> None:
    """Payloads of a group of native taps into ONE int16 buffer (file i at dst[dst_off[i]:]) in one library call."""
    n = len(infos)
    if n == 0:
        return
    if dst.dtype != np.int16 or not dst.flags.c_contiguous:
        raise ValueError("dst must be a C-contiguous int16 array")
    for info, o in zip(infos, dst_off):
        if not info.native:
            raise ValueError(f"{info.path} is not 16-bit PCM")
        if o < 0 or o + info.frames * info.channels > dst.size:
            raise ValueError("dst is too small for the payloads")
    lib = _lib.load()
    arr, keep = _path_array([i.path for i in infos])
    d_off = np.array([i.data_offset for i in infos], dtype=np.int64)
    fr = np.array([i.frames for i in infos], dtype=np.int64)
    ch = np.array([i.channels for i in infos], dtype=np.int32)
    do = np.array(list(dst_off), dtype=np.int64)
    status = np.zeros(n, dtype=np.int32)
    check(lib.ira_wav_read_pcm16_batch(C.addressof(arr), d_off.ctypes.data, fr.ctypes.data, ch.ctypes.data, do.ctypes.data,
                                       dst.ctypes.data, n, _io_threads(), status.ctypes.data), "ira_wav_read_pcm16_batch")
    del keep
    for i, info in enumerate(infos):
        rc = int(status[i])
        _raise_like_scipy(rc, info.path)
        check(rc, f"ira_wav_read_pcm16({info.path})")


def _validate(info: TapInfo, expected_rate: int) -> None:
    # messages of the reference's validate_audio_format (analyse/io.py:161-178), channel mode "mono_or_stereo"
    if info.sample_rate_hz != expected_rate:
        raise ValueError(f"Expected sample rate {expected_rate} Hz, but got {info.sample_rate_hz} Hz "
                         f"for file {info.path}")
    if info.channels not in (1, 2):
        raise ValueError(f"Expected mono or stereo (1 or 2 channels) but got {info.channels} channels "
                         f"for file {info.path}")


def channel_names(channels: int, mono_downmix: bool) -> List[str]:
    """Channel policy of analyse/io.py:66-95."""
    if channels == 1 or mono_downmix:
        return ["mono"]
    return ["left", "right"]


class TapSet:
    """
    A set of tap files resident on the device: ONE upload (int16 for native taps), any number of channel-policy views.

    view(mono_downmix) -> (ChannelBatch, [(file index, channel name), ...]) converts on the device
    (ira_pcm16_to_channels) the first time a policy is asked for and caches the batch, so a report whose blocks
    disagree about the policy (the reference's rt60bands / group delay / diffusion blocks keep their own,
    report.py:172-186) reads and uploads every file once.
    """

    def __init__(self, eng: Engine, paths: Sequence[str | Path], expected_sample_rate_hz: int = 48_000,
                 upload: bool = True):
        """upload=False stops after the HOST half (headers probed, payloads read into pinned staging): it touches no
        stream, so it can run on a worker thread while the caller's thread drives the GPU; call upload() on the
        caller's thread before the first view()."""
        t = eng.torch
        self.eng = eng
        self.infos = probe_taps(paths)
        self.expected_sample_rate_hz = int(expected_sample_rate_hz)
        for info in self.infos:
            _validate(info, self.expected_sample_rate_hz)
        # ---- native files: raw int16 into pinned memory (4-byte aligned per file), one upload ---------------------
        self._native = [i for i, info in enumerate(self.infos) if info.native and info.frames > 0]
        self._pcm_off = {}
        total = 0
        for i in self._native:
            self._pcm_off[i] = total
            total += (self.infos[i].frames * self.infos[i].channels + 1) & ~1   # stereo frames stay 4-byte aligned
        self._pcm_dev = None
        if total:
            # pinned staging straight from torch's caching host allocator (a block of an earlier step is reused, nothing
            # is pinned anew); the files are read concurrently inside ONE library call (ira_wav_read_pcm16_batch)
            stage = t.empty(total, dtype=t.int16, pin_memory=True)
            read_taps_pcm16([self.infos[i] for i in self._native], stage.numpy(), [self._pcm_off[i] for i in self._native])
            self._stage = stage
            if upload:
                self.upload()
        # ---- other encodings: decoded by the Python reader when first needed -------------------------------------
        self._loaded = {}
        self._views = {}

    def upload(self) -> None:
        """Enqueue the one host-to-device copy of the staged payloads on the caller's current stream (idempotent)."""
        stage = getattr(self, "_stage", None)
        if stage is not None and self._pcm_dev is None:
            self._pcm_dev = stage.to(self.eng.device, non_blocking=True)
        self._stage = None

    def __len__(self) -> int:
        return len(self.infos)

    def view(self, use_mono_downmix_for_stereo: bool = False) -> Tuple[ChannelBatch, List[Tuple[int, str]]]:
        mono = bool(use_mono_downmix_for_stereo)
        if mono in self._views:
            return self._views[mono]
        self.upload()
        eng, t, infos = self.eng, self.eng.torch, self.infos
        labels: List[Tuple[int, str]] = []
        lens: List[int] = []
        for i, info in enumerate(infos):
            for name in channel_names(info.channels, mono):
                labels.append((i, name))
                lens.append(info.frames)
        lens_a = np.asarray(lens, dtype=np.int64)
        off = np.zeros(len(lens), dtype=np.int64)
        if len(lens) > 1:
            off[1:] = np.cumsum(lens_a[:-1])
        x = eng.empty(int(lens_a.sum()), t.float32)
        first_channel = {}
        for k, (i, _) in enumerate(labels):
            first_channel.setdefault(i, k)
        if self._native:
            # ONE conversion launch for the whole group: a job table instead of a launch (and an event pair) per tap
            nat = self._native
            src_off = np.array([self._pcm_off[i] for i in nat], dtype=np.int64)
            frames = np.array([infos[i].frames for i in nat], dtype=np.int64)
            chans = np.array([infos[i].channels for i in nat], dtype=np.int32)
            modes = np.array([1 if (mono and infos[i].channels == 2) else 0 for i in nat], dtype=np.int32)
            dst_off = np.array([int(off[first_channel[i]]) for i in nat], dtype=np.int64)
            d_src, d_fr, d_ch, d_mo, d_dst = eng.job_tables(src_off, frames, chans, modes, dst_off)
            check(eng.lib.ira_pcm16_to_channels_jobs(int(self._pcm_dev.data_ptr()), int(d_src.data_ptr()),
                                                     int(d_fr.data_ptr()), int(d_ch.data_ptr()), int(d_mo.data_ptr()),
                                                     int(d_dst.data_ptr()), len(nat), int(frames.max()),
                                                     int(x.data_ptr()), eng.stream), "ira_pcm16_to_channels_jobs")
        if self._pcm_dev is not None:
            self._pcm_dev.record_stream(t.cuda.current_stream(eng.device))
        for i, info in enumerate(infos):
            if info.native or info.frames == 0:
                continue
            from .analyse.io import get_analysis_channels, load_wav_file
            if i not in self._loaded:
                self._loaded[i] = load_wav_file(info.path, self.expected_sample_rate_hz, "mono_or_stereo", False)
            for j, (_, c) in enumerate(get_analysis_channels(self._loaded[i], mono)):
                o = int(off[first_channel[i] + j])
                x[o : o + c.size].copy_(t.from_numpy(np.ascontiguousarray(c, dtype=np.float32)))
        self._views[mono] = (eng.wrap(x, off, lens_a), labels)
        return self._views[mono]


    def mix_peaks(self, files: Sequence[int]) -> np.ndarray:
        """argmax |0.5*(L+R)| (first maximum) of the given STEREO files -- the alignment point of the reference's stereo
        diffusion metrics (diffusion.py:326-335).  Native taps: peak pick of the device's mono-downmix view (for int16
        data the float32 mean equals the reference's float64 mean rounded once).  Decoded float files: the reference's
        own expression on the host copy, so that double rounding cannot move a tie."""
        out = np.zeros(len(files), dtype=np.int64)
        batch, _ = self.view(True)
        peaks = None
        for j, f in enumerate(files):
            if self.infos[f].channels != 2:
                raise ValueError("mix_peaks is defined for stereo files")
            if f in self._loaded:
                smp = self._loaded[f].samples
                comb = ((smp[:, 0].astype(np.float64) + smp[:, 1].astype(np.float64)) * 0.5).astype(np.float32)
                out[j] = int(np.argmax(np.abs(comb))) if comb.size else 0
            else:
                if peaks is None:
                    peaks = self.eng.peaks(batch)
                out[j] = int(peaks[f])                        # the downmix view holds exactly one channel per file
        return out


def ingest_taps(eng: Engine, paths: Sequence[str | Path], use_mono_downmix_for_stereo: bool = False,
                expected_sample_rate_hz: int = 48_000) -> Tuple[ChannelBatch, List[Tuple[int, str]]]:
    """
    Tap files -> (device batch of analysis channels, [(file index, channel name), ...] in batch order).

    All native taps share ONE pinned int16 staging buffer and ONE host-to-device copy; each file then gets one
    conversion launch that writes its planar channels at their place in the flat float32 batch buffer.
    """
    return TapSet(eng, paths, expected_sample_rate_hz).view(use_mono_downmix_for_stereo)
