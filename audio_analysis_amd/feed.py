"""
Host -> device feed for batched runs: channel batches that live in PINNED host memory are uploaded on a dedicated
copy stream into a small ring of device buffers, one batch ahead of the step that analyses them, so that the PCIe
transfer of batch k+1 runs under the kernels of batch k (SURVEY.md section 8d puts the host-to-device copy inside the
unit of work: the reference re-loads the file in every report block, report.py:222-398 -> io.py:181-221).

Two wire formats:
  float32  4 bytes per sample, copied straight into the batch buffer;
  int16    2 bytes per sample (what a PCM16 tap holds, recorder.hpp:49-53), converted to float32 (x/32768 clipped:
           io.py:46-64) on the way.
The transfer is an ordinary asynchronous copy (hipMemcpyAsync on the copy stream; PCM16 then takes one conversion launch,
ira_pcm16_to_channels).  DeviceFeed(pull=True) moves the batch with a KERNEL instead (ira_host_pull: a few workgroups
read the pinned batch through the PCIe link and write HBM, converting PCM16 in the same pass, no int16 staging buffer in
HBM).  Measured in alternating order inside one process (tools/upload_ab.py, full report, MI355X): 256 x 10 s per step
copy engine 9.4-10.8 k, pull kernel (8 workgroups) 10.36-10.39 k IRs/s; 64 x 10 s 9.5-9.7 k vs 9.0-9.5 k -- the same
within run-to-run noise, so the copy engine stays the default.  What does NOT work is a large pull grid: 32-48 workgroups
keep so many PCIe reads in flight that the fabric queues the analysis kernels' HBM reads share back up (the peak pick
went from 0.09 to 1.2 ms, the step from 6.9 to 8.4 ms).

A float32 / int16 batch crosses as `copy_streams` pieces on as many streams (default 2: one copy of 492 MB ran at 9.6 k or
11.8 k IRs/s depending on the box, two halves at 11.9-12.1 k on all of them; four pieces 11.5 k); batches below 192 MB go as
one copy (the cross-stream events cost more than they buy: 256 x 2 s IRs per step ran at 96 k instead of 137 k IRs/s split).

A ChannelBatch handed out by push() carries the event recorded behind its upload (+ conversion): the peak pick and
every report lane wait for THAT event only (pipeline.FullReport.submit), never for the copy stream as a whole.

Ring discipline (host-ordered, no device-side fences needed): with `depth` device buffers the caller may have at most
depth-1 batches pushed and not yet finished -- push() of batch k+depth-1 ... reuses the buffer of batch k-1, whose step
the caller has already finished (FullReport.finish waits for every lane of that step).  run_pipelined() below keeps to
that: push(k+1), submit(k), finish(k-1).
"""
from __future__ import annotations

from typing import Callable, Iterable, Iterator, Optional, Sequence, Tuple

import numpy as np

from ._lib import check
from .engine import ChannelBatch, Engine


class HostBatch:
    """One batch of equal- or ragged-length mono channels in pinned host memory (flat, channel after channel)."""

    def __init__(self, eng: Engine, channels: Sequence[np.ndarray] | np.ndarray, pcm16: bool = False):
        t = eng.torch
        if isinstance(channels, np.ndarray) and channels.ndim == 2:
            lens = np.full(channels.shape[0], channels.shape[1], dtype=np.int64)
            flat = channels.reshape(-1)
        else:
            lens = np.array([int(c.size) for c in channels], dtype=np.int64)
            flat = np.concatenate([np.asarray(c).reshape(-1) for c in channels]) if len(lens) else np.zeros(0, np.float32)
        self.length = lens
        self.off = np.zeros(lens.size, dtype=np.int64)
        if lens.size > 1:
            self.off[1:] = np.cumsum(lens[:-1])
        self.total = int(lens.sum())
        self.pcm16 = bool(pcm16)
        if self.pcm16:
            if flat.dtype != np.int16:
                raise ValueError("pcm16 host batches hold int16 samples")
            self.pinned = t.empty(max(self.total, 1), dtype=t.int16, pin_memory=True)
        else:
            flat = flat.astype(np.float32, copy=False)
            self.pinned = t.empty(max(self.total, 1), dtype=t.float32, pin_memory=True)
        self.pinned.numpy()[: self.total] = flat

    @property
    def nbytes(self) -> int:
        return self.total * (2 if self.pcm16 else 4)


class DeviceFeed:
    def __init__(self, eng: Engine, max_samples: int, depth: int = 4, pull: bool = False, pull_workgroups: int = 8,
                 copy_streams: int = 2, high_priority: bool = False):
        t = eng.torch
        self.eng = eng
        prio = -1 if high_priority else 0
        # copy_streams > 1: a batch crosses PCIe as that many pieces on that many streams (copy engines) at once; the
        # batch's ready event waits for all of them
        self._prio = prio
        self._all_side = [t.cuda.Stream(device=eng.device, priority=prio) for _ in range(max(0, int(copy_streams) - 1))]
        self.side_streams = list(self._all_side)
        # timing: None, or a list that push() appends (bytes, [(start event, end event) per piece]) to -- HIP events on the
        # copy streams themselves, so that a bench line can state how long a step's upload took UNDER the kernels of the
        # step before it, and what the link does alone (upload_times())
        self.timing = None
        self.split_bytes = 192 << 20                # smaller batches go as one copy (256 x 2 s = 98 MB: 137 k vs 96 k IRs/s)
        self.pull = bool(pull)                     # True: the batch crosses PCIe under ira_host_pull instead of hipMemcpyAsync
        self.pull_workgroups = int(pull_workgroups)
        self.depth = int(depth)
        self.copy_stream = t.cuda.Stream(device=eng.device, priority=prio)
        self._x = [eng.empty(max_samples, t.float32) for _ in range(self.depth)]
        self._pcm = None
        self._max = int(max_samples)
        self._k = 0

    def push(self, hb: HostBatch) -> ChannelBatch:
        """Enqueue the upload (and int16 conversion) of a host batch on the copy stream; returns at once."""
        eng, t = self.eng, self.eng.torch
        if hb.total > self._max:
            raise ValueError("host batch larger than the feed's device buffers")
        slot = self._k % self.depth
        self._k += 1
        x = self._x[slot]
        marks = [] if self.timing is not None else None

        def timed_copy(dst, src, stream):
            """one piece: an asynchronous copy on `stream` (current), bracketed by events when the feed is being timed"""
            if marks is None:
                dst.copy_(src, non_blocking=True)
                return
            e0, e1 = t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True)
            e0.record(stream)
            dst.copy_(src, non_blocking=True)
            e1.record(stream)
            marks.append((e0, e1))

        with t.cuda.stream(self.copy_stream):
            done = False
            if self.pull:
                # pull kernel: reads the pinned batch through the PCIe link itself and converts PCM16 on the way
                # (ira_host_pull in include/ira.h)
                ev = None
                if marks is not None:
                    ev = (t.cuda.Event(enable_timing=True), t.cuda.Event(enable_timing=True))
                    ev[0].record(self.copy_stream)
                rc = eng.lib.ira_host_pull(int(hb.pinned.data_ptr()), int(hb.total), 1 if hb.pcm16 else 0,
                                           int(x.data_ptr()), int(self.pull_workgroups), eng.stream)
                if rc == -3:                                  # IRA_E_UNSUPPORTED: not mapped host memory
                    self.pull = False
                else:
                    check(rc, "ira_host_pull")
                    done = True
                    if ev is not None:
                        ev[1].record(self.copy_stream)
                        marks.append(ev)
            if done:
                pass
            elif hb.pcm16:
                if self._pcm is None:
                    self._pcm = [eng.empty(self._max, t.int16) for _ in range(self.depth)]
                pcm = self._pcm[slot]
                timed_copy(pcm[: hb.total], hb.pinned[: hb.total], self.copy_stream)
                # mono channels laid end to end convert like ONE mono file of `total` frames
                check(eng.lib.ira_pcm16_to_channels(int(pcm.data_ptr()), int(hb.total), 1, 0, int(x.data_ptr()),
                                                    eng.stream), "ira_pcm16_to_channels")
            elif self.side_streams and hb.nbytes >= self.split_bytes:
                pieces = len(self.side_streams) + 1
                step = -(-hb.total // pieces)
                for i, side in enumerate(self.side_streams):
                    lo, hi = (i + 1) * step, min(hb.total, (i + 2) * step)
                    if lo >= hi:
                        continue
                    side.wait_stream(self.copy_stream)      # ordered behind whatever used this slot before
                    with t.cuda.stream(side):
                        timed_copy(x[lo:hi], hb.pinned[lo:hi], side)
                timed_copy(x[: min(step, hb.total)], hb.pinned[: min(step, hb.total)], self.copy_stream)
                for side in self.side_streams:
                    self.copy_stream.wait_stream(side)
            else:
                timed_copy(x[: hb.total], hb.pinned[: hb.total], self.copy_stream)
            batch = eng.wrap(x, hb.off, hb.length)          # offsets/lengths + the ready event, all on the copy stream
        if marks:
            self.timing.append((hb.nbytes, marks))
        return batch

    def set_mode(self, copy_streams: Optional[int] = None, pull: Optional[bool] = None) -> None:
        """Change how the next batches cross the link: pieces per batch (copy engines), or the pull kernel."""
        t = self.eng.torch
        if pull is not None:
            self.pull = bool(pull)
        if copy_streams is not None:
            want = max(0, int(copy_streams) - 1)
            while len(self._all_side) < want:
                self._all_side.append(t.cuda.Stream(device=self.eng.device, priority=self._prio))
            self.side_streams = self._all_side[:want]

    def mode(self) -> str:
        return "pull kernel" if self.pull else f"copy engine, {len(self.side_streams) + 1} piece(s)"

    def autotune(self, run_steps: Callable[[int], float], alone_GBps: Optional[float], steps: int = 4,
                 accept: float = 0.75, first_pieces: int = 2):
        """
        Pick the upload method on THIS box, under THIS job's kernels (untimed warm-up work, like a transform plan).  Why: the
        same build uploads a 492 MB batch at 55 GB/s under the report's kernels on one MI355X box and at 26 GB/s on another
        (and at 26 GB/s on the first one as four pieces): whether an asynchronous copy rides a DMA engine or falls back to a
        copy kernel that queues behind the analysis kernels is the runtime's choice per stream, and it differs between hosts
        (profiles/r05_upload_ab.txt).  run_steps(n) must run n pipelined steps through this feed and return their wall
        seconds.  Arms, in order: `first_pieces` pieces (two: the default), one piece, the pull kernel.  The first copy-engine arm whose
        rate UNDER the kernels reaches `accept` x the link's rate alone is kept without trying the rest; otherwise the arm
        with the shortest step wins.  Returns the list of arms tried (mode, ms per step, upload ms, GB/s) for the bench line.
        """
        tried = []
        arms = [(max(1, int(first_pieces)), False), (1, False), (2, True)]
        if arms[0] == arms[1]:
            arms.pop(1)
        for streams, pull in arms:
            self.set_mode(copy_streams=streams, pull=pull)
            run_steps(1)
            self.eng.sync()
            self.timing = []
            dt = run_steps(steps)
            self.eng.sync()
            up = self.upload_times()
            self.timing = None
            tried.append({"mode": self.mode(), "ms_per_step": 1e3 * dt / steps,
                          "upload_ms": None if up is None else up["ms_per_upload"],
                          "upload_GBps": None if up is None else up["GBps"], "_set": (streams, pull)})
            if up is not None and alone_GBps and up["GBps"] >= accept * alone_GBps:
                break
        best = min(tried, key=lambda r: r["ms_per_step"])
        self.set_mode(copy_streams=best["_set"][0], pull=best["_set"][1])
        for r in tried:
            r["chosen"] = r is best
            r.pop("_set")
        return tried

    def upload_times(self):
        """Reduce and clear the timing list (call after a device synchronisation): per upload the time from the start of
        its first piece to the end of its last (ms) and its bytes; also the span from the first upload's start to the last
        upload's end, i.e. how much of that stretch the copy engines were busy."""
        rec, self.timing = self.timing or [], []
        if not rec:
            return None
        first = rec[0][1][0][0]
        spans, nbytes, t_end = [], [], 0.0
        for b, marks in rec:
            starts = [first.elapsed_time(e0) for e0, _ in marks]
            ends = [first.elapsed_time(e1) for _, e1 in marks]
            spans.append(max(ends) - min(starts))
            nbytes.append(b)
            t_end = max(t_end, max(ends))
        tot_ms, tot_b = float(sum(spans)), float(sum(nbytes))
        return {"uploads": len(rec), "pieces_per_upload": len(rec[0][1]), "bytes_per_upload": tot_b / len(rec),
                "ms_per_upload": tot_ms / len(rec), "ms_per_upload_max": float(max(spans)),
                "GBps": tot_b / tot_ms / 1e6 if tot_ms > 0 else None,
                "busy_fraction_of_stretch": tot_ms / t_end if t_end > 0 else None}


def run_pipelined(report, feed: DeviceFeed, host_batches: Iterable[HostBatch],
                  on_records: Optional[Callable[[np.ndarray], None]] = None) -> int:
    """
    Software pipeline over batches: upload k+1 | kernels of k | read-back of k-1.  Every batch that enters is
    uploaded, analysed (FullReport.submit) and finished (records handed to on_records) before this returns.
    Returns the number of batches processed.
    """
    it: Iterator[HostBatch] = iter(host_batches)
    count = 0
    pending = None
    try:
        nxt = feed.push(next(it))
    except StopIteration:
        return 0
    while nxt is not None:
        cur = nxt
        try:
            nxt = feed.push(next(it))                      # batch k+1 starts crossing PCIe now
        except StopIteration:
            nxt = None
        handle = report.submit(cur)                        # waits for batch k's upload + peak pick only
        if pending is not None:
            rec = report.finish(pending)
            if on_records is not None:
                on_records(rec)
        pending = handle
        count += 1
    if pending is not None:
        rec = report.finish(pending)
        if on_records is not None:
            on_records(rec)
    return count
