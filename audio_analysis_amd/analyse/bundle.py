"""
Bundle runner: meta.json + taps/<name>.wav -> reports/<name>/<name>_report.md + reports/bundle_report.md.

Host-side mirror of the reference's analyse/bundle.py (:29-74).  Bundle layout is the one the reference's C++
recorder writes (include/analysis/recorder.hpp:102-126).  Taps are independent files, so with
torch.distributed initialised (one process per GPU) each rank analyses a contiguous block of taps
(audio_analysis_amd.dist.shard_files) and rank 0 writes the index; single process = the reference's loop.
"""
from __future__ import annotations

import json
import os
import sys
import time
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional

from .. import dist as _dist
from .report import ReportSettings

LAST_HOST_MS_PER_GROUP = None     # IRA_BUNDLE_TIMING=1: host milliseconds per group and phase of the last run_bundle_metrics call


@dataclass(frozen=True)
class BundleRunSettings:
    reports_subdir: str = "reports"
    report_settings: Optional[ReportSettings] = None
    # extra fields (defaults reproduce the reference's outputs): taps analysed per device batch, and the number of CPU
    # worker processes that render PNGs off the critical path (0 = render inline like the reference)
    taps_per_batch: int = 16
    plot_workers: int = 0


def run_bundle_report(bundle_root: str | Path, settings: Optional[BundleRunSettings] = None) -> Path:
    """
    reference bundle.py:35-74.  Taps are analysed `taps_per_batch` files at a time through report.run_reports_batched
    (every block runs once over the channels of the whole group; per-tap Markdown string-identical to the per-file
    path).  Error semantics are the reference's: the first tap that cannot be analysed raises, taps before it have
    their reports written -- a failing group is re-run tap by tap to find it.
    """
    from .report import PlotPool, run_reports_batched

    settings = settings or BundleRunSettings()
    root = Path(bundle_root)
    meta = json.loads((root / "meta.json").read_text())
    taps: List[str] = list(meta.get("taps", []))
    reports = root / settings.reports_subdir
    reports.mkdir(parents=True, exist_ok=True)

    rank, _, world = _dist.env_world()
    lo, hi = _dist.shard_files(len(taps), rank, world)
    rs = settings.report_settings
    draw = bool(rs.render_plots) if rs is not None else True
    pool = PlotPool(settings.plot_workers) if (settings.plot_workers > 0 and draw) else None
    # Abort semantics of the reference's serial loop (bundle.py:56-67): the first tap that cannot be analysed raises --
    # whatever the exception type (ValueError from validation, FileNotFoundError / OSError from a missing or truncated
    # tap) -- and every tap before it has its report on disk.  A failing group is therefore re-run tap by tap.  With
    # several ranks, a rank that fails still joins the barrier (the others would otherwise wait for the collective's
    # time-out) and re-raises afterwards; the index is not written then.
    failure: Optional[BaseException] = None
    try:
        step = max(1, int(settings.taps_per_batch))
        for a in range(lo, hi, step):
            group = taps[a : min(hi, a + step)]
            items = [(root / "taps" / f"{tap}.wav", reports / tap / tap) for tap in group]
            try:
                run_reports_batched(items, rs, plot_pool=pool)
            except Exception:
                if len(items) == 1:
                    raise
                for item in items:                              # find the offending tap the way the reference would
                    run_reports_batched([item], rs, plot_pool=pool)
                raise
    except Exception as exc:                                    # noqa: BLE001 -- re-raised below, after the barrier
        failure = exc
    finally:
        if pool is not None:
            pool.close()
    failed_somewhere = _dist.any_rank_true(failure is not None)
    if failure is not None:
        raise failure
    if failed_somewhere:
        raise RuntimeError("bundle report aborted: another rank failed on one of its taps")

    index = reports / "bundle_report.md"
    if rank == 0:
        lines = [
            "# IR Bundle Report\n",
            f"**Bundle:** `{root}`\n",
            f"**Sample rate:** {meta.get('sample_rate_hz')}\n",
            f"**Length (samples):** {meta.get('length_samples')}\n",
            "\n## Taps\n",
        ]
        lines += [f"- [{tap}]({settings.reports_subdir}/{tap}/{tap}_report.md)" for tap in taps]
        index.write_text("\n".join(lines) + "\n")
    return index


def run_bundle_metrics(bundle_root: str | Path, settings=None, use_mono_downmix_for_stereo: bool = False,
                       taps_per_step: int = 32, rank_world=None, gather: bool = True):
    """
    Batched metrics-only pass over a bundle (SURVEY.md section 8f rank 2 + section 8e): every rank ingests its block
    of taps natively (audio_analysis_amd.ingest: int16 upload, conversion on the device), runs the metrics-only full
    report (audio_analysis_amd.pipeline.FullReport) `taps_per_step` files at a time, and ONE gather brings the
    fixed-width records to rank 0.  Returns ([(tap name, channel name), ...], records (channels x METRICS_WIDTH)) on
    rank 0 and (None, None) elsewhere.  No PNGs, no Markdown: this is the data-parallel axis of the reference's
    serial loop (bundle.py:56-67) without its per-file plotting.
    """
    import numpy as np

    from ..engine import get_engine
    from ..ingest import TapSet
    from ..pipeline import METRICS_WIDTH, FullReport, FullReportSettings

    root = Path(bundle_root)
    meta = json.loads((root / "meta.json").read_text())
    taps: List[str] = list(meta.get("taps", []))
    rank, _, world = _dist.env_world()
    if rank_world is not None:                       # caller-defined shard (rank, world) instead of the launcher's
        rank, world = int(rank_world[0]), int(rank_world[1])
    # Uniform bundles (what the recorder writes: every tap has meta.length_samples frames) shard as contiguous blocks of
    # files; RAGGED bundles are sorted by size and dealt round-robin so that every rank gets the same share of samples
    # (SURVEY.md section 8e).  Both channels of a file stay on one rank either way; rank 0 restores bundle order.
    def _size(name):
        try:
            return (root / "taps" / f"{name}.wav").stat().st_size
        except OSError:
            return 0                                 # a missing tap raises in the ingest of the rank that owns it

    sizes = [_size(t) for t in taps] if world > 1 else []
    ragged = world > 1 and len(set(sizes)) > 1
    if ragged:
        mine = [taps[i] for i in _dist.balanced_assignment(sizes, world)[rank]]
    else:
        lo, hi = _dist.shard_files(len(taps), rank, world)
        mine = taps[lo:hi]
    eng = get_engine()
    fr = FullReport(eng, settings or FullReportSettings())
    rows, labels = [], []
    step = max(1, int(taps_per_step))
    rate = int(meta.get("sample_rate_hz", 48_000) or 48_000)
    groups = [mine[a : a + step] for a in range(0, len(mine), step)]

    # diagnostics (IRA_BUNDLE_TIMING=1): host seconds per phase of the loop; nothing is timed when the switch is off
    timing = {} if os.environ.get("IRA_BUNDLE_TIMING") else None

    def lap(key, t0):
        if timing is None:
            return 0.0
        now = time.perf_counter()
        timing[key] = timing.get(key, 0.0) + (now - t0)
        return now

    def host_half(names):                            # headers + payload reads into pinned staging: no GPU call in here
        t0 = time.perf_counter() if timing is not None else 0.0
        ts = TapSet(eng, [root / "taps" / f"{t}.wav" for t in names], rate, upload=False)
        if timing is not None:                       # (the reader thread's own total; dict updates are atomic under the GIL)
            timing["reader thread busy"] = timing.get("reader thread busy", 0.0) + (time.perf_counter() - t0)
        return ts

    # Three groups in flight.  While this thread works on groups k, k-1 and k-2, a worker thread reads group k+1 from
    # disk (the readers inside libira release the GIL); every GPU call stays on this thread:
    #   group k    upload + conversion enqueued, peak pick started behind them (not waited for)
    #   group k-1  its peaks have arrived meanwhile: all report kernels are enqueued (submit)
    #   group k-2  its records are read back (finish)
    # so the GPU always holds one enqueued step while the host finishes another, and uploads overlap compute.
    from concurrent.futures import ThreadPoolExecutor
    uploaded = None                                  # group k-1: batch waiting for its peaks
    pending = []                                     # group k-2 (and older): submitted steps whose records are on their way
    # (IRA_BUNDLE_DEPTH submitted steps in flight before the oldest is read back: 2 or 3 were measured against 1 at the end
    # of round 4 -- no gain, the loop is bound by the host's ~10 ms of work per 256 channels either way)
    depth = max(1, int(os.environ.get("IRA_BUNDLE_DEPTH", "1")))
    # The reader thread runs `ahead_n` groups ahead (default 2; IRA_BUNDLE_PREFETCH is the A/B switch).  With one group ahead
    # the reader only starts on group k+1 when the main thread picks up group k, and then has to win the interpreter lock
    # back from a main thread that spends the next milliseconds in pure Python (submit).  Measured, config 5 on one GPU,
    # alternating (profiles/r05_bundle_ab.txt): one group ahead 9.6 k stereo taps/s, two 10.8 k, three 10.0-10.2 k.  A shorter
    # interpreter switch interval for the duration of the loop (IRA_BUNDLE_SWITCH_US, default off) was measured too:
    # 0.5 ms 9.8-10.7 k, 0.1 ms 9.2-9.3 k -- not adopted.
    ahead_n = max(1, int(os.environ.get("IRA_BUNDLE_PREFETCH", "2")))
    switch_before = sys.getswitchinterval()
    switch_us = float(os.environ.get("IRA_BUNDLE_SWITCH_US", "0"))
    if switch_us > 0:
        sys.setswitchinterval(switch_us * 1e-6)
    with ThreadPoolExecutor(max_workers=1, thread_name_prefix="ira-prefetch") as ahead:
        queue = [ahead.submit(host_half, groups[i]) for i in range(min(ahead_n, len(groups)))]
        for gi, names in enumerate(groups):
            t0 = time.perf_counter() if timing is not None else 0.0
            tapset = queue.pop(0).result()
            t0 = lap("wait for the reader thread", t0)
            if gi + ahead_n < len(groups):
                queue.append(ahead.submit(host_half, groups[gi + ahead_n]))
            batch, lab = tapset.view(use_mono_downmix_for_stereo)
            labels += [(names[i], ch) for i, ch in lab]
            t0 = lap("view (upload + conversion enqueued)", t0)
            fr.prepare(batch)
            t0 = lap("prepare (peak pick started)", t0)
            if uploaded is not None:
                pending.append(fr.submit(uploaded))
                t0 = lap("submit", t0)
                if len(pending) > depth:
                    rows.append(fr.finish(pending.pop(0)))
                    t0 = lap("finish", t0)
            uploaded = batch
        sys.setswitchinterval(switch_before)
        if timing is not None and groups:
            global LAST_HOST_MS_PER_GROUP            # for callers that report it (bench.py, config 5)
            LAST_HOST_MS_PER_GROUP = {k: 1e3 * v / len(groups) for k, v in timing.items()}
            print("[bundle] host ms per group: " + ", ".join(f"{k} {1e3 * v / len(groups):.2f}" for k, v in timing.items()),
                  file=sys.stderr)
    if uploaded is not None:
        pending.append(fr.submit(uploaded))
    for handle in pending:
        rows.append(fr.finish(handle))
    local = np.concatenate(rows, axis=0) if rows else np.zeros((0, METRICS_WIDTH))
    if not gather:                                   # this rank's (labels, records) only; the caller gathers
        return labels, local
    records = _dist.gather_metrics(local, eng.device)
    if world > 1:
        import torch.distributed as td
        gathered = [None] * world if rank == 0 else None
        td.gather_object(labels, gathered, dst=0)
        labels = [x for part in gathered for x in part] if rank == 0 else None
        if rank == 0 and ragged:
            place = {name: i for i, name in enumerate(taps)}
            order = np.argsort(np.array([place[name] for name, _ in labels], dtype=np.int64), kind="stable")
            labels = [labels[i] for i in order]
            records = records[order]
    return (labels, records) if rank == 0 else (None, None)
