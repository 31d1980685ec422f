"""
Batched, metrics-only "full report" over a device-resident batch of channels.

One step = every enabled block of the reference's `report` (decay, rt60bands, frequency response,
spectrogram, waterfall, modal cloud; report.py:252-386) PLUS the separate `filter` and `zplane` commands
(cli.py:1362-1393, :1562-1593) that BASELINE.json's metric adds, for every channel of the batch.  Large result
arrays (EDC curves, spectrograms, waterfall slices, log-bin curves, spectra, band signals) are produced in
HBM and stay there; only the fixed-width per-channel metrics record comes back to the host, which is also
what is gathered across ranks (audio_analysis_amd.dist).  PNG rendering is excluded (78 % of the reference's
report time is matplotlib).  Group delay and diffusion
are not part of the step.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional

import numpy as np

from .analyse import decay as _decay
from .analyse import diffusion as _diff
from .analyse import group_delay as _gd
from .analyse import filterplot as _filter
from .analyse import frequency_response as _fr
from .analyse import modalcloud as _modal
from .analyse import rt60bands as _bands
from .analyse import spectrogram as _spec
from .analyse import waterfall as _wf
from .analyse import zplane as _zp
from .engine import ChannelBatch, Engine

# ---- metrics record layout (float64 per channel) ----------------------------------------------------------
MAX_BANDS = 26
M_STATUS, M_NSAMPLES, M_START, M_EARLY10 = 0, 1, 2, 3
M_FIT_EDT, M_FIT_T20, M_FIT_T30 = 4, 12, 20                 # 8 doubles each (ira_curve_fits record)
M_BANDS = 28                                                # MAX_BANDS x (t30, t20, edt)
M_FR_PEAK, M_FR_CENTROID, M_FILT_PEAK, M_FILT_1K = 106, 107, 108, 109
M_SPEC_FRAMES, M_WF_SLICES, M_WF_BINS = 110, 111, 112
M_MODAL_POINTS, M_MODAL_MEDIAN, M_MODAL_P90, M_MODAL_MAX = 113, 114, 115, 116
M_AR_POLES, M_AR_MAX_R, M_AR_MEDIAN_R, M_AR_UNSTABLE = 117, 118, 119, 120
M_NBANDS = 121
M_GD_MEDIAN, M_GD_P10, M_GD_P90 = 122, 123, 124        # section 8f blocks (off by default)
M_DIFF_AC_MEDIAN, M_DIFF_ED_MEDIAN = 125, 126
# How far to trust the pole fit: the solver's condition estimate of the float64 Gram matrix A^T A (between cond and
# order x cond).  <= 1e9: plain normal equations; <= 1e13: refined (corrected semi-normal equations); above: solved in
# double-double arithmetic (engine.Engine.ar_exact_cond) -- all three agree with the reference's SVD-based lstsq to 1e-8.
# NaN: no AR block, or a rank-deficient / non-finite fit.
M_AR_COND = 127
METRICS_WIDTH = 128

# ---- per-IR status (M_STATUS): 0 = analysed; otherwise the bit of the FIRST report block, in the reference's block order
# (report.py:252-386, then the filter / zplane commands), that would have raised ValueError for this channel -- the
# reference aborts the file there (bundle.py:56-67 propagates it).  In a device batch the other channels proceed: the
# failing channel keeps NaN metrics and this code, one bad IR does not poison a batch.
ST_OK = 0
ST_DECAY_TOO_SHORT = 1          # decay.py:146-147      "Not enough samples after trimming/ignoring to compute EDC."
ST_BANDS_TOO_SHORT = 2          # rt60bands.py:346-347  "Not enough samples for rt60bands analysis."
ST_FR_TOO_SHORT = 4             # frequency_response.py:201-202
ST_SPECTROGRAM_TOO_SHORT = 8    # spectrogram.py:197-198 (need at least n_fft samples)
ST_WATERFALL_TOO_SHORT = 16     # waterfall.py:375-376
ST_MODAL_TOO_SHORT = 32         # modalcloud.py:315-316
ST_FILTER_TOO_SHORT = 64        # filterplot.py:141-142
ST_ZPLANE_TOO_SHORT = 128       # zplane.py: fewer than two samples after the time selection
ST_EMPTY_FREQUENCY_RANGE = 256  # frequency_response.py:244-245 (set from the device statistics record)
ST_ZPLANE_NOT_FINITE = 512      # zplane.py:117: numpy.linalg.lstsq raises LinAlgError on NaN / infinite samples
STATUS_MESSAGES = {
    ST_DECAY_TOO_SHORT: "Not enough samples after trimming/ignoring to compute EDC.",
    ST_BANDS_TOO_SHORT: "Not enough samples for rt60bands analysis.",
    ST_FR_TOO_SHORT: "Not enough samples after trimming/selection to analyse spectrum.",
    ST_SPECTROGRAM_TOO_SHORT: "Not enough samples after trimming/selection for spectrogram (need at least n_fft).",
    ST_WATERFALL_TOO_SHORT: "Not enough samples after trimming/selection for waterfall (need at least n_fft).",
    ST_MODAL_TOO_SHORT: "Not enough samples after trimming/selection for modal cloud (need at least n_fft).",
    ST_FILTER_TOO_SHORT: "Not enough samples after trimming/selection to analyse filter response.",
    ST_ZPLANE_TOO_SHORT: "Not enough samples after trimming/selection for the AR fit.",
    ST_EMPTY_FREQUENCY_RANGE: "Selected frequency range is empty.",
    ST_ZPLANE_NOT_FINITE: "SVD did not converge in Linear Least Squares",
}


@dataclass(frozen=True)
class FullReportSettings:
    sample_rate_hz: int = 48_000
    run_decay: bool = True
    run_rt60_bands: bool = True
    run_frequency_response: bool = True
    run_filter: bool = True
    run_spectrogram: bool = True
    run_waterfall: bool = True
    run_modal_cloud: bool = True
    run_zplane: bool = True
    # section 8f rows: default-on blocks of the reference's `report`, off in the headline metric
    run_group_delay: bool = False
    run_diffusion: bool = False
    group_delay: _gd.GroupDelayAnalysisSettings = _gd.GroupDelayAnalysisSettings()
    diffusion: _diff.DiffusionAnalysisSettings = _diff.DiffusionAnalysisSettings(hop_seconds=0.05, max_lag_milliseconds=5.0)
    decay: _decay.DecayAnalysisSettings = _decay.DecayAnalysisSettings()
    rt60_bands: _bands.Rt60BandsAnalysisSettings = _bands.Rt60BandsAnalysisSettings()
    frequency_response: _fr.FrequencyResponseAnalysisSettings = _fr.FrequencyResponseAnalysisSettings()
    filter: _filter.FilterAnalysisSettings = _filter.FilterAnalysisSettings()
    spectrogram: _spec.SpectrogramAnalysisSettings = _spec.SpectrogramAnalysisSettings()
    waterfall: _wf.WaterfallAnalysisSettings = _wf.WaterfallAnalysisSettings()
    modal_cloud: _modal.ModalCloudAnalysisSettings = _modal.ModalCloudAnalysisSettings()
    zplane: _zp.ZPlaneAnalysisSettings = _zp.ZPlaneAnalysisSettings(ar_order=64)

    def blocks(self) -> List[str]:
        names = []
        for flag, label in (("run_decay", "decay"), ("run_rt60_bands", f"rt60bands[{self.rt60_bands.band_mode}]"),
                            ("run_frequency_response", "fr"), ("run_filter", "filter"),
                            ("run_spectrogram", f"spectrogram[{self.spectrogram.n_fft}/{self.spectrogram.hop_length}]"),
                            ("run_waterfall", "waterfall"),
                            ("run_modal_cloud", f"modalcloud[{self.modal_cloud.n_fft}/{self.modal_cloud.hop_length}]"),
                            ("run_zplane", f"zplane[ar{self.zplane.ar_order}]"), ("run_group_delay", "groupdelay"),
                            ("run_diffusion", f"diffusion[hop{self.diffusion.hop_seconds * 1e3:g}ms/lag"
                                              f"{self.diffusion.max_lag_milliseconds:g}ms]")):
            if getattr(self, flag):
                names.append(label)
        return names


def _same_spectrum(a, b) -> bool:
    """fr and filter analyse the identical windowed segment when these fields agree -> share one rFFT."""
    keys = ("trim_to_peak", "ignore_leading_seconds", "analysis_duration_seconds", "use_hann_window",
            "magnitude_floor_db", "f_min_hz", "f_max_hz")
    # The fr block log-smooths its dB curve IN PLACE on the device when smoothing_log_bins > 1
    # (frequency_response.py:117-169); the reference's filterplot has no smoothing and reads the raw spectrum
    # (filterplot.py:152-170), so a smoothed curve is never shared.
    if int(getattr(a, "smoothing_log_bins", 0) or 0) > 1:
        return False
    return all(getattr(a, k) == getattr(b, k) for k in keys)


def _row_median_p90_max(a: np.ndarray):
    """numpy.nanmedian / numpy.nanpercentile(90) / numpy.nanmax along the rows of `a` (every row holds at least one number),
    the same values bit for bit: the mean of the two middle order statistics, and numpy's linear interpolation between
    neighbouring order statistics with its own `_lerp` (a + (b - a) t, or b - (b - a)(1 - t) for t >= 0.5).  Vectorised:
    numpy's nan-functions go row by row through apply_along_axis (9 ms per 256 channels x 240 points, more than the rest
    of a step's host work together)."""
    srt = np.sort(a, axis=1)                                   # NaNs last
    cnt = np.sum(~np.isnan(a), axis=1)
    rows = np.arange(a.shape[0])
    lo, hi = srt[rows, (cnt - 1) // 2], srt[rows, cnt // 2]
    med = np.where(cnt % 2 == 1, lo, (lo + hi) / 2.0)          # numpy: mean of the two middle values = their sum / 2
    virt = (cnt - 1) * np.true_divide(90, 100)
    prev = np.floor(virt)
    t = virt - prev
    ip = prev.astype(np.int64)
    va, vb = srt[rows, ip], srt[rows, np.minimum(ip + 1, cnt - 1)]
    d = vb - va
    p90 = np.where(t >= 0.5, vb - d * (1.0 - t), va + d * t)
    return med, p90, srt[rows, cnt - 1]


class FullReport:
    """Runs the metrics-only report over a ChannelBatch; keeps the big arrays of the last step on the device."""

    def __init__(self, engine: Engine, settings: Optional[FullReportSettings] = None):
        self.eng = engine
        self.s = settings or FullReportSettings()
        self.device_results: Dict[str, dict] = {}

    def prepare(self, batch: ChannelBatch) -> None:
        """Optional: start the peak pick of a batch that has just been uploaded without waiting for it, so that the host
        can finish the previous step meanwhile (bundle path: upload k+1 overlaps the read-back of step k)."""
        self.eng.peaks_begin(batch)

    def run(self, batch: ChannelBatch) -> np.ndarray:
        """One step: submit + finish."""
        return self.finish(self.submit(batch))

    # submit() only ENQUEUES: every kernel of the step, then asynchronous copies of the small result records into
    # pinned memory, then an event.  finish() waits for that event and builds the metrics record on the host.
    # Calling submit(next batch) BEFORE finish(previous) keeps the GPU busy while the host post-processes: the one
    # host round trip a step needs (the peak pick that fixes every block's geometry) runs on a high-priority side
    # stream, so it is not queued behind the previous step's kernels.
    def channel_status(self, batch: ChannelBatch) -> np.ndarray:
        """Per-channel status codes (ST_*) from lengths, peaks and settings alone -- every "too short" ValueError of the
        reference's blocks is a pure function of those.  Needs batch.peak."""
        from .analyse._common import segment_bounds
        s, sr = self.s, self.s.sample_rate_hz
        n = batch.count
        st = np.zeros(n, dtype=np.int64)
        peaks = batch.peak if batch.peak is not None else np.zeros(n, dtype=np.int64)

        def seg_len(i, cfg, duration=True):
            trim = bool(cfg.trim_to_peak)
            dur = getattr(cfg, "analysis_duration_seconds", None) if duration else None
            return segment_bounds(int(batch.length[i]), int(peaks[i]) if trim else 0, sr, trim,
                                  float(cfg.ignore_leading_seconds), dur)[1]

        checks = []
        if s.run_decay:
            checks.append((ST_DECAY_TOO_SHORT, lambda i: seg_len(i, s.decay, duration=False) < 4))
        if s.run_rt60_bands:
            checks.append((ST_BANDS_TOO_SHORT, lambda i: int(batch.length[i]) < 8))
        if s.run_frequency_response:
            checks.append((ST_FR_TOO_SHORT, lambda i: seg_len(i, s.frequency_response) < 32))
        if s.run_spectrogram:
            checks.append((ST_SPECTROGRAM_TOO_SHORT, lambda i: seg_len(i, s.spectrogram) < int(s.spectrogram.n_fft)))
        if s.run_waterfall:
            checks.append((ST_WATERFALL_TOO_SHORT, lambda i: seg_len(i, s.waterfall) < int(s.waterfall.n_fft)))
        if s.run_modal_cloud:
            checks.append((ST_MODAL_TOO_SHORT, lambda i: seg_len(i, s.modal_cloud) < int(s.modal_cloud.n_fft)))
        if s.run_filter:
            checks.append((ST_FILTER_TOO_SHORT, lambda i: seg_len(i, s.filter) < 32))
        if s.run_zplane:
            def z_short(i):
                z = s.zplane
                skip = int(round(float(z.ignore_leading_seconds) * sr))
                start = min(max((int(peaks[i]) if z.trim_to_peak else 0) + skip, 0), int(batch.length[i]))
                left = int(batch.length[i]) - start
                if z.analysis_duration_seconds is not None:
                    left = min(max(1, int(round(float(z.analysis_duration_seconds) * sr))), left)
                return left < 2
            checks.append((ST_ZPLANE_TOO_SHORT, z_short))
        # the common case costs one vectorised comparison: nothing can fail when the shortest post-peak tail is long
        need = max([8, 32] + [int(c.n_fft) for c, on in ((s.spectrogram, s.run_spectrogram), (s.waterfall, s.run_waterfall),
                                                        (s.modal_cloud, s.run_modal_cloud)) if on])
        margin = int(sr * 0.5)
        if n and int(np.min(batch.length - peaks)) >= need + margin and not any(
                getattr(c, "analysis_duration_seconds", None) is not None or float(c.ignore_leading_seconds) > 0.4
                for c in (s.decay, s.frequency_response, s.filter, s.spectrogram, s.waterfall, s.modal_cloud, s.zplane)):
            return st
        for i in range(n):
            for code, bad in checks:
                if bad(i):
                    st[i] = code                          # the FIRST failing block, like the reference's abort
                    break
        return st

    def submit(self, batch: ChannelBatch) -> dict:
        """Enqueue one step.  Channels the reference would refuse (too few samples for an enabled block) are left out of
        the device work and come back with their status code and NaN metrics; the rest of the batch proceeds."""
        eng = self.eng
        if batch.peak is None:
            eng.peaks_begin(batch)                         # side stream, behind the upload of this batch and nothing else
            eng.peaks(batch)                               # the one host round trip every block's geometry needs
        status = self.channel_status(batch)
        if not status.any():
            return self._submit_valid(batch)
        good = np.nonzero(status == 0)[0]
        h = self._submit_valid(eng.subset(batch, good)) if good.size else None
        return dict(scatter=True, inner=h, good=good, status=status, n=batch.count, length=batch.length.copy())

    def _submit_valid(self, batch: ChannelBatch) -> dict:
        eng, s, sr = self.eng, self.s, self.s.sample_rate_hz
        t = eng.torch
        n = batch.count
        m = np.full((n, METRICS_WIDTH), np.nan, dtype=np.float64)
        m[:, M_STATUS] = 0.0
        m[:, M_NSAMPLES] = batch.length
        res: Dict[str, dict] = {}
        fut: Dict[str, object] = {}
        state = {"spectrum": None, "filt": None}

        # ---- independent groups of blocks; Engine.block_streams() decides how many streams they are dealt onto -----------
        def bands_group():
            if s.run_rt60_bands:
                bands, band_values, have = _bands.rt60_bands_device(eng, batch, sr, s.rt60_bands, defer=True)
                res["rt60bands"] = dict(bands=bands, values=band_values, have=have)

        def spectrum_group():
            spectrum = None
            if s.run_frequency_response:
                share = s.run_filter and _same_spectrum(s.frequency_response, s.filter)
                spectrum = _fr.spectrum_device(eng, batch, sr, s.frequency_response, "spectrum", want_phase=share,
                                               unwrap=bool(s.filter.unwrap_phase),
                                               degrees=s.filter.phase_mode == "degrees")
                res["spectrum"] = spectrum
                fut["spectrum_stats"] = eng.fetch(spectrum["stats"])
            filt = None
            if s.run_filter:
                if spectrum is not None and spectrum["phase"] is not None:
                    filt = spectrum
                else:
                    filt = _fr.spectrum_device(eng, batch, sr, s.filter, "filter response", want_phase=True,
                                               unwrap=bool(s.filter.unwrap_phase),
                                               degrees=s.filter.phase_mode == "degrees")
                    res["filter"] = filt
                    fut["filter_stats"] = eng.fetch(filt["stats"])
            state["spectrum"], state["filt"] = spectrum, filt
            if s.run_group_delay:
                gdev = _gd.group_delay_device(eng, batch, sr, s.group_delay)
                res["groupdelay"] = gdev
                fut["gd_stats"] = _gd.summary_statistics_device(eng, gdev, sr, s.group_delay)

        def modal_group():
            if s.run_modal_cloud:
                mc = _modal.modal_cloud_device(eng, batch, sr, s.modal_cloud)
                res["modal"] = mc
                fut["modal_fits"] = eng.fetch(mc["fits"])

        def decay_group():
            if s.run_decay:
                d = _decay.decay_device(eng, batch, sr, s.decay)
                res["decay"] = d
                fut["decay_fits"], fut["decay_cross"] = eng.fetch(d["fits"]), eng.fetch(d["cross"])

        def stft_group():
            if s.run_spectrogram:
                # nothing downstream reads the matrix in this pipeline: take the frame-major layout where the kernel has it
                sp = _spec.spectrogram_device(eng, batch, sr, s.spectrogram, frame_major=True)
                res["spectrogram"] = sp
                m[:, M_SPEC_FRAMES] = sp["cols"]
            if s.run_waterfall:
                wf = _wf.waterfall_device(eng, batch, sr, s.waterfall)
                res["waterfall"] = wf
                m[:, M_WF_SLICES] = wf["cols"]
                m[:, M_WF_BINS] = wf["nsel"]
            if s.run_diffusion:
                ddev = _diff.diffusion_device(eng, batch, sr, s.diffusion)
                res["diffusion"] = ddev
                fut["diff_ac"], fut["diff_ed"] = eng.fetch(ddev["ac"]), eng.fetch(ddev["ed"])

        def zplane_group():
            if s.run_zplane:
                res["zplane"] = dict(finish=_zp.zplane_device(eng, batch, sr, s.zplane, defer=True, with_status=True))

        main = t.cuda.current_stream(eng.device)
        lanes = eng.block_streams()
        groups = [bands_group, spectrum_group, zplane_group, decay_group, modal_group, stft_group]
        done = []
        if lanes is None:
            # one stream: the batch may have been uploaded (and its offset / length tables allocated) on a feed's copy
            # stream -- wait for its upload and keep the allocator from recycling its arrays under this stream's kernels
            if batch.ready is not None:
                main.wait_event(batch.ready)
            for buf in (batch.x, batch.off_dev, batch.len_dev):
                buf.record_stream(main)
            for work in groups:
                work()
            ev = t.cuda.Event()
            ev.record(main)
            done.append(ev)
        else:
            # Round 3, measured at 256 x 10 s per step (each deal three times, alternating, one box): two lanes -- the long
            # transforms and the Schroeder fits (memory-side work) on one, the float64 / float32 STFTs and the AR fit
            # (arithmetic) on the other -- 13.4-13.5 k IRs/s; [[0, 1], [4, 3, 5, 2]] 12.3-13.1 k; [[0, 1, 2], [4, 3, 5]]
            # 13.1-13.2 k; three lanes (round 2's default, bands | spectrum | rest: two families of long transforms side by
            # side evict each other's work arrays) 11.6-12.3 k; one lane 12.4 k.
            deal = {2: [[0, 1, 3], [4, 5, 2]], 3: [[0], [1], [4, 3, 5, 2]], 4: [[0], [1], [4, 3], [5, 2]]}[len(lanes)]
            if eng.lane_deal is not None and len(eng.lane_deal) == len(lanes):      # A/B (tools only)
                deal = eng.lane_deal
            # A lane waits for THIS batch's upload only (not for another lane's previous step: steps overlap across
            # lanes) and is ordered behind its own earlier work by being a stream.  The batch's arrays were allocated
            # on the caller's stream: record_stream keeps the allocator from recycling them while a lane may read them.
            for lane, members in zip(lanes, deal):
                if batch.ready is not None:
                    lane.wait_event(batch.ready)
                else:
                    lane.wait_stream(main)
                for buf in (batch.x, batch.off_dev, batch.len_dev):
                    buf.record_stream(lane)
                with t.cuda.stream(lane):
                    for g in members:
                        groups[g]()
                    ev = t.cuda.Event()
                    ev.record(lane)
                    done.append(ev)
        spectrum, filt = state["spectrum"], state["filt"]
        # the handle keeps the batch (its device tables) alive until finish() has seen every lane's event
        return dict(n=n, m=m, res=res, fut=fut, done=done, spectrum=spectrum, filt=filt, batch=batch)

    def finish(self, h: dict) -> np.ndarray:
        if h.get("scatter"):
            full = np.full((h["n"], METRICS_WIDTH), np.nan, dtype=np.float64)
            full[:, M_STATUS] = h["status"]
            full[:, M_NSAMPLES] = h["length"]
            if h["inner"] is not None:
                full[h["good"]] = self._finish_valid(h["inner"])
            return full
        return self._finish_valid(h)

    def _finish_valid(self, h: dict) -> np.ndarray:
        s = self.s
        n, m, res, fut, spectrum, filt = h["n"], h["m"], h["res"], h["fut"], h["spectrum"], h["filt"]
        for ev in h["done"]:                               # the only wait of the step (one event per lane)
            ev.synchronize()
        # ---- the fixed-width record from the small result records (already in pinned host memory) ---------------
        if s.run_rt60_bands:
            bands = res["rt60bands"]["bands"]
            values = res["rt60bands"]["values"]
            values = values() if callable(values) else values
            nb = min(len(bands), MAX_BANDS)
            m[:, M_NBANDS] = nb
            if nb:
                m[:, M_BANDS : M_BANDS + 3 * nb] = values[:, :nb, :].reshape(n, 3 * nb)
        if s.run_zplane:
            poles, _, ar_status, ar_cond = res["zplane"]["finish"]()
            m[:, M_AR_COND] = np.where(ar_status == 4.0, np.nan, ar_cond)      # status 4: info[3] holds the rank instead
            not_finite = ar_status == _zp.AR_STATUS_NOT_FINITE
            m[not_finite & (m[:, M_STATUS] == 0.0), M_STATUS] = float(ST_ZPLANE_NOT_FINITE)
            sizes = np.array([p.size for p in poles])
            m[:, M_AR_POLES] = sizes
            if n and sizes.min() == sizes.max() and sizes[0] > 0:
                rad = np.abs(np.stack(poles))
                m[:, M_AR_MAX_R], m[:, M_AR_MEDIAN_R] = rad.max(axis=1), np.median(rad, axis=1)
                m[:, M_AR_UNSTABLE] = (rad >= 1.0).sum(axis=1)
            else:
                for i, p in enumerate(poles):
                    if p.size:
                        rad = np.abs(p)
                        m[i, M_AR_MAX_R], m[i, M_AR_MEDIAN_R] = float(rad.max()), float(np.median(rad))
                        m[i, M_AR_UNSTABLE] = int(np.sum(rad >= 1.0))
        if s.run_decay:
            d = res["decay"]
            fits = fut["decay_fits"].get()
            cross = fut["decay_cross"].get()
            m[:, M_START] = d["starts"]
            ok = ~np.isnan(cross[:, 0]) & ~np.isnan(cross[:, 1]) & (cross[:, 1] >= cross[:, 0])
            m[ok, M_EARLY10] = cross[ok, 1] - cross[ok, 0]
            slot = {"EDT": M_FIT_EDT, "T20": M_FIT_T20, "T30": M_FIT_T30}
            for j, (name, _) in enumerate(d["specs"]):
                m[:, slot[name] : slot[name] + 8] = fits[:, j, :]
        if spectrum is not None:
            st = fut["spectrum_stats"].get()
            m[:, M_FR_PEAK] = st[:, 2]
            with np.errstate(invalid="ignore", divide="ignore"):
                m[:, M_FR_CENTROID] = np.where(st[:, 4] > 0.0, st[:, 3] / st[:, 4], st[:, 5])
            m[st[:, 0] < 1.0, M_STATUS] = float(ST_EMPTY_FREQUENCY_RANGE)
        if filt is not None:
            st = fut["filter_stats"].get() if filt is not spectrum else st
            m[:, M_FILT_PEAK] = st[:, 2]
            m[:, M_FILT_1K] = st[:, 7]
        if s.run_modal_cloud:
            rec = fut["modal_fits"].get().reshape(n, res["modal"]["nbins"], 8)
            valid = rec[:, :, 0] == 1.0
            m[:, M_MODAL_POINTS] = valid.sum(axis=1)
            rt = np.where(valid, rec[:, :, 6], np.nan)
            some = valid.any(axis=1)
            if some.any():
                med, p90, mx = _row_median_p90_max(rt[some])
                m[some, M_MODAL_MEDIAN], m[some, M_MODAL_P90], m[some, M_MODAL_MAX] = med, p90, mx
        if s.run_group_delay:
            vals, gam, cnt = fut["gd_stats"]
            m[:, M_GD_MEDIAN : M_GD_P90 + 1] = _gd.finish_summary_statistics(vals.get(), gam, cnt)
        if s.run_diffusion:
            import warnings
            ac, ed, dd = fut["diff_ac"].get(), fut["diff_ed"].get(), res["diffusion"]
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", category=RuntimeWarning)          # all-NaN series stay NaN
                for i in range(n):
                    o, f = int(dd["off"][i]), int(dd["frames"][i])
                    m[i, M_DIFF_AC_MEDIAN] = float(np.nanmedian(ac[o : o + f]))
                    m[i, M_DIFF_ED_MEDIAN] = float(np.nanmedian(ed[o : o + f]))
        self.device_results = res
        h.pop("batch", None)
        return m
