"""
Report mode: run the standard suite of analyses on one WAV and write "<basename>_report.md".

Host-side mirror of the reference's analyse/report.py (ReportSettings :96-132, ReportResults :135-140,
run_report_from_wav_file :222-398): same block order, headings, image-link file names (including the
reference's hard-coded _left/_right links and the un-suffixed group-delay link), summary code blocks and
header block.  Differences, all outside the accelerated path:
  * the WAV is read ONCE and uploaded ONCE (native int16 ingest, audio_analysis_amd.ingest); every block runs on
    that one device-resident batch (the reference re-reads and re-converts the file ten times);
  * run_reports_batched runs every block ONCE over the channels of MANY files and
    writes each file's Markdown from its slice of the results -- string-identical to the per-file path; PNG rendering
    can be handed to a pool of CPU worker processes (`plot_pool`) so it leaves the critical path;
  * the impulse-response waveform plots (no numerics) are drawn on the CPU from the host copy of the file, only when
    PNGs are rendered at all; their Markdown block is always the reference's.  Group delay and diffusion (section 8f
    rows) run on the GPU;
  * `render_plots=False` (extra field, default True) skips the CPU-side PNG rendering.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, replace
from pathlib import Path
from typing import Any, Callable, List, Optional, Sequence, Tuple

from ..engine import get_engine
from . import decay as _decay
from . import diffusion as _diff
from . import frequency_response as _fr
from . import group_delay as _gd
from . import impulse_response as _irv
from . import modalcloud as _modal
from . import plotting
from . import rt60bands as _bands
from . import spectrogram as _spec
from . import waterfall as _wf
from .io import DEFAULT_EXPECTED_SAMPLE_RATE_HZ
from .decay import DecayAnalysisSettings, DecayPlotSettings
from .diffusion import DiffusionAnalysisSettings
from .frequency_response import FrequencyResponseAnalysisSettings, FrequencyResponsePlotSettings
from .group_delay import GroupDelayAnalysisSettings, GroupDelayPlotSettings
from .modalcloud import ModalCloudAnalysisSettings, ModalCloudPlotSettings
from .rt60bands import Rt60BandsAnalysisSettings, Rt60BandsPlotSettings
from .spectrogram import SpectrogramAnalysisSettings, SpectrogramPlotSettings
from .waterfall import WaterfallAnalysisSettings, WaterfallPlotSettings


@dataclass(frozen=True)
class ReportSettings:
    common_use_mono_downmix_for_stereo: bool = False
    common_trim_to_peak: bool = True
    common_ignore_leading_seconds: float = 0.0

    run_impulse_response_plots: bool = True
    run_decay: bool = True
    run_rt60_bands: bool = True
    run_frequency_response: bool = True
    run_group_delay: bool = True
    run_spectrogram: bool = True
    run_waterfall: bool = True
    run_diffusion: bool = True
    run_modal_cloud: bool = True
    run_echo_density: bool = True

    expected_sample_rate_hz: int = DEFAULT_EXPECTED_SAMPLE_RATE_HZ

    ir_view_settings: Optional[Any] = None
    decay_analysis_settings: Optional[DecayAnalysisSettings] = None
    decay_plot_settings: Optional[DecayPlotSettings] = None
    rt60_bands_settings: Optional[Rt60BandsAnalysisSettings] = None
    rt60_bands_plot_settings: Optional[Rt60BandsPlotSettings] = None
    frequency_response_analysis_settings: Optional[FrequencyResponseAnalysisSettings] = None
    frequency_response_plot_settings: Optional[FrequencyResponsePlotSettings] = None
    group_delay_analysis_settings: Optional[Any] = None
    group_delay_plot_settings: Optional[Any] = None
    spectrogram_analysis_settings: Optional[SpectrogramAnalysisSettings] = None
    spectrogram_plot_settings: Optional[SpectrogramPlotSettings] = None
    waterfall_analysis_settings: Optional[WaterfallAnalysisSettings] = None
    waterfall_plot_settings: Optional[WaterfallPlotSettings] = None
    diffusion_analysis_settings: Optional[Any] = None
    modal_cloud_analysis_settings: Optional[ModalCloudAnalysisSettings] = None
    modal_cloud_plot_settings: Optional[ModalCloudPlotSettings] = None

    render_plots: bool = True


@dataclass(frozen=True)
class ReportResults:
    input_wav_file_path: Path
    output_basename: Path
    summary_markdown_path: Path
    summary_markdown: str


def _section(title: str) -> str:
    return f"\n## {title}\n\n"


def _code(text: str) -> str:
    text = text.strip()
    return f"```text\n{text}\n```\n" if text else "_(no output)_\n"


def _image(basename: Path, suffix: str, alt: str = "") -> str:
    name = f"{basename.name}{suffix}.png"
    return f"![{alt or name}]({name})\n\n"


def _apply_common_overrides(settings_obj: Any, report_settings: ReportSettings) -> Any:
    """Fan the three common_* fields into any settings dataclass that has the matching field."""
    if settings_obj is None:
        return None
    names = {f.name for f in dataclasses.fields(settings_obj)}
    common = {
        "use_mono_downmix_for_stereo": report_settings.common_use_mono_downmix_for_stereo,
        "trim_to_peak": report_settings.common_trim_to_peak,
        "ignore_leading_seconds": report_settings.common_ignore_leading_seconds,
    }
    picked = {k: v for k, v in common.items() if k in names}
    return replace(settings_obj, **picked) if picked else settings_obj


def _header(path: Path, frames: int, expected_rate: int) -> str:
    # the reference loads with upmix for this block, so it always reports two channels (report.py:193-214)
    dur = frames / expected_rate if expected_rate > 0 else 0.0
    return (
        "# Offline Reverb Analysis Report\n\n"
        f"**Input WAV:** `{path}`  \n"
        f"**Sample rate:** {expected_rate} Hz (expected {expected_rate} Hz)  \n"
        f"**Channels:** 2  \n"
        f"**Samples:** {frames}  \n"
        f"**Duration:** {dur:.6f} s\n\n"
        "---\n"
    )


def _render(task) -> None:
    """One deferred PNG: (plotting function name, argument tuple).  Module-level so worker processes can run it."""
    name, args = task
    getattr(plotting, name)(*args)


class PlotPool:
    """
    CPU worker processes that render the report PNGs off the critical path (the
    reference renders inline and spends 78 % of a report there, section 3.1).  Workers are spawned (never forked from
    a process that holds the GPU), import matplotlib only, and receive the result dataclasses by pickle.
    """

    def __init__(self, workers: Optional[int] = None):
        import multiprocessing as mp
        import os
        from concurrent.futures import ProcessPoolExecutor
        n = workers or max(1, min(8, (len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 2) - 1))
        self._ex = ProcessPoolExecutor(max_workers=n, mp_context=mp.get_context("spawn"))
        self._pending = []

    def submit(self, task) -> None:
        self._pending.append(self._ex.submit(_render, task))

    def wait(self) -> int:
        """Block until every submitted PNG is written; re-raises the first rendering error.  Returns the count."""
        done = 0
        pending, self._pending = self._pending, []
        for f in pending:
            f.result()
            done += 1
        return done

    def close(self) -> None:
        self.wait()
        self._ex.shutdown()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def _texts_per_file(per_channel: Sequence[str], labels: Sequence[Tuple[int, str]], nfiles: int, joiner: str = "\n"
                    ) -> List[str]:
    """Per-file summary text from one text piece per channel (the summaries are one entry per channel, in order)."""
    out: List[List[str]] = [[] for _ in range(nfiles)]
    for piece, (i, _) in zip(per_channel, labels):
        out[i].append(piece)
    return [joiner.join(p) for p in out]


def _group(results: list, labels: Sequence[Tuple[int, str]], nfiles: int) -> List[list]:
    out: List[list] = [[] for _ in range(nfiles)]
    for r, (i, _) in zip(results, labels):
        out[i].append(r)
    return out


def run_reports_batched(
    items: Sequence[Tuple[str | Path, str | Path]],
    settings: Optional[ReportSettings] = None,
    plot_pool: Optional[PlotPool] = None,
) -> List[ReportResults]:
    """
    items = [(input WAV, output basename), ...].  Every enabled block runs ONCE over the channels of all files (one
    device batch per channel policy); each file's "<basename>_report.md" is then assembled from its slice.  With
    `plot_pool` the PNGs are rendered by worker processes (call plot_pool.wait() before reading them); without it
    they are rendered inline like the reference does.
    """
    from ..ingest import TapSet

    settings = settings or ReportSettings()
    wavs = [Path(a) for a, _ in items]
    bases = [Path(b) for _, b in items]
    nf = len(wavs)
    for b in bases:
        b.parent.mkdir(parents=True, exist_ok=True)
    show = False
    draw = bool(settings.render_plots)
    plot: Callable = plot_pool.submit if plot_pool is not None else _render

    eng = get_engine()
    taps = TapSet(eng, wavs, settings.expected_sample_rate_hz)        # one read, one upload
    sr = settings.expected_sample_rate_hz

    def view(mono_downmix: bool):
        batch, labels = taps.view(bool(mono_downmix))
        return [n for _, n in labels], batch, labels

    names, batch, labels = view(settings.common_use_mono_downmix_for_stereo)
    mono_mix = settings.common_use_mono_downmix_for_stereo

    md: List[List[str]] = [[_header(w, info.frames, settings.expected_sample_rate_hz)]
                           for w, info in zip(wavs, taps.infos)]
    skipped: List[str] = []

    if settings.run_impulse_response_plots:
        ir = _apply_common_overrides(settings.ir_view_settings or _irv.ImpulseResponseViewSettings(), settings)
        for f in range(nf):
            if draw:
                plot(("render_ir_views", (str(wavs[f]), ir, str(bases[f]), settings.expected_sample_rate_hz)))
            md[f] += [_section("Impulse response"), _image(bases[f], "", "Impulse response overview"),
                      _image(bases[f], "_early", "Early reflections"), _image(bases[f], "_tail", "Tail (log magnitude)")]

    if settings.run_decay:
        s = _apply_common_overrides(settings.decay_analysis_settings or DecayAnalysisSettings(), settings)
        dev = _decay.decay_device(eng, batch, sr, s)
        if draw:
            res = _decay.decay_records_to_results(dev, dev["fits"].cpu().numpy(), dev["cross"].cpu().numpy(),
                                                  dev["edc"].cpu().numpy(), sr, names)
        else:                                   # text only: the EDC curves stay in HBM
            res = _decay.decay_results_without_curves(dev, sr, names)
        for f, r in enumerate(_group(res, labels, nf)):
            if draw:
                plot(("render_decay", (r, s, settings.decay_plot_settings or DecayPlotSettings(),
                                       f"Decay (EDC) — {wavs[f]}", plotting.png_path(bases[f], "_decay"), show)))
            md[f] += [_section("Decay / EDC"), _image(bases[f], "_decay", "Decay analysis (T20/T30/RT60/EDT)"),
                      _code(_decay.summarise_decay_results_text(r))]

    if settings.run_rt60_bands:
        # Reference quirk kept: Rt60BandsAnalysisSettings has none of the three common_* field names (they live
        # in its nested decay_settings), so the overrides do not reach this block -- it analyses left/right with
        # its own trim/ignore policy even in a --mono report (report.py:267-269 with :172-186).
        s = _apply_common_overrides(settings.rt60_bands_settings or Rt60BandsAnalysisSettings(), settings)
        b_names, b_batch, b_labels = view(s.decay_settings.use_mono_downmix_for_stereo)
        bands, values, have = _bands.rt60_bands_device(eng, b_batch, sr, s)
        res = _bands.rt60_bands_results(bands, values, have, sr, b_names)
        ps = settings.rt60_bands_plot_settings or Rt60BandsPlotSettings()
        if ps.legend_values and str(s.band_mode).lower() in ("octave", "third"):
            ps = replace(ps, legend_values=False)
        for f, r in enumerate(_group(res, b_labels, nf)):
            if draw:
                plot(("render_rt60_bands", (r, s, ps, f"RT60 bands — {wavs[f]}",
                                            plotting.png_path(bases[f], "_rt60bands"), show)))
            md[f] += [_section("RT60 by band"), _image(bases[f], "_rt60bands", "RT60 by frequency band"),
                      _code(_bands.summarise_rt60_bands_results_text(r, bool(s.include_t20), bool(s.include_edt)))]

    if settings.run_frequency_response:
        s = _apply_common_overrides(settings.frequency_response_analysis_settings
                                    or FrequencyResponseAnalysisSettings(), settings)
        dev = _fr.spectrum_device(eng, batch, sr, s, "spectrum")
        if draw:
            res = _fr.frequency_response_results(dev, sr, names, s)
            texts = []
            for f, r in enumerate(_group(res, labels, nf)):
                plot(("render_frequency_response", (r, s, settings.frequency_response_plot_settings
                                                    or FrequencyResponsePlotSettings(),
                                                    f"Frequency response (spectrum) — {wavs[f]}",
                                                    plotting.png_path(bases[f], "_fr"), show)))
                texts.append(_fr.summarise_frequency_response_results_text(r))
        else:                                   # text only: peak / centroid from the statistics records
            texts = _texts_per_file(_fr.frequency_response_summary_lines(dev, sr, names, s), labels, nf)
        for f in range(nf):
            md[f] += [_section("Frequency response"), _image(bases[f], "_fr", "Frequency response spectrum"),
                      _code(texts[f])]

    if settings.run_group_delay:
        s = _apply_common_overrides(settings.group_delay_analysis_settings or GroupDelayAnalysisSettings(), settings)
        g_names, g_batch, g_labels = view(s.use_mono_downmix_for_stereo)
        dev = _gd.group_delay_device(eng, g_batch, sr, s)
        ps = settings.group_delay_plot_settings or GroupDelayPlotSettings()
        if draw:
            texts = []
            for f, rs in enumerate(_group(_gd.group_delay_results(dev, sr, g_names, s), g_labels, nf)):
                for r in rs:
                    plot(("render_group_delay", (r, s, ps, f"Group delay ({r.channel_name})",
                                                 plotting.png_path(bases[f], f"_groupdelay_{r.channel_name}"), show)))
                texts.append(_gd.summarise_group_delay_results_text(rs))
        else:                                   # text only: median / p10 / p90 by device-side order statistics
            lines = _gd.group_delay_summary_lines(eng, dev, sr, g_names, s)
            texts = [_gd.join_group_delay_summary(g) for g in _group(lines, g_labels, nf)]
        for f in range(nf):
            md[f] += [_section("Group delay"), _image(bases[f], "_groupdelay", "Group delay vs frequency"),
                      _code(texts[f])]

    if settings.run_spectrogram:
        s = _apply_common_overrides(settings.spectrogram_analysis_settings or SpectrogramAnalysisSettings(), settings)
        dev = _spec.spectrogram_device(eng, batch, sr, s, frame_major=not draw)
        if draw:
            texts = []
            for f, rs in enumerate(_group(_spec.spectrogram_results(dev, sr, names, s), labels, nf)):
                for r in rs:
                    plot(("render_spectrogram", (r, s, settings.spectrogram_plot_settings or SpectrogramPlotSettings(),
                                                 f"Spectrogram — {wavs[f]} — {r.channel_name}",
                                                 plotting.png_path(bases[f], f"_spectrogram_{r.channel_name}"), show)))
                texts.append(_spec.summarise_spectrogram_results_text(rs))
        else:                                   # text only: the matrices stay in HBM
            texts = _texts_per_file(_spec.spectrogram_summary_lines(dev, sr, names, s), labels, nf)
        for f in range(nf):
            md[f] += [_section("Spectrogram"), _image(bases[f], "_spectrogram_left", "Spectrogram (left)")]
            if not mono_mix:
                md[f].append(_image(bases[f], "_spectrogram_right", "Spectrogram (right)"))
            md[f].append(_code(texts[f]))

    if settings.run_waterfall:
        s = _apply_common_overrides(settings.waterfall_analysis_settings or WaterfallAnalysisSettings(), settings)
        dev = _wf.waterfall_device(eng, batch, sr, s)
        if draw:
            texts = []
            for f, rs in enumerate(_group(_wf.waterfall_results(dev, sr, names, s), labels, nf)):
                for r in rs:
                    plot(("render_waterfall", (r, s, settings.waterfall_plot_settings or WaterfallPlotSettings(),
                                               f"Waterfall — {wavs[f]} — {r.channel_name}",
                                               plotting.png_path(bases[f], f"_waterfall_{r.channel_name}"), show)))
                texts.append(_wf.summarise_waterfall_results_text(rs))
        else:                                   # text only: the slice blocks stay in HBM
            texts = _texts_per_file(_wf.waterfall_summary_lines(dev, sr, names), labels, nf)
        for f in range(nf):
            md[f] += [_section("Waterfall"), _image(bases[f], "_waterfall_left", "Waterfall plot (left)")]
            if not mono_mix:
                md[f].append(_image(bases[f], "_waterfall_right", "Waterfall plot (right)"))
            md[f].append(_code(texts[f]))

    if settings.run_diffusion:
        s = _apply_common_overrides(settings.diffusion_analysis_settings
                                    or DiffusionAnalysisSettings(hop_seconds=0.05, max_lag_milliseconds=5.0), settings)
        d_names, d_batch, d_labels = view(s.use_mono_downmix_for_stereo)
        res = _diff.diffusion_results(_diff.diffusion_device(eng, d_batch, sr, s), sr, d_names)
        grouped = _group(res, d_labels, nf)
        if not s.use_mono_downmix_for_stereo:
            first = {}
            for k, (i, _) in enumerate(d_labels):
                first.setdefault(i, k)
            stereo = [f for f in range(nf) if len(grouped[f]) == 2]
            if stereo:
                series = _diff.stereo_series_device(eng, d_batch, [first[f] for f in stereo],
                                                    taps.mix_peaks(stereo), sr, s)
                for f, (corr0, iacc) in zip(stereo, series):
                    grouped[f] = [_diff.DiffusionChannelResult(
                        channel_name=r.channel_name, sample_rate_hz=r.sample_rate_hz,
                        series=dataclasses.replace(r.series, corr0=corr0, iacc_max=iacc)) for r in grouped[f]]
        for f, rs in enumerate(grouped):
            if draw:
                plot(("render_diffusion", (rs, f"Diffusion — {wavs[f]}", plotting.png_path(bases[f], "_diffusion"), show)))
            md[f] += [_section("Diffusion / echo density proxy"),
                      _image(bases[f], "_diffusion", "Diffusion metrics over time"),
                      _code(_diff.summarise_diffusion_results_text(rs))]

    if settings.run_modal_cloud:
        s = _apply_common_overrides(settings.modal_cloud_analysis_settings or ModalCloudAnalysisSettings(), settings)
        dev = _modal.modal_cloud_device(eng, batch, sr, s)
        res = _modal.modal_records_to_results(dev, dev["fits"].cpu().numpy(), sr, names)
        for f, rs in enumerate(_group(res, labels, nf)):
            if draw:
                for r in rs:
                    plot(("render_modal_cloud", (r, s, settings.modal_cloud_plot_settings or ModalCloudPlotSettings(),
                                                 f"Modal cloud — {wavs[f]} — {r.channel_name}",
                                                 plotting.png_path(bases[f], f"_modalcloud_{r.channel_name}"), show)))
            md[f] += [_section("Modal cloud"), _image(bases[f], "_modalcloud_left", "Modal cloud (left)")]
            if not mono_mix:
                md[f].append(_image(bases[f], "_modalcloud_right", "Modal cloud (right)"))
            md[f].append(_code(_modal.summarise_modal_cloud_results_text(rs)))

    out: List[ReportResults] = []
    for f in range(nf):
        if skipped:
            md[f] += [_section("Skipped blocks"),
                      "_(not part of the GPU-accelerated hot path: " + ", ".join(skipped) + ")_\n"]
        text = "".join(md[f]).rstrip() + "\n"
        out_path = Path(f"{bases[f]}_report.md")
        out_path.parent.mkdir(parents=True, exist_ok=True)
        out_path.write_text(text, encoding="utf-8")
        out.append(ReportResults(input_wav_file_path=wavs[f], output_basename=bases[f], summary_markdown_path=out_path,
                                 summary_markdown=text))
    return out


def run_report_from_wav_file(
    input_wav_file_path: str | Path,
    output_basename: str | Path,
    settings: Optional[ReportSettings] = None,
) -> ReportResults:
    """The reference's entry point (report.py:222-398): one file = a batch of one."""
    return run_reports_batched([(input_wav_file_path, output_basename)], settings)[0]
