"""
Command-line interface: `python -m audio_analysis_amd.analyse.cli <command> ...` (also `python -m analyse.cli`
through the top-level shim package).  Same sub-commands, flag spellings, dests and defaults as the reference's
analyse/cli.py (:110-1186 parser, :1210-1662 dispatch) for the commands on the accelerated path:
ir, zplane, groupdelay, diffusion, deconvolve, bundle, decay, rt60bands, fr, filter, spectrogram, waterfall, modalcloud, report.
The reference's inconsistent spellings are kept verbatim (--no_show vs --no-show, --ignore-leading vs
--ignore_leading_seconds, rt60bands --trim_to_peak being store_true with default True).
ir draws its three waveform PNGs on the CPU (there are no numerics in it).

The parser is table driven: one row per flag.
"""
from __future__ import annotations

import argparse
from pathlib import Path
from typing import Optional

BOOL = argparse.BooleanOptionalAction
INPUT = ("--input", dict(dest="input_wav_file_path", type=str, required=True))
OUTPUT = ("--output", dict(dest="output_basename", type=str, default=None))
NO_SHOW = ("--no_show", dict(action="store_true"))
MONO = ("--mono", dict(dest="use_mono_downmix", action="store_true", default=False))
TRIM = ("--trim_to_peak", dict(action=BOOL, default=True))
IGNORE = ("--ignore-leading", dict(dest="ignore_leading_seconds", type=float, default=0.0))
DURATION = ("--duration", dict(dest="analysis_duration_seconds", type=float, default=None))
NO_HANN = ("--no_hann_window", dict(action="store_true"))


def F(name, default, **kw):
    return (name, dict(type=float, default=default, **kw))


def I(name, default, **kw):
    return (name, dict(type=int, default=default, **kw))


def S(name, default, choices=None, **kw):
    return (name, dict(type=str, default=default, choices=choices, **kw))


COMMANDS = {
    "ir": [
        INPUT, F("--early-window", 0.08, dest="early_window_seconds"), F("--floor-db", -120.0, dest="log_magnitude_floor_db"),
        ("--mono", dict(dest="use_mono_downmix", action="store_true")), OUTPUT, NO_SHOW,
    ],
    "zplane": [
        INPUT, OUTPUT, ("--no-show", dict(dest="no_show", action="store_true")),
        ("--mono", dict(dest="use_mono_downmix_for_stereo", action="store_true")),
        ("--no-trim", dict(dest="trim_to_peak", action="store_false")), IGNORE, DURATION,
        I("--ar-order", 256, dest="ar_order"), ("--zeros", dict(dest="derive_zeros", action="store_true")),
        I("--zero-order", 64, dest="zero_order"), F("--radius", 1.2, dest="limit_radius"),
        F("--ridge", 0.0, dest="ridge_lambda"),
    ],
    "groupdelay": [
        INPUT, OUTPUT, ("--no-show", dict(dest="no_show", action="store_true")),
        ("--mono", dict(dest="use_mono_downmix_for_stereo", action="store_true")),
        ("--no-trim", dict(dest="trim_to_peak", action="store_false")), IGNORE, DURATION,
        I("--fft", None, dest="fft_size"), I("--smooth", 0, dest="smoothing_bins"),
        F("--fmin", 20.0, dest="f_min_hz"), F("--fmax", 20000.0, dest="f_max_hz"),
    ],
    "diffusion": [
        INPUT, OUTPUT, NO_SHOW, MONO, TRIM, IGNORE, F("--window_seconds", 0.05), F("--hop_seconds", 0.01),
        F("--max_lag_milliseconds", 10.0), F("--echo_density_threshold_rms", 1.0),
        ("--echo_density_normalise_to_gaussian", dict(action=BOOL, default=True)),
    ],
    "deconvolve": [
        ("--recorded_wav_file_path", dict(type=str, required=True)), ("--sweep_wav_file_path", dict(type=str, required=True)),
        ("--output_ir_wav_file_path", dict(type=str, default=None)), F("--regularization_relative", 1e-10),
        ("--normalise_peak", dict(action=BOOL, default=True)), F("--target_peak", 0.95),
        ("--remove_dc", dict(action=BOOL, default=True)),
        S("--output_length_mode", "recorded", ["recorded", "full_fft"]),
    ],
    "bundle": [("--input", dict(dest="bundle_root", type=str, required=True)),
               S("--reports-subdir", "reports", dest="reports_subdir")],
    "decay": [
        INPUT, OUTPUT, NO_SHOW, TRIM, IGNORE, F("--edc_floor_db", -120.0), F("--fit_lower_limit_db", -80.0),
        I("--smoothing", 0, dest="edc_smoothing_window_samples"), MONO, ("--compute_edt", dict(action=BOOL, default=True)),
    ],
    "rt60bands": [
        INPUT, OUTPUT, NO_SHOW, S("--band_mode", "three", ["three", "octave", "third"]), F("--f_min_hz", 31.5),
        F("--f_max_hz", 16000.0), ("--legend_values", dict(action=BOOL, default=None)), F("--low_upper_hz", 250.0),
        F("--mid_center_hz", 1000.0), F("--mid_width_octaves", 2.0), F("--high_lower_hz", 4000.0),
        F("--transition_width_octaves", 1.0 / 6.0), ("--include_t20", dict(action="store_true")),
        ("--include_edt", dict(action="store_true")), MONO,
        ("--trim_to_peak", dict(action="store_true", default=True)), IGNORE, F("--edc_floor_db", -120.0),
        F("--fit_lower_limit_db", -80.0), I("--smoothing", 0, dest="edc_smoothing_window_samples"),
    ],
    "fr": [
        INPUT, OUTPUT, NO_SHOW, MONO, TRIM, IGNORE, DURATION, F("--magnitude_floor_db", -120.0), F("--f_min_hz", 20.0),
        F("--f_max_hz", 20000.0), I("--smoothing_log_bins", 0), I("--log_bins_per_octave", 96), NO_HANN,
    ],
    "filter": [
        INPUT, OUTPUT, NO_SHOW, MONO, TRIM, IGNORE, DURATION, F("--magnitude_floor_db", -120.0), F("--f_min_hz", 20.0),
        F("--f_max_hz", 20000.0), S("--phase_mode", "degrees", ["degrees", "radians"]),
        ("--no_unwrap_phase", dict(action="store_true")), NO_HANN,
    ],
    "spectrogram": [
        INPUT, OUTPUT, NO_SHOW, MONO, TRIM, IGNORE, DURATION, I("--n_fft", 4096), I("--hop_length", 512), NO_HANN,
        F("--floor_db", -120.0), F("--f_min_hz", 20.0), F("--f_max_hz", 20000.0), F("--dynamic_range_db", 90.0),
    ],
    "waterfall": [
        INPUT, OUTPUT, NO_SHOW, MONO, TRIM, IGNORE, DURATION, I("--n_fft", 4096), I("--hop_length", 512), NO_HANN,
        F("--f_min_hz", 20.0), F("--f_max_hz", 20000.0), S("--style", "3d", ["3d", "2d"]),
        S("--slice_mode", "auto", ["auto", "uniform_time", "uniform_frames"]), I("--num_slices", 18),
        F("--slice_spacing_seconds", 0.05), F("--start_time_seconds", 0.0), F("--end_time_seconds", None),
        S("--db_reference", "global_max", ["global_max", "slice_max"]), F("--dynamic_range_db", 80.0),
        F("--floor_db", -120.0), I("--smoothing_log_bins", 0), I("--log_bins_per_octave", 96), F("--elev_deg", 30.0),
        F("--azim_deg", -60.0), F("--ridge_offset_db", 6.0),
    ],
    "modalcloud": [
        INPUT, OUTPUT, NO_SHOW, MONO, TRIM, IGNORE, DURATION, I("--n_fft", 8192), I("--hop_length", 512), NO_HANN,
        F("--f_min_hz", 20.0), F("--f_max_hz", 20000.0), S("--metric", "t30", ["t30", "t20", "edt"]),
        I("--log_bins_per_octave", 24), I("--min_bins", 24), F("--fit_lower_limit_db", -80.0), I("--min_fit_points", 10),
        F("--min_peak_db_above_floor", 20.0), F("--floor_db", -120.0),
        ("--show_median_curve", dict(action=BOOL, default=True)), F("--median_octave_window", 0.25),
        F("--ylim_seconds_min", None), F("--ylim_seconds_max", None),
    ],
    "report": [
        INPUT, ("--output", dict(dest="output_basename", type=str, required=True)), MONO, TRIM,
        F("--ignore_leading_seconds", 0.0),
    ] + [(f"--{k}", dict(dest=f"run_{k}", action=BOOL, default=True))
         for k in ("ir", "decay", "rt60bands", "fr", "gd", "spectrogram", "waterfall", "diffusion", "modalcloud",
                   "echodensity")],
}
OUT_OF_SCOPE = ()


def build_parser() -> argparse.ArgumentParser:
    top = argparse.ArgumentParser(prog="analyse", description="Offline analysis tools for reverb outputs (plots, metrics).")
    sub = top.add_subparsers(dest="command_name", required=True, help="Analysis to run. Use: analyse <command> --help")
    for name, rows in COMMANDS.items():
        p = sub.add_parser(name)
        for flag, kw in rows:
            p.add_argument(flag, **{k: v for k, v in kw.items() if not (k == "choices" and v is None)})
    for name in OUT_OF_SCOPE:
        p = sub.add_parser(name, help="not part of the GPU-accelerated path")
        p.add_argument("rest", nargs=argparse.REMAINDER)
    return top


def parse_arguments(argv=None) -> argparse.Namespace:
    return build_parser().parse_args(argv)


def _basename(a) -> Optional[str]:
    return None if a.output_basename is None else str(Path(a.output_basename))


def main(argv=None) -> None:
    a = parse_arguments(argv)
    cmd = str(a.command_name)
    if cmd in OUT_OF_SCOPE:
        raise SystemExit(f"'{cmd}' is outside the GPU-accelerated path of audio_analysis_amd; use the reference for it.")

    if cmd == "ir":
        from .impulse_response import ImpulseResponseViewSettings, plot_ir_from_wav_file
        plot_ir_from_wav_file(str(a.input_wav_file_path),
                              ImpulseResponseViewSettings(early_window_seconds=float(a.early_window_seconds),
                                                          log_magnitude_floor_db=float(a.log_magnitude_floor_db),
                                                          use_mono_downmix=bool(a.use_mono_downmix)),
                              _basename(a), not bool(a.no_show))
    elif cmd == "decay":
        from .decay import DecayAnalysisSettings, DecayPlotSettings, plot_decay_from_wav_file, summarise_decay_results_text
        s = DecayAnalysisSettings(
            trim_to_peak=bool(a.trim_to_peak), ignore_leading_seconds=float(a.ignore_leading_seconds),
            edc_floor_db=float(a.edc_floor_db), fit_lower_limit_db=float(a.fit_lower_limit_db),
            edc_smoothing_window_samples=int(a.edc_smoothing_window_samples),
            use_mono_downmix_for_stereo=bool(a.use_mono_downmix), compute_edt=bool(a.compute_edt))
        r = plot_decay_from_wav_file(str(a.input_wav_file_path), s, DecayPlotSettings(), _basename(a), not bool(a.no_show))
        print(summarise_decay_results_text(r))
    elif cmd == "rt60bands":
        from .decay import DecayAnalysisSettings
        from .rt60bands import (Rt60BandsAnalysisSettings, Rt60BandsPlotSettings, plot_rt60_bands_from_wav_file,
                                summarise_rt60_bands_results_text)
        d = DecayAnalysisSettings(
            trim_to_peak=bool(a.trim_to_peak), ignore_leading_seconds=float(a.ignore_leading_seconds),
            edc_floor_db=float(a.edc_floor_db), fit_lower_limit_db=float(a.fit_lower_limit_db),
            edc_smoothing_window_samples=int(a.edc_smoothing_window_samples),
            use_mono_downmix_for_stereo=bool(a.use_mono_downmix), compute_edt=bool(a.include_edt))
        s = Rt60BandsAnalysisSettings(
            band_mode=str(a.band_mode), low_upper_hz=float(a.low_upper_hz), mid_center_hz=float(a.mid_center_hz),
            mid_width_octaves=float(a.mid_width_octaves), high_lower_hz=float(a.high_lower_hz), f_min_hz=float(a.f_min_hz),
            f_max_hz=float(a.f_max_hz), transition_width_octaves=float(a.transition_width_octaves),
            include_t20=bool(a.include_t20), include_edt=bool(a.include_edt), decay_settings=d)
        legend = (str(a.band_mode) == "three") if a.legend_values is None else bool(a.legend_values)
        r = plot_rt60_bands_from_wav_file(str(a.input_wav_file_path), s, Rt60BandsPlotSettings(legend_values=legend),
                                          _basename(a), not bool(a.no_show))
        print(summarise_rt60_bands_results_text(r, include_t20=s.include_t20, include_edt=s.include_edt))
    elif cmd == "fr":
        from .frequency_response import (FrequencyResponseAnalysisSettings, FrequencyResponsePlotSettings,
                                         plot_frequency_response_from_wav_file, summarise_frequency_response_results_text)
        s = FrequencyResponseAnalysisSettings(
            use_mono_downmix_for_stereo=bool(a.use_mono_downmix), trim_to_peak=bool(a.trim_to_peak),
            ignore_leading_seconds=float(a.ignore_leading_seconds), analysis_duration_seconds=a.analysis_duration_seconds,
            use_hann_window=not bool(a.no_hann_window), magnitude_floor_db=float(a.magnitude_floor_db),
            f_min_hz=float(a.f_min_hz), f_max_hz=float(a.f_max_hz), smoothing_log_bins=int(a.smoothing_log_bins),
            log_bins_per_octave=int(a.log_bins_per_octave))
        r = plot_frequency_response_from_wav_file(str(a.input_wav_file_path), s, FrequencyResponsePlotSettings(),
                                                  _basename(a), not bool(a.no_show))
        print(summarise_frequency_response_results_text(r))
    elif cmd == "filter":
        from .filterplot import (FilterAnalysisSettings, FilterPlotSettings, plot_filter_response_from_wav_file,
                                 summarise_filter_response_results_text)
        s = FilterAnalysisSettings(
            use_mono_downmix_for_stereo=bool(a.use_mono_downmix), trim_to_peak=bool(a.trim_to_peak),
            ignore_leading_seconds=float(a.ignore_leading_seconds), analysis_duration_seconds=a.analysis_duration_seconds,
            use_hann_window=not bool(a.no_hann_window), magnitude_floor_db=float(a.magnitude_floor_db),
            f_min_hz=float(a.f_min_hz), f_max_hz=float(a.f_max_hz), phase_mode=str(a.phase_mode),
            unwrap_phase=not bool(a.no_unwrap_phase))
        r = plot_filter_response_from_wav_file(str(a.input_wav_file_path), s, FilterPlotSettings(), _basename(a),
                                               not bool(a.no_show))
        print(summarise_filter_response_results_text(r))
    elif cmd == "spectrogram":
        from .spectrogram import (SpectrogramAnalysisSettings, SpectrogramPlotSettings, plot_spectrogram_from_wav_file,
                                  summarise_spectrogram_results_text)
        dyn = float(a.dynamic_range_db)
        s = SpectrogramAnalysisSettings(
            use_mono_downmix_for_stereo=bool(a.use_mono_downmix), trim_to_peak=bool(a.trim_to_peak),
            ignore_leading_seconds=float(a.ignore_leading_seconds), analysis_duration_seconds=a.analysis_duration_seconds,
            n_fft=int(a.n_fft), hop_length=int(a.hop_length), use_hann_window=not bool(a.no_hann_window),
            floor_db=float(a.floor_db), f_min_hz=float(a.f_min_hz), f_max_hz=float(a.f_max_hz),
            dynamic_range_db=None if dyn <= 0.0 else dyn)
        r = plot_spectrogram_from_wav_file(str(a.input_wav_file_path), s, SpectrogramPlotSettings(), _basename(a),
                                           not bool(a.no_show))
        print(summarise_spectrogram_results_text(r))
    elif cmd == "waterfall":
        from .waterfall import (WaterfallAnalysisSettings, WaterfallPlotSettings, plot_waterfall_from_wav_file,
                                summarise_waterfall_results_text)
        s = WaterfallAnalysisSettings(
            use_mono_downmix_for_stereo=bool(a.use_mono_downmix), trim_to_peak=bool(a.trim_to_peak),
            ignore_leading_seconds=float(a.ignore_leading_seconds), analysis_duration_seconds=a.analysis_duration_seconds,
            n_fft=int(a.n_fft), hop_length=int(a.hop_length), use_hann_window=not bool(a.no_hann_window),
            f_min_hz=float(a.f_min_hz), f_max_hz=float(a.f_max_hz), slice_mode=str(a.slice_mode),
            num_slices=int(a.num_slices), slice_spacing_seconds=float(a.slice_spacing_seconds),
            start_time_seconds=float(a.start_time_seconds), end_time_seconds=a.end_time_seconds,
            db_reference=str(a.db_reference), smoothing_log_bins=int(a.smoothing_log_bins),
            log_bins_per_octave=int(a.log_bins_per_octave), dynamic_range_db=float(a.dynamic_range_db),
            floor_db=float(a.floor_db))
        ps = WaterfallPlotSettings(style=str(a.style), elev_deg=float(a.elev_deg), azim_deg=float(a.azim_deg),
                                   ridge_offset_db=float(a.ridge_offset_db))
        r = plot_waterfall_from_wav_file(str(a.input_wav_file_path), s, ps, _basename(a), not bool(a.no_show))
        print(summarise_waterfall_results_text(r))
    elif cmd == "modalcloud":
        from .modalcloud import (ModalCloudAnalysisSettings, ModalCloudPlotSettings, plot_modal_cloud_from_wav_file,
                                 summarise_modal_cloud_results_text)
        s = ModalCloudAnalysisSettings(
            use_mono_downmix_for_stereo=bool(a.use_mono_downmix), trim_to_peak=bool(a.trim_to_peak),
            ignore_leading_seconds=float(a.ignore_leading_seconds), analysis_duration_seconds=a.analysis_duration_seconds,
            n_fft=int(a.n_fft), hop_length=int(a.hop_length), use_hann_window=not bool(a.no_hann_window),
            f_min_hz=float(a.f_min_hz), f_max_hz=float(a.f_max_hz), log_bins_per_octave=int(a.log_bins_per_octave),
            min_bins=int(a.min_bins), metric=str(a.metric), fit_lower_limit_db=float(a.fit_lower_limit_db),
            min_fit_points=int(a.min_fit_points), min_peak_db_above_floor=float(a.min_peak_db_above_floor),
            floor_db=float(a.floor_db))
        ylim = None
        if a.ylim_seconds_min is not None and a.ylim_seconds_max is not None:
            ylim = (float(a.ylim_seconds_min), float(a.ylim_seconds_max))
        ps = ModalCloudPlotSettings(show_median_curve=bool(a.show_median_curve),
                                    median_octave_window=float(a.median_octave_window), ylim_seconds=ylim)
        r = plot_modal_cloud_from_wav_file(str(a.input_wav_file_path), s, ps, _basename(a), not bool(a.no_show))
        print(summarise_modal_cloud_results_text(r))
    elif cmd == "zplane":
        from .zplane import (ZPlaneAnalysisSettings, ZPlanePlotSettings, plot_zplane_from_wav_file,
                             summarise_zplane_results_text)
        s = ZPlaneAnalysisSettings(
            use_mono_downmix_for_stereo=bool(a.use_mono_downmix_for_stereo), trim_to_peak=bool(a.trim_to_peak),
            ignore_leading_seconds=float(a.ignore_leading_seconds), analysis_duration_seconds=a.analysis_duration_seconds,
            ar_order=int(a.ar_order), derive_zeros=bool(a.derive_zeros), zero_order=int(a.zero_order),
            ridge_lambda=float(a.ridge_lambda))
        r = plot_zplane_from_wav_file(str(a.input_wav_file_path), s, ZPlanePlotSettings(limit_radius=float(a.limit_radius)),
                                      _basename(a), not bool(a.no_show))
        print(summarise_zplane_results_text(r))
    elif cmd == "groupdelay":
        from .group_delay import (GroupDelayAnalysisSettings, GroupDelayPlotSettings, plot_group_delay_from_wav_file,
                                  summarise_group_delay_results_text)
        s = GroupDelayAnalysisSettings(
            use_mono_downmix_for_stereo=bool(a.use_mono_downmix_for_stereo), trim_to_peak=bool(a.trim_to_peak),
            ignore_leading_seconds=float(a.ignore_leading_seconds), analysis_duration_seconds=a.analysis_duration_seconds,
            fft_size=a.fft_size, smoothing_bins=int(a.smoothing_bins), f_min_hz=float(a.f_min_hz),
            f_max_hz=float(a.f_max_hz))
        r = plot_group_delay_from_wav_file(str(a.input_wav_file_path), s, GroupDelayPlotSettings(), _basename(a),
                                           not bool(a.no_show))
        print(summarise_group_delay_results_text(r))
    elif cmd == "diffusion":
        from .diffusion import DiffusionAnalysisSettings, plot_diffusion_from_wav_file, summarise_diffusion_results_text
        s = DiffusionAnalysisSettings(
            use_mono_downmix_for_stereo=bool(a.use_mono_downmix), trim_to_peak=bool(a.trim_to_peak),
            ignore_leading_seconds=float(a.ignore_leading_seconds), window_seconds=float(a.window_seconds),
            hop_seconds=float(a.hop_seconds), max_lag_milliseconds=float(a.max_lag_milliseconds),
            echo_density_threshold_rms=float(a.echo_density_threshold_rms),
            echo_density_normalise_to_gaussian=bool(a.echo_density_normalise_to_gaussian))
        r = plot_diffusion_from_wav_file(str(a.input_wav_file_path), s, _basename(a), not bool(a.no_show))
        print(summarise_diffusion_results_text(r))
    elif cmd == "report":
        from .report import ReportSettings, run_report_from_wav_file
        rs = ReportSettings(
            common_use_mono_downmix_for_stereo=bool(a.use_mono_downmix), common_trim_to_peak=bool(a.trim_to_peak),
            common_ignore_leading_seconds=float(a.ignore_leading_seconds), run_impulse_response_plots=bool(a.run_ir),
            run_decay=bool(a.run_decay), run_rt60_bands=bool(a.run_rt60bands), run_frequency_response=bool(a.run_fr),
            run_group_delay=bool(a.run_gd), run_spectrogram=bool(a.run_spectrogram), run_waterfall=bool(a.run_waterfall),
            run_diffusion=bool(a.run_diffusion), run_modal_cloud=bool(a.run_modalcloud),
            run_echo_density=bool(a.run_echodensity))
        res = run_report_from_wav_file(str(a.input_wav_file_path), str(Path(a.output_basename)), rs)
        print(res.summary_markdown)
        print(f"Wrote: {res.summary_markdown_path}")
    elif cmd == "deconvolve":
        from .deconvolve import DeconvolveSettings, default_output_ir_path, deconvolve_from_wav_files
        out = a.output_ir_wav_file_path
        out = str(default_output_ir_path(a.recorded_wav_file_path)) if out is None else str(Path(out))
        s = DeconvolveSettings(regularization_relative=float(a.regularization_relative),
                               normalise_peak=bool(a.normalise_peak), target_peak=float(a.target_peak),
                               remove_dc=bool(a.remove_dc), output_length_mode=str(a.output_length_mode))
        r = deconvolve_from_wav_files(str(a.recorded_wav_file_path), str(a.sweep_wav_file_path), s, out)
        print(f"Wrote IR WAV: {out}")
        print(f"  sample_rate_hz={r.sample_rate_hz}")
        print(f"  channels={r.samples.shape[1]}")
        print(f"  length_seconds={r.samples.shape[0] / float(r.sample_rate_hz):.3f}")
    elif cmd == "bundle":
        from .bundle import BundleRunSettings, run_bundle_report
        import os
        # flags are the reference's; the two knobs of the batched path come from the environment
        index = run_bundle_report(str(a.bundle_root), settings=BundleRunSettings(
            reports_subdir=str(a.reports_subdir), taps_per_batch=int(os.environ.get("IRA_TAPS_PER_BATCH", "16")),
            plot_workers=int(os.environ.get("IRA_PLOT_WORKERS", "0"))))
        print(f"Wrote bundle report index: {index}")
    else:
        raise ValueError(f"Unknown command: {cmd}")


if __name__ == "__main__":
    main()
