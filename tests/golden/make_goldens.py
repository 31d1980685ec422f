#!/usr/bin/env python3
"""
Generate the golden vectors under tests/golden/ by importing the REFERENCE implementation
(/root/reference/analyse) in the build container and running it on seeded inputs.

    PYTHONPATH=/root/reference PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg \
        python3 tests/golden/make_goldens.py

Outputs (committed): tests/golden/goldens.npz (arrays) and tests/golden/goldens.json (scalars,
strings, metadata incl. numpy/scipy versions).  Only inputs and expected outputs are stored --
no reference source text.  The reference cannot travel to the GPU box; these files can.

Inputs come from audio_analysis_amd.synth (SURVEY.md section 8d generator) and are stored in
the npz as well, so the fixtures do not depend on RNG stream stability.
"""

from __future__ import annotations

import json
import os
import sys
import tempfile
from dataclasses import replace
from pathlib import Path

import numpy as np
import scipy
from scipy.io import wavfile

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
# the REFERENCE's `analyse` package must win over this repo's import-name shim of the same name
sys.path[:] = [p for p in sys.path if p.rstrip("/") != "/root/reference"]
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

from audio_analysis_amd.synth import synth_ir  # noqa: E402

import analyse.decay as rdecay  # noqa: E402
import analyse.rt60bands as rbands  # noqa: E402
import analyse.spectrogram as rspec  # noqa: E402
import analyse.waterfall as rwf  # noqa: E402
import analyse.modalcloud as rmodal  # noqa: E402
import analyse.frequency_response as rfr  # noqa: E402
import analyse.filterplot as rfilt  # noqa: E402
import analyse.zplane as rz  # noqa: E402
import analyse.io as rio  # noqa: E402
import analyse.report as rreport  # noqa: E402
import analyse.group_delay as rgd  # noqa: E402
import analyse.diffusion as rdiff  # noqa: E402

SR = 48000
ARR = {}
META = {"numpy": np.__version__, "scipy": scipy.__version__, "cases": {}}


def put(key, value):
    ARR[key] = np.asarray(value)


def fit_to_list(f):
    if f is None:
        return None
    return [f.range_db[0], f.range_db[1], f.start_time_seconds, f.end_time_seconds,
            f.slope_db_per_second, f.intercept_db, f.r_squared, f.rt60_seconds]


def first_prime_at_least(n):
    def is_p(k):
        if k < 2:
            return False
        i = 2
        while i * i <= k:
            if k % i == 0:
                return False
            i += 1
        return True
    while not is_p(n):
        n += 1
    return n


def decay_case(tag, x, **kw):
    s = rdecay.DecayAnalysisSettings(**kw)
    r = rdecay.analyse_decay_for_channel(x, SR, "mono", s)
    put(f"{tag}/decay/edc_db", r.edc_db)
    META["cases"][f"{tag}/decay"] = dict(
        kw=kw, start=r.analysis_start_sample_index, n=int(r.edc_db.size),
        early=r.early_decay_10db_time_seconds,
        fits={k: fit_to_list(v) for k, v in r.fits.items()},
        summary=rdecay.summarise_decay_results_text([r]),
    )


def bands_case(tag, x, mode, **kw):
    s = rbands.Rt60BandsAnalysisSettings(band_mode=mode, **kw)
    r = rbands.analyse_rt60_bands_for_channel(x, SR, "mono", s)
    META["cases"][f"{tag}/rt60bands/{mode}"] = dict(
        kw={k: v for k, v in kw.items() if k != "decay_settings"},
        bands=[[b.name, b.centre_hz, b.kind, b.low_edge_hz, b.high_edge_hz] for b in r.band_definitions],
        metrics={k: [m.rt60_t30_seconds, m.rt60_t20_seconds, m.edt_seconds]
                 for k, m in r.band_metrics_by_name.items()},
        summary=rbands.summarise_rt60_bands_results_text([r], s.include_t20, s.include_edt),
    )


def main():
    # ---------------- inputs -----------------------------------------------------------------
    xa = synth_ir(0, 0, 24000, rt60_seconds=0.15)                      # 0.5 s, fast decay
    xb = synth_ir(1, 0, 48000, rt60_seconds=0.32)                      # 1 s
    xb16 = synth_ir(1, 0, 48000, rt60_seconds=0.32, pcm16_round_trip=True)
    d_c = 240 + 2
    n_c = d_c + first_prime_at_least(23000)                            # trimmed length is prime
    xc = synth_ir(2, 0, n_c, rt60_seconds=0.2)
    n_d = 240 + 3 + 2 * 37 * 311                                       # large prime factor, even
    xd = synth_ir(3, 0, n_d, rt60_seconds=0.25)
    xe = synth_ir(4, 0, 24000, rt60_seconds=0.2, lowpass_pole=0.6)     # coloured -> worse AR conditioning
    xs_l = synth_ir(5, 0, 36000, rt60_seconds=0.25)
    xs_r = synth_ir(5, 1, 36000, rt60_seconds=0.25)
    for k, v in dict(xa=xa, xb=xb, xb16=xb16, xc=xc, xd=xd, xe=xe, xs_l=xs_l, xs_r=xs_r).items():
        put(f"in/{k}", v)

    # ---------------- a1: conversion ----------------------------------------------------------
    rng = np.random.default_rng(7)
    i16 = rng.integers(-32768, 32768, size=257, dtype=np.int16)
    i16[:4] = [-32768, 32767, 0, -1]
    i32 = rng.integers(-2**31, 2**31, size=257, dtype=np.int64).astype(np.int32)
    f32 = (rng.standard_normal(257) * 0.8).astype(np.float32)
    put("io/i16", i16); put("io/i16_f32", rio.convert_wav_samples_to_float32(i16))
    put("io/i32", i32); put("io/i32_f32", rio.convert_wav_samples_to_float32(i32))
    put("io/f32", f32); put("io/f32_f32", rio.convert_wav_samples_to_float32(f32))
    st = np.stack([xs_l, xs_r], axis=1)
    la = rio.LoadedAudio(samples=st, sample_rate_hz=SR, file_path=Path("x.wav"))
    put("io/downmix", rio.get_analysis_channels(la, True)[0][1])

    # ---------------- a3-a6: decay ------------------------------------------------------------
    decay_case("xa", xa)
    decay_case("xa_edt", xa, compute_edt=True)
    decay_case("xa_ign", xa, compute_edt=True, ignore_leading_seconds=0.0105)
    decay_case("xa_notrim", xa, trim_to_peak=False)
    decay_case("xa_smooth", xa, edc_smoothing_window_samples=33)
    decay_case("xb", xb, compute_edt=True)
    decay_case("xb16", xb16, compute_edt=True)
    decay_case("xc", xc, compute_edt=True)

    # ---------------- a7-a10: rt60 bands --------------------------------------------------------
    for mode in ("three", "octave", "third"):
        bands_case("xb", xb, mode, include_t20=True, include_edt=True)
        bands_case("xd", xd, mode)
    bands_case("xb16", xb16, "third")
    bands_case("xc", xc, "octave", include_t20=True)
    bands_case("xb_ign", xb, "three", include_t20=True,
               decay_settings=rdecay.DecayAnalysisSettings(ignore_leading_seconds=0.004))
    META["cases"]["xb_ign/rt60bands/three"]["kw"]["ignore_leading_seconds"] = 0.004
    # masks themselves (float32 arithmetic) on one awkward axis
    fr_axis = np.fft.rfftfreq(n_d, d=1.0 / float(SR)).astype(np.float32)
    put("mask/axis_n", np.array([n_d]))
    put("mask/lp250", rbands._make_lowpass_mask(fr_axis, 250.0, 1.0 / 6.0, 24000.0))
    put("mask/hp4000", rbands._make_highpass_mask(fr_axis, 4000.0, 1.0 / 6.0, 24000.0))
    put("mask/bp500_2000", rbands._make_bandpass_mask(fr_axis, 500.0, 2000.0, 1.0 / 6.0, 24000.0))

    # ---------------- a11-a12: STFT / spectrogram --------------------------------------------------
    for nfft in (4096, 8192):
        seg = xb[243 : 243 + nfft + 7 * 512 + 100]
        t, f, m = rspec._compute_stft_magnitude_db(seg, SR, nfft, 512, True, -120.0)
        put(f"stft{nfft}/mag_db", m); put(f"stft{nfft}/time", t); put(f"stft{nfft}/freq", f)
        META["cases"][f"stft{nfft}"] = dict(seg_start=243, seg_len=int(seg.size))
    t, f, m = rspec._compute_stft_magnitude_db(xa[240:240 + 6000], SR, 1024, 256, False, -100.0)
    put("stft1024rect/mag_db", m)
    r = rspec.analyse_spectrogram_for_channel(xa, SR, "mono", rspec.SpectrogramAnalysisSettings())
    put("xa/spectrogram/mag_db", r.magnitude_db)
    META["cases"]["xa/spectrogram"] = dict(start=r.analysis_start_sample_index, length=r.analysis_length_samples,
                                           summary=rspec.summarise_spectrogram_results_text([r]))
    r = rspec.analyse_spectrogram_for_channel(
        xb, SR, "mono", rspec.SpectrogramAnalysisSettings(ignore_leading_seconds=0.01, analysis_duration_seconds=0.5))
    put("xb_sel/spectrogram/mag_db_dec", r.magnitude_db[::7, ::3])
    META["cases"]["xb_sel/spectrogram"] = dict(start=r.analysis_start_sample_index, length=r.analysis_length_samples,
                                               shape=list(r.magnitude_db.shape),
                                               summary=rspec.summarise_spectrogram_results_text([r]))

    # ---------------- a13-a14: waterfall ---------------------------------------------------------
    sel = {}
    for T in (2, 3, 17, 18, 19, 40, 179, 929, 2224):
        ft = (np.arange(T, dtype=np.float32) * 512.0 / 48000.0).astype(np.float32)
        sel[str(T)] = dict(
            auto=rwf._select_slice_frame_indices(ft, rwf.WaterfallAnalysisSettings()).tolist(),
            uniform_frames=rwf._select_slice_frame_indices(
                ft, rwf.WaterfallAnalysisSettings(slice_mode="uniform_frames", num_slices=7)).tolist(),
            uniform_time=rwf._select_slice_frame_indices(
                ft, rwf.WaterfallAnalysisSettings(slice_mode="uniform_time", slice_spacing_seconds=0.03,
                                                  start_time_seconds=0.02, end_time_seconds=0.5)).tolist(),
            auto_window=rwf._select_slice_frame_indices(
                ft, rwf.WaterfallAnalysisSettings(start_time_seconds=0.05, end_time_seconds=0.3, num_slices=9)).tolist(),
        )
    META["cases"]["slice_select"] = sel
    for tag, x, kw in (("xa", xa, {}), ("xb", xb, {}), ("xb_slice", xb, dict(db_reference="slice_max", dynamic_range_db=60.0)),
                       ("xb_smooth", xb, dict(smoothing_log_bins=9))):
        r = rwf.analyse_waterfall_for_channel(x, SR, "mono", rwf.WaterfallAnalysisSettings(**kw))
        put(f"{tag}/waterfall/rel_db", r.slice_magnitude_rel_db)
        put(f"{tag}/waterfall/slice_times", r.slice_times_seconds)
        put(f"{tag}/waterfall/freq", r.frequency_hz)
        META["cases"][f"{tag}/waterfall"] = dict(kw=kw, start=r.analysis_start_sample_index,
                                                 length=r.analysis_length_samples,
                                                 summary=rwf.summarise_waterfall_results_text([r]))

    # ---------------- a15-a16: modal cloud -------------------------------------------------------
    put("modal/edges", rmodal._build_log_bins(20.0, 20000.0, 24, 24))
    for tag, x, kw in (("xb", xb, {}), ("xb16", xb16, {}), ("xb_t20", xb, dict(metric="t20")),
                       ("xd_4096", xd, dict(n_fft=4096, hop_length=256, metric="edt"))):
        s = rmodal.ModalCloudAnalysisSettings(**kw)
        r = rmodal.analyse_modal_cloud_for_channel(x, SR, "mono", s)
        put(f"{tag}/modal/points", np.array([[p.centre_hz, p.rt60_seconds, p.r_squared] for p in r.points],
                                            dtype=np.float64).reshape(-1, 3))
        META["cases"][f"{tag}/modal"] = dict(kw=kw, start=r.analysis_start_sample_index,
                                             length=r.analysis_length_samples, metric=r.metric,
                                             summary=rmodal.summarise_modal_cloud_results_text([r]))
    # intermediate log-bin curves for xb (pins a15 separately from a16)
    seg = xb.astype(np.float64)[243:].astype(np.float32)
    t, f, m = rmodal._compute_stft_magnitude_db(seg, SR, 8192, 512, True, -120.0)
    fm = (f >= 20.0) & (f <= 20000.0)
    c, curves = rmodal._aggregate_to_log_bins(f[fm], m[fm, :], rmodal._build_log_bins(20.0, 20000.0, 24, 24))
    put("xb/modal/centres", c); put("xb/modal/curves", curves)

    # ---------------- a17-a18: fr / filter -------------------------------------------------------
    for tag, x, kw in (("xa", xa, {}), ("xc", xc, {}), ("xd", xd, {}),
                       ("xa_sel", xa, dict(ignore_leading_seconds=0.002, analysis_duration_seconds=0.1)),
                       ("xa_rect", xa, dict(use_hann_window=False))):
        r = rfr.analyse_frequency_response_for_channel(x, SR, "mono", rfr.FrequencyResponseAnalysisSettings(**kw))
        put(f"{tag}/fr/mag_db", r.magnitude_db)
        META["cases"][f"{tag}/fr"] = dict(kw=kw, start=r.analysis_start_sample_index, length=r.analysis_length_samples,
                                          peak=r.peak_frequency_hz, centroid=r.spectral_centroid_hz,
                                          summary=rfr.summarise_frequency_response_results_text([r]))
        r = rfilt.analyse_filter_response_for_channel(x, SR, "mono", rfilt.FilterAnalysisSettings(**kw))
        put(f"{tag}/filter/mag_db", r.magnitude_db); put(f"{tag}/filter/phase", r.phase_response)
        META["cases"][f"{tag}/filter"] = dict(kw=kw, start=r.analysis_start_sample_index,
                                              length=r.analysis_length_samples, peak=r.peak_frequency_hz,
                                              mag1k=r.magnitude_at_1khz_db,
                                              summary=rfilt.summarise_filter_response_results_text([r]))
    r = rfr.analyse_frequency_response_for_channel(xa, SR, "mono", rfr.FrequencyResponseAnalysisSettings(smoothing_log_bins=9))
    put("xa_smooth/fr/mag_db", r.magnitude_db)
    META["cases"]["xa_smooth/fr"] = dict(peak=r.peak_frequency_hz, centroid=r.spectral_centroid_hz)
    r = rfilt.analyse_filter_response_for_channel(xa, SR, "mono", rfilt.FilterAnalysisSettings(phase_mode="radians", unwrap_phase=False))
    put("xa_rad/filter/phase", r.phase_response)

    # ---------------- a19-a22: zplane ------------------------------------------------------------
    zc = {}
    for tag, x, p, ridge in (("xa_p8", xa, 8, 0.0), ("xa_p64", xa, 64, 0.0), ("xa_p256", xa, 256, 0.0),
                             ("xa_p64_ridge", xa, 64, 1e-6), ("xe_p64", xe, 64, 0.0), ("xb16_p64", xb16, 64, 0.0),
                             ("xc_p32", xc, 32, 0.0)):
        start = int(np.argmax(np.abs(x)))
        seg = x[start:].astype(np.float64)
        seg = seg / float(np.max(np.abs(seg)))
        a = rz._fit_ar_least_squares(seg, p, ridge)
        poles = rz._roots_from_poly_descending(a)
        b = rz._derive_fir_numerator_from_ar(a, seg, 64)
        zeros = rz._roots_from_poly_descending(b)
        put(f"{tag}/zplane/a", a); put(f"{tag}/zplane/poles", poles)
        put(f"{tag}/zplane/b", b); put(f"{tag}/zplane/zeros", zeros)
        rad = np.abs(poles)
        A = np.lib.stride_tricks.sliding_window_view(seg, p + 1)[:, ::-1][:, 1:]
        sv = np.linalg.svd(A, compute_uv=False)
        res = rz.ChannelZPlaneResult("mono", SR, poles, None)
        zc[tag] = dict(order=p, ridge=ridge, start=start, max_r=float(rad.max()), med_r=float(np.median(rad)),
                       unstable=int((rad >= 1.0).sum()), cond_A=float(sv[0] / sv[-1]),
                       summary=rz.summarise_zplane_results_text([res]))
    META["cases"]["zplane"] = zc

    # ---------------- section 8f: group delay ----------------------------------------------------
    # the reference has no per-channel entry for group delay (the selection lives in its plot function,
    # group_delay.py:159-171); the segment bounds are recomputed here and stored with the case
    def gd_case(tag, x, **kw):
        st = rgd.GroupDelayAnalysisSettings(**kw)
        start = int(np.argmax(np.abs(x))) if st.trim_to_peak else 0
        start += int(round(float(st.ignore_leading_seconds) * SR))
        start = max(0, min(start, len(x)))
        if st.analysis_duration_seconds is None:
            seg = x[start:]
        else:
            seg = x[start : start + max(1, int(round(float(st.analysis_duration_seconds) * SR)))]
        r = rgd._compute_group_delay_from_ir(seg, SR, st)
        put(f"{tag}/gd/freq", r.frequency_hz); put(f"{tag}/gd/gd", r.group_delay_samples)
        res = rgd.ChannelGroupDelayResult("mono", SR, r.frequency_hz, r.group_delay_samples)
        META["cases"][f"{tag}/gd"] = dict(kw=kw, start=start, length=int(seg.size),
                                          summary=rgd.summarise_group_delay_results_text([res]))

    gd_case("xa", xa)
    gd_case("xc", xc)
    gd_case("xb_sel", xb, ignore_leading_seconds=0.002, analysis_duration_seconds=0.2)
    gd_case("xa_rect", xa, use_hann_window=False)
    gd_case("xa_fft", xa, fft_size=16384)                 # n_fft < segment length: rfft truncates the windowed segment
    gd_case("xa_fft3", xa, fft_size=3 * 5 * 7 * 256)      # not a power of two
    gd_case("xa_smooth", xa, smoothing_bins=9)
    gd_case("xa_nounwrap", xa, unwrap_phase=False, f_min_hz=100.0, f_max_hz=5000.0)

    # ---------------- section 8f: diffusion --------------------------------------------------------
    def diff_case(tag, x, **kw):
        st = rdiff.DiffusionAnalysisSettings(**kw)
        r = rdiff.analyse_diffusion_for_channel(x, SR, "mono", st)
        put(f"{tag}/diff/time", r.series.time_seconds); put(f"{tag}/diff/ac", r.series.max_abs_autocorr)
        put(f"{tag}/diff/ed", r.series.echo_density)
        META["cases"][f"{tag}/diff"] = dict(kw=kw, frames=int(r.series.time_seconds.size),
                                            summary=rdiff.summarise_diffusion_results_text([r]))

    diff_case("xa", xa)
    diff_case("xb", xb)
    diff_case("xb16", xb16)
    diff_case("xc_ign", xc, ignore_leading_seconds=0.0123)
    diff_case("xa_short", xa, window_seconds=0.012, hop_seconds=0.004, max_lag_milliseconds=20.0)   # lag > window-2
    diff_case("xa_thr", xa, echo_density_threshold_rms=1.5, echo_density_normalise_to_gaussian=False)
    diff_case("xa_notrim", xa, trim_to_peak=False, hop_seconds=0.025)
    xsil = xa.copy(); xsil[6000:] = 0.0                    # digital silence -> NaN frames
    put("in/xsil", xsil)
    diff_case("xsil", xsil)

    # ---------------- a23: report markdown + full-file command path ---------------------------------
    tmp = Path(tempfile.mkdtemp(prefix="goldens_"))
    rep = {}
    files = {
        "stereo16": (np.stack([xs_l, xs_r], axis=1), np.int16),
        "mono16": (xs_l[:, None], np.int16),
        "stereof32": (np.stack([xs_l, xs_r], axis=1), np.float32),
    }
    for name, (data, dt) in files.items():
        wav = tmp / f"{name}.wav"
        if dt == np.int16:
            q = (data * np.float32(32767.0)).astype(np.int16)
            wavfile.write(str(wav), SR, q if q.shape[1] > 1 else q[:, 0])
            put(f"report/{name}/pcm", q)
        else:
            wavfile.write(str(wav), SR, data.astype(np.float32))
            put(f"report/{name}/pcm", data.astype(np.float32))
        for variant, st in (("default", rreport.ReportSettings(run_impulse_response_plots=False, run_group_delay=False,
                                                               run_diffusion=False)),
                            ("monomix", rreport.ReportSettings(run_impulse_response_plots=False, run_group_delay=False,
                                                               run_diffusion=False,
                                                               common_use_mono_downmix_for_stereo=True,
                                                               common_ignore_leading_seconds=0.003)),
                            ("full", rreport.ReportSettings(run_impulse_response_plots=False)),
                            ("fullmix", rreport.ReportSettings(run_impulse_response_plots=False,
                                                               common_use_mono_downmix_for_stereo=True,
                                                               common_ignore_leading_seconds=0.003))):
            if name == "mono16" and variant in ("monomix", "fullmix"):
                continue
            if name == "stereof32" and variant in ("full", "fullmix"):
                continue
            out = tmp / f"out_{name}_{variant}" / "rep"
            res = rreport.run_report_from_wav_file(wav, out, st)
            rep[f"{name}/{variant}"] = dict(
                markdown=res.summary_markdown.replace(str(wav), "{WAV}"),
            )
    # zplane command numerics on the stereo file (plot function is the only public entry, zplane.py:176)
    zres = rz.plot_zplane_from_wav_file(str(tmp / "stereo16.wav"), rz.ZPlaneAnalysisSettings(ar_order=32),
                                        rz.ZPlanePlotSettings(), output_basename=str(tmp / "z"), show_interactive=False)
    rep["stereo16/zplane32"] = dict(summary=rz.summarise_zplane_results_text(zres))
    for r_ in zres:
        put(f"report/stereo16/zplane32/{r_.channel_name}", r_.poles)
    dres = rdiff.analyse_diffusion_from_wav_file(tmp / "stereo16.wav", rdiff.DiffusionAnalysisSettings())
    rep["stereo16/diffusion"] = dict(summary=rdiff.summarise_diffusion_results_text(dres),
                                     names=[r_.channel_name for r_ in dres])
    for r_ in dres:
        put(f"report/stereo16/diff/{r_.channel_name}/ac", r_.series.max_abs_autocorr)
        put(f"report/stereo16/diff/{r_.channel_name}/ed", r_.series.echo_density)
    put("report/stereo16/diff/corr0", dres[0].series.corr0); put("report/stereo16/diff/iacc", dres[0].series.iacc_max)
    put("report/stereo16/diff/time", dres[0].series.time_seconds)
    dres = rdiff.analyse_diffusion_from_wav_file(tmp / "stereo16.wav",
                                                 rdiff.DiffusionAnalysisSettings(ignore_leading_seconds=0.02, max_lag_milliseconds=1.0))
    put("report/stereo16/diff_ign/corr0", dres[0].series.corr0); put("report/stereo16/diff_ign/iacc", dres[0].series.iacc_max)
    gres = rgd.plot_group_delay_from_wav_file(str(tmp / "stereo16.wav"), rgd.GroupDelayAnalysisSettings(),
                                              rgd.GroupDelayPlotSettings(), output_basename=str(tmp / "g"), show_interactive=False)
    rep["stereo16/groupdelay"] = dict(summary=rgd.summarise_group_delay_results_text(gres))
    for r_ in gres:
        put(f"report/stereo16/gd/{r_.channel_name}", r_.group_delay_samples)
    fres = rfilt.analyse_filter_response_from_wav_file(tmp / "stereo16.wav", rfilt.FilterAnalysisSettings())
    rep["stereo16/filter"] = dict(summary=rfilt.summarise_filter_response_results_text(fres))
    META["cases"]["report"] = rep

    # ---------------- CLI surface (flag spellings, dests, defaults) as data ---------------------------------
    import argparse
    import analyse.cli as rcli
    grabbed = {}
    real_parse = argparse.ArgumentParser.parse_args

    def _grab(self, *a, **k):
        grabbed["parser"] = self
        raise SystemExit

    argparse.ArgumentParser.parse_args = _grab
    try:
        rcli.parse_arguments()
    except SystemExit:
        pass
    finally:
        argparse.ArgumentParser.parse_args = real_parse
    sub = [a for a in grabbed["parser"]._actions if isinstance(a, argparse._SubParsersAction)][0]
    surface = {}
    for name, sp in sub.choices.items():
        rows = []
        for act in sp._actions:
            if act.dest == "help":
                continue
            rows.append(dict(flags=list(act.option_strings), dest=act.dest, kind=type(act).__name__,
                             type=getattr(act.type, "__name__", None), default=act.default, required=bool(act.required),
                             choices=list(act.choices) if act.choices else None))
        surface[name] = rows
    META["cases"]["cli_surface"] = surface

    np.savez_compressed(HERE / "goldens.npz", **ARR)
    (HERE / "goldens.json").write_text(json.dumps(META, indent=1, sort_keys=True))
    tot = (HERE / "goldens.npz").stat().st_size + (HERE / "goldens.json").stat().st_size
    print(f"wrote {len(ARR)} arrays, {tot/1e6:.2f} MB")


if __name__ == "__main__":
    main()
