"""
Pins oracle/ira_oracle.py against the golden vectors generated from the reference itself
(tests/golden/make_goldens.py).  Both sides ran the same NumPy, so float arrays are expected
bit-identical; where LAPACK driver choice could differ (closed-form vs lstsq is NOT used in the
oracle -- it calls lstsq like the reference) we still assert exact equality and only relax where
noted.
"""
import numpy as np
import pytest

from oracle import ira_oracle as O

SR = 48000


def _fit_list(f):
    return None if f is None else [f["range_hi"], f["range_lo"], f["start_t"], f["end_t"], f["slope"],
                                   f["intercept"], f["r2"], f["rt60"]]


def test_io_conversion(golden):
    g, c, _ = golden
    for k in ("i16", "i32", "f32"):
        np.testing.assert_array_equal(O.pcm_to_float32(g[f"io/{k}"]), g[f"io/{k}_f32"])
    st = np.stack([g["in/xs_l"], g["in/xs_r"]], axis=1)
    ch = O.analysis_channels(st, True)
    assert ch[0][0] == "mono"
    np.testing.assert_array_equal(ch[0][1], g["io/downmix"])
    assert [n for n, _ in O.analysis_channels(st, False)] == ["left", "right"]


@pytest.mark.parametrize("tag,inp", [("xa", "xa"), ("xa_edt", "xa"), ("xa_ign", "xa"), ("xa_notrim", "xa"),
                                     ("xa_smooth", "xa"), ("xb", "xb"), ("xb16", "xb16"), ("xc", "xc")])
def test_decay(golden, tag, inp):
    g, c, _ = golden
    case = c[f"{tag}/decay"]
    r = O.analyse_decay(g[f"in/{inp}"], SR, **case["kw"])
    assert r["start"] == case["start"]
    np.testing.assert_array_equal(r["edc_db"], g[f"{tag}/decay/edc_db"])
    assert r["early_10db"] == case["early"]
    assert set(r["fits"]) == set(case["fits"])
    for k, f in r["fits"].items():
        assert _fit_list(f) == case["fits"][k]


@pytest.mark.parametrize("key", ["xb/rt60bands/three", "xb/rt60bands/octave", "xb/rt60bands/third",
                                 "xd/rt60bands/three", "xd/rt60bands/octave", "xd/rt60bands/third",
                                 "xb16/rt60bands/third", "xc/rt60bands/octave", "xb_ign/rt60bands/three"])
def test_rt60_bands(golden, key):
    g, c, _ = golden
    case = c[key]
    tag, _, mode = key.split("/")
    kw = dict(case["kw"])
    ign = kw.pop("ignore_leading_seconds", 0.0)
    r = O.analyse_rt60_bands(g[f"in/{tag.split('_')[0]}"], SR, decay=dict(ignore_leading_seconds=ign),
                             band_mode=mode, **kw)
    got = [[b["name"], b["centre_hz"], b["kind"], b["low_edge_hz"], b["high_edge_hz"]] for b in r["bands"]]
    assert got == case["bands"]
    for name, m in r["metrics"].items():
        assert [m["t30"], m["t20"], m["edt"]] == case["metrics"][name], name


def test_band_counts_at_48k():
    assert len(O.band_definitions(SR, band_mode="octave")) == 9
    third = O.band_definitions(SR, band_mode="third")
    assert len(third) == 26 and third[-1]["name"] == "12699Hz"


def test_masks(golden):
    g, c, _ = golden
    n = int(g["mask/axis_n"][0])
    f = np.fft.rfftfreq(n, d=1.0 / float(SR)).astype(np.float32)
    np.testing.assert_array_equal(O.lowpass_mask(f, 250.0, 1 / 6, 24000.0), g["mask/lp250"])
    np.testing.assert_array_equal(O.highpass_mask(f, 4000.0, 1 / 6, 24000.0), g["mask/hp4000"])
    bp = O.band_mask(f, dict(kind="bandpass", low_edge_hz=500.0, high_edge_hz=2000.0), 1 / 6, 24000.0)
    np.testing.assert_array_equal(bp, g["mask/bp500_2000"])


def test_stft(golden):
    g, c, _ = golden
    xb = g["in/xb"]
    for nfft in (4096, 8192):
        cs = c[f"stft{nfft}"]
        seg = xb[cs["seg_start"] : cs["seg_start"] + cs["seg_len"]]
        t, f, m = O.stft_mag_db(seg, SR, nfft, 512, True, -120.0)
        np.testing.assert_array_equal(m, g[f"stft{nfft}/mag_db"])
        np.testing.assert_array_equal(t, g[f"stft{nfft}/time"])
        np.testing.assert_array_equal(f, g[f"stft{nfft}/freq"])
    _, _, m = O.stft_mag_db(g["in/xa"][240:6240], SR, 1024, 256, False, -100.0)
    np.testing.assert_array_equal(m, g["stft1024rect/mag_db"])


def test_spectrogram(golden):
    g, c, _ = golden
    r = O.analyse_spectrogram(g["in/xa"], SR)
    assert (r["start"], r["length"]) == (c["xa/spectrogram"]["start"], c["xa/spectrogram"]["length"])
    np.testing.assert_array_equal(r["magnitude_db"], g["xa/spectrogram/mag_db"])
    r = O.analyse_spectrogram(g["in/xb"], SR, ignore_leading_seconds=0.01, analysis_duration_seconds=0.5)
    cs = c["xb_sel/spectrogram"]
    assert (r["start"], r["length"], list(r["magnitude_db"].shape)) == (cs["start"], cs["length"], cs["shape"])
    np.testing.assert_array_equal(r["magnitude_db"][::7, ::3], g["xb_sel/spectrogram/mag_db_dec"])


def test_slice_selection(golden):
    _, c, _ = golden
    for T, modes in c["slice_select"].items():
        ft = O.frame_times(int(T), 512, SR)
        assert O.select_slice_frames(ft).tolist() == modes["auto"]
        assert O.select_slice_frames(ft, slice_mode="uniform_frames", num_slices=7).tolist() == modes["uniform_frames"]
        assert O.select_slice_frames(ft, slice_mode="uniform_time", slice_spacing_seconds=0.03,
                                     start_time_seconds=0.02, end_time_seconds=0.5).tolist() == modes["uniform_time"]
        assert O.select_slice_frames(ft, start_time_seconds=0.05, end_time_seconds=0.3,
                                     num_slices=9).tolist() == modes["auto_window"]


@pytest.mark.parametrize("tag,inp", [("xa", "xa"), ("xb", "xb"), ("xb_slice", "xb"), ("xb_smooth", "xb")])
def test_waterfall(golden, tag, inp):
    g, c, _ = golden
    cs = c[f"{tag}/waterfall"]
    r = O.analyse_waterfall(g[f"in/{inp}"], SR, **cs["kw"])
    assert (r["start"], r["length"]) == (cs["start"], cs["length"])
    np.testing.assert_array_equal(r["slice_times_seconds"], g[f"{tag}/waterfall/slice_times"])
    np.testing.assert_array_equal(r["frequency_hz"], g[f"{tag}/waterfall/freq"])
    np.testing.assert_array_equal(r["slice_rel_db"], g[f"{tag}/waterfall/rel_db"])


def test_modal_log_bins(golden):
    g, c, _ = golden
    np.testing.assert_array_equal(O.log_bin_edges(20.0, 20000.0, 24, 24), g["modal/edges"])
    seg = g["in/xb"].astype(np.float64)[243:].astype(np.float32)
    t, f, m = O.stft_mag_db(seg, SR, 8192, 512, True, -120.0)
    fm = (f >= 20.0) & (f <= 20000.0)
    cen, cur = O.aggregate_log_bins(f[fm], m[fm, :], g["modal/edges"])
    np.testing.assert_array_equal(cen, g["xb/modal/centres"])
    np.testing.assert_array_equal(cur, g["xb/modal/curves"])


@pytest.mark.parametrize("tag,inp", [("xb", "xb"), ("xb16", "xb16"), ("xb_t20", "xb"), ("xd_4096", "xd")])
def test_modal_cloud(golden, tag, inp):
    g, c, _ = golden
    cs = c[f"{tag}/modal"]
    r = O.analyse_modal_cloud(g[f"in/{inp}"], SR, **cs["kw"])
    assert (r["start"], r["length"], r["metric"]) == (cs["start"], cs["length"], cs["metric"])
    pts = np.array(r["points"], dtype=np.float64).reshape(-1, 3)
    np.testing.assert_array_equal(pts, g[f"{tag}/modal/points"])


@pytest.mark.parametrize("tag,inp", [("xa", "xa"), ("xc", "xc"), ("xd", "xd"), ("xa_sel", "xa"), ("xa_rect", "xa")])
def test_fr_and_filter(golden, tag, inp):
    g, c, _ = golden
    x = g[f"in/{inp}"]
    cs = c[f"{tag}/fr"]
    r = O.analyse_frequency_response(x, SR, **cs["kw"])
    assert (r["start"], r["length"], r["peak_hz"], r["centroid_hz"]) == (cs["start"], cs["length"], cs["peak"], cs["centroid"])
    np.testing.assert_array_equal(r["magnitude_db"], g[f"{tag}/fr/mag_db"])
    cs = c[f"{tag}/filter"]
    r = O.analyse_filter_response(x, SR, **cs["kw"])
    assert (r["start"], r["length"], r["peak_hz"], r["mag_1k_db"]) == (cs["start"], cs["length"], cs["peak"], cs["mag1k"])
    np.testing.assert_array_equal(r["magnitude_db"], g[f"{tag}/filter/mag_db"])
    np.testing.assert_array_equal(r["phase"], g[f"{tag}/filter/phase"])


def test_fr_smoothing_and_radians(golden):
    g, c, _ = golden
    r = O.analyse_frequency_response(g["in/xa"], SR, smoothing_log_bins=9)
    np.testing.assert_array_equal(r["magnitude_db"], g["xa_smooth/fr/mag_db"])
    assert (r["peak_hz"], r["centroid_hz"]) == (c["xa_smooth/fr"]["peak"], c["xa_smooth/fr"]["centroid"])
    r = O.analyse_filter_response(g["in/xa"], SR, phase_mode="radians", unwrap_phase=False)
    np.testing.assert_array_equal(r["phase"], g["xa_rad/filter/phase"])


@pytest.mark.parametrize("tag,inp", [("xa_p8", "xa"), ("xa_p64", "xa"), ("xa_p256", "xa"), ("xa_p64_ridge", "xa"),
                                     ("xe_p64", "xe"), ("xb16_p64", "xb16"), ("xc_p32", "xc")])
def test_zplane(golden, tag, inp):
    g, c, _ = golden
    cs = c["zplane"][tag]
    r = O.analyse_zplane(g[f"in/{inp}"], SR, ar_order=cs["order"], ridge_lambda=cs["ridge"], derive_zeros=True)
    assert r["start"] == cs["start"]
    np.testing.assert_array_equal(r["a"], g[f"{tag}/zplane/a"])
    np.testing.assert_array_equal(r["poles"], g[f"{tag}/zplane/poles"])
    np.testing.assert_array_equal(r["zeros"], g[f"{tag}/zplane/zeros"])
    assert (r["max_radius"], r["median_radius"], r["unstable"]) == (cs["max_r"], cs["med_r"], cs["unstable"])


# ---- section 8f rows: group delay and diffusion ----------------------------------------------------------------
GD_CASES = [("xa", "xa"), ("xc", "xc"), ("xb_sel", "xb"), ("xa_rect", "xa"), ("xa_fft", "xa"), ("xa_fft3", "xa"),
            ("xa_smooth", "xa"), ("xa_nounwrap", "xa")]
DIFF_CASES = [("xa", "xa"), ("xb", "xb"), ("xb16", "xb16"), ("xc_ign", "xc"), ("xa_short", "xa"), ("xa_thr", "xa"),
              ("xa_notrim", "xa"), ("xsil", "xsil")]


@pytest.mark.parametrize("tag,inp", GD_CASES)
def test_group_delay_oracle_bit_exact(golden, tag, inp):
    g, c, _ = golden
    cs = c[f"{tag}/gd"]
    o = O.analyse_group_delay(g[f"in/{inp}"], SR, **cs["kw"])
    assert (o["start"], o["length"]) == (cs["start"], cs["length"])
    np.testing.assert_array_equal(o["freq"], g[f"{tag}/gd/freq"])
    np.testing.assert_array_equal(o["gd"], g[f"{tag}/gd/gd"])
    assert O.group_delay_summary(["mono"], [o["gd"]]) == cs["summary"]


@pytest.mark.parametrize("tag,inp", DIFF_CASES)
def test_diffusion_oracle_bit_exact(golden, tag, inp):
    g, c, _ = golden
    cs = c[f"{tag}/diff"]
    o = O.analyse_diffusion(g[f"in/{inp}"], SR, **cs["kw"])
    assert o["time"].size == cs["frames"]
    for key, name in (("time", "time"), ("ac", "ac"), ("ed", "ed")):
        np.testing.assert_array_equal(o[key], g[f"{tag}/diff/{name}"])       # NaN == NaN under assert_array_equal
    assert O.diffusion_summary(["mono"], [o]) == cs["summary"]


def test_diffusion_stereo_oracle_bit_exact(golden):
    g, c, _ = golden
    pcm = g["report/stereo16/pcm"]
    chans = O.analysis_channels(O.pcm_to_float32(pcm), False)
    (nl, l), (nr, r) = chans
    st = O.diffusion_stereo(l, r, SR)
    np.testing.assert_array_equal(st["corr0"], g["report/stereo16/diff/corr0"])
    np.testing.assert_array_equal(st["iacc"], g["report/stereo16/diff/iacc"])
    series = [O.analyse_diffusion(l, SR), O.analyse_diffusion(r, SR)]
    for name, d in zip((nl, nr), series):
        np.testing.assert_array_equal(d["ac"], g[f"report/stereo16/diff/{name}/ac"])
        np.testing.assert_array_equal(d["ed"], g[f"report/stereo16/diff/{name}/ed"])
    assert O.diffusion_summary([nl, nr], series, st) == c["report"]["stereo16/diffusion"]["summary"]
    st2 = O.diffusion_stereo(l, r, SR, ignore_leading_seconds=0.02, max_lag_milliseconds=1.0)
    np.testing.assert_array_equal(st2["corr0"], g["report/stereo16/diff_ign/corr0"])
    np.testing.assert_array_equal(st2["iacc"], g["report/stereo16/diff_ign/iacc"])
    gd = [O.analyse_group_delay(ch, SR)["gd"] for ch in (l, r)]
    for name, v in zip((nl, nr), gd):
        np.testing.assert_array_equal(v, g[f"report/stereo16/gd/{name}"])
    assert O.group_delay_summary([nl, nr], gd) == c["report"]["stereo16/groupdelay"]["summary"]


# ---------------------------------------------------------------------------------------- section 8f rank 4
def test_deconvolve_oracle_reproduces_the_reference_bit_for_bit():
    from pathlib import Path
    z = np.load(Path(__file__).resolve().parent / "golden" / "deconvolve.npz")
    r, sw = z["stereo/recorded_pcm16"], z["stereo/sweep"]
    assert np.array_equal(O.deconvolve(r, sw), z["stereo/default"])
    assert np.array_equal(O.deconvolve(r, sw, normalise_peak=False, remove_dc=False), z["stereo/raw"])
    assert np.array_equal(O.deconvolve(r, sw, output_length_mode="full_fft", regularization_relative=1e-6,
                                       target_peak=0.5), z["stereo/full"])
    assert np.array_equal(O.deconvolve(z["mono/recorded_f32"], z["mono/sweep"]), z["mono/default"])
    sweep_file = O.sweep_downmix(O.pcm_to_float32(z["file/sweep_pcm16"]))
    assert np.array_equal(O.deconvolve(r, sweep_file), z["file/ir"])
    assert O.next_power_of_two(15000) == 16384 and O.next_power_of_two(16384) == 16384 and O.next_power_of_two(1) == 1
    with pytest.raises(ValueError):
        O.deconvolve(r[:5], sw)
    with pytest.raises(ValueError):
        O.deconvolve(r, sw, output_length_mode="nope")


def test_band_signals_and_masks_vs_reference():
    """a8 / a9 pinned directly (tests/golden/make_band_goldens.py ran the reference's mask builders and _apply_fft_mask):
    the oracle's masks and band-filtered signals are bit-identical for a Bluestein length and a smooth length."""
    from pathlib import Path
    gb = np.load(Path(__file__).resolve().parent / "golden" / "band_signals.npz")
    gi = np.load(Path(__file__).resolve().parent / "golden" / "goldens.npz")
    e = gb["third1k_edges"]
    bands = {"lp250": dict(kind="lowpass", low_edge_hz=None, high_edge_hz=250.0),
             "bp500_2000": dict(kind="bandpass", low_edge_hz=500.0, high_edge_hz=2000.0),
             "hp4000": dict(kind="highpass", low_edge_hz=4000.0, high_edge_hz=None),
             "third1k": dict(kind="bandpass", low_edge_hz=float(e[0]), high_edge_hz=float(e[1]))}
    for tag in ("xd", "xb"):
        x = gi[f"in/{tag}"]
        f = np.fft.rfftfreq(x.size, d=1.0 / 48000.0).astype(np.float32)
        for name, band in bands.items():
            m = O.band_mask(f, band, 1.0 / 6.0, 24000.0)
            assert m.dtype == np.float32
            np.testing.assert_array_equal(m, gb[f"mask/{tag}/{name}"])
            y = np.fft.irfft(np.fft.rfft(x.astype(np.float64)) * m, n=x.size).astype(np.float32)
            np.testing.assert_array_equal(y, gb[f"y/{tag}/{name}"])
