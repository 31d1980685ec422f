"""GPU parity: waterfall slices and modal cloud vs golden vectors / oracle."""
import numpy as np
import pytest

from oracle import ira_oracle as O

pytestmark = pytest.mark.gpu
SR = 48000


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def test_slice_selection_exact(golden):
    from audio_analysis_amd.analyse import waterfall as wf
    _, c, _ = golden
    for T, modes in c["slice_select"].items():
        ft = (np.arange(int(T), dtype=np.float32) * 512.0 / 48000.0).astype(np.float32)
        assert wf._select_slice_frame_indices(ft, wf.WaterfallAnalysisSettings()).tolist() == modes["auto"]
        assert wf._select_slice_frame_indices(
            ft, wf.WaterfallAnalysisSettings(slice_mode="uniform_frames", num_slices=7)).tolist() == modes["uniform_frames"]
        assert wf._select_slice_frame_indices(
            ft, wf.WaterfallAnalysisSettings(slice_mode="uniform_time", slice_spacing_seconds=0.03,
                                             start_time_seconds=0.02, end_time_seconds=0.5)).tolist() == modes["uniform_time"]
        assert wf._select_slice_frame_indices(
            ft, wf.WaterfallAnalysisSettings(start_time_seconds=0.05, end_time_seconds=0.3,
                                             num_slices=9)).tolist() == modes["auto_window"]


@pytest.mark.parametrize("tag,inp", [("xa", "xa"), ("xb", "xb"), ("xb_slice", "xb"), ("xb_smooth", "xb")])
def test_waterfall_vs_golden(golden, tag, inp, monkeypatch):
    from audio_analysis_amd.analyse import waterfall as wf
    g, c, _ = golden
    monkeypatch.setattr(wf, "smooth_log_frequency", lambda *a, **k: pytest.fail("host smoothing ran"))
    cs = c[f"{tag}/waterfall"]
    r = wf.analyse_waterfall_for_channel(g[f"in/{inp}"], SR, "mono", wf.WaterfallAnalysisSettings(**cs["kw"]))
    assert (r.analysis_start_sample_index, r.analysis_length_samples) == (cs["start"], cs["length"])
    np.testing.assert_array_equal(r.slice_times_seconds, g[f"{tag}/waterfall/slice_times"])     # frame pick: exact
    np.testing.assert_array_equal(r.frequency_hz, g[f"{tag}/waterfall/freq"])
    ref = g[f"{tag}/waterfall/rel_db"]
    assert r.slice_magnitude_rel_db.shape == ref.shape
    np.testing.assert_allclose(r.slice_magnitude_rel_db, ref, rtol=0, atol=3e-5)   # f64 butterflies
    assert wf.summarise_waterfall_results_text([r]) == cs["summary"]


def test_modal_log_bins_exact(golden):
    from audio_analysis_amd.analyse import modalcloud as mc
    g, c, _ = golden
    np.testing.assert_array_equal(mc._build_log_bins(20.0, 20000.0, 24, 24), g["modal/edges"])
    freq = np.fft.rfftfreq(8192, 1 / 48000.0).astype(np.float32)
    sel = freq[(freq >= 20.0) & (freq <= 20000.0)]
    cen, first, count = mc.log_bin_rows(sel, g["modal/edges"])
    cen_o, first_o, count_o = O.log_bin_membership(sel, g["modal/edges"])
    np.testing.assert_array_equal(cen, cen_o)
    np.testing.assert_array_equal(count, count_o)
    np.testing.assert_array_equal(first[count > 0], first_o[count_o > 0])


def test_modal_curves_vs_golden(golden):
    """a15 separately from a16: the (240, T) log-bin curves."""
    from audio_analysis_amd.analyse import modalcloud as mc
    from audio_analysis_amd.engine import get_engine
    g, c, _ = golden
    eng = get_engine()
    seg = g["in/xb"][243:]
    b = eng.upload([seg])
    nfr = np.array([1 + (seg.size - 8192) // 512], dtype=np.int32)
    mag, off, cols = eng.stft_mag_db(b.x, b.off, nfr, 8192, 512, True, -120.0, 64)
    freq = np.fft.rfftfreq(8192, 1 / 48000.0).astype(np.float32)
    rows = np.nonzero((freq >= 20.0) & (freq <= 20000.0))[0]
    cen, first, count = mc.log_bin_rows(freq[rows], g["modal/edges"])
    cur, _ = eng.logbin_aggregate(mag, off, cols, int(rows[0]), first, count)
    got = cur.cpu().numpy()[: cen.size * nfr[0]].reshape(cen.size, nfr[0])
    ref = g["xb/modal/curves"]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(got[ok], ref[ok], rtol=0, atol=2e-5)
    assert np.mean(got[ok] == ref[ok]) > 0.99


def test_frame_major_f64_stft_and_logbin_vs_golden(golden):
    """v4 kernel (f64 / 8192, frame-major) against the golden STFT and log-bin curves, and bit-equal to the (F, T) path."""
    from audio_analysis_amd.analyse import modalcloud as mc
    from audio_analysis_amd.engine import get_engine
    g, c, _ = golden
    eng = get_engine()
    assert eng.stft_frame_major_ok(8192, 64)
    cs = c["stft8192"]
    seg = g["in/xb"][cs["seg_start"] : cs["seg_start"] + cs["seg_len"]]
    b = eng.upload([seg])
    nfr = np.array([1 + (seg.size - 8192) // 512], dtype=np.int32)
    ft, _, _ = eng.stft_mag_db(b.x, b.off, nfr, 8192, 512, True, -120.0, 64)
    tf, _, _ = eng.stft_mag_db(b.x, b.off, nfr, 8192, 512, True, -120.0, 64, frame_major=True)
    a = ft.cpu().numpy()[: 4097 * nfr[0]].reshape(4097, nfr[0])
    t = tf.cpu().numpy()[: 4097 * nfr[0]].reshape(nfr[0], 4097).T
    ref = g["stft8192/mag_db"]
    np.testing.assert_allclose(t, ref, rtol=0, atol=2e-5)
    assert np.mean(t == ref) > 0.99
    assert np.mean(t == a) > 0.999 and np.max(np.abs(t - a)) < 2e-5
    # log-bin curves from the frame-major matrix
    seg = g["in/xb"][243:]
    b = eng.upload([seg])
    nfr = np.array([1 + (seg.size - 8192) // 512], dtype=np.int32)
    mag, off, cols = eng.stft_mag_db(b.x, b.off, nfr, 8192, 512, True, -120.0, 64, frame_major=True)
    freq = np.fft.rfftfreq(8192, 1 / 48000.0).astype(np.float32)
    rows = np.nonzero((freq >= 20.0) & (freq <= 20000.0))[0]
    cen, first, count = mc.log_bin_rows(freq[rows], g["modal/edges"])
    cur, _ = eng.logbin_aggregate(mag, off, cols, int(rows[0]), first, count, frame_major_rows=4097)
    got = cur.cpu().numpy()[: cen.size * nfr[0]].reshape(cen.size, nfr[0])
    refc = g["xb/modal/curves"]
    assert np.array_equal(np.isnan(got), np.isnan(refc))
    ok = ~np.isnan(refc)
    np.testing.assert_allclose(got[ok], refc[ok], rtol=0, atol=2e-5)
    assert np.mean(got[ok] == refc[ok]) > 0.99


def test_fused_stft_logbin_matches_the_two_kernel_path(golden):
    """ira_stft_logbin (dB matrix never written) gives the curves of ira_stft_mag_db + ira_logbin_aggregate."""
    from audio_analysis_amd.analyse import modalcloud as mc
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    g, c, _ = golden
    eng = get_engine()
    chans = [g["in/xb"], synth_ir(77, 0, 60001, rt60_seconds=0.5), synth_ir(78, 0, 30011, rt60_seconds=0.2)]
    b = eng.upload(chans)
    st = mc.ModalCloudAnalysisSettings()
    fused = mc.modal_cloud_device(eng, b, SR, st, fused=True)
    split = mc.modal_cloud_device(eng, b, SR, st, fused=False)
    cf, cs = fused["curves"].cpu().numpy(), split["curves"].cpu().numpy()
    assert cf.shape == cs.shape and np.array_equal(np.isnan(cf), np.isnan(cs))
    ok = ~np.isnan(cs)
    assert np.max(np.abs(cf[ok] - cs[ok])) < 2e-5 and np.mean(cf[ok] == cs[ok]) > 0.999
    np.testing.assert_array_equal(fused["fits"].cpu().numpy()[:, 0], split["fits"].cpu().numpy()[:, 0])   # same valid bins
    # and the fused kernel directly against the golden curves (segment and bins exactly as the golden script took them)
    seg = g["in/xb"][243:]
    bs = eng.upload([seg])
    nfr = np.array([1 + (seg.size - 8192) // 512], dtype=np.int32)
    freq = np.fft.rfftfreq(8192, 1 / 48000.0).astype(np.float32)
    rows = np.nonzero((freq >= 20.0) & (freq <= 20000.0))[0]
    cen, first, count = mc.log_bin_rows(freq[rows], g["modal/edges"])
    cur, _ = eng.stft_logbin(bs.x, bs.off, nfr, 8192, 512, True, -120.0, int(rows[0]), first, count)
    got = cur.cpu().numpy()[: cen.size * nfr[0]].reshape(cen.size, nfr[0])
    ref = g["xb/modal/curves"]
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    okr = ~np.isnan(ref)
    np.testing.assert_allclose(got[okr], ref[okr], rtol=0, atol=2e-5)
    assert np.mean(got[okr] == ref[okr]) > 0.99


@pytest.mark.parametrize("tag,inp", [("xb", "xb"), ("xb16", "xb16"), ("xb_t20", "xb"), ("xd_4096", "xd")])
def test_modal_cloud_vs_golden(golden, tag, inp):
    from audio_analysis_amd.analyse import modalcloud as mc
    g, c, _ = golden
    cs = c[f"{tag}/modal"]
    r = mc.analyse_modal_cloud_for_channel(g[f"in/{inp}"], SR, "mono", mc.ModalCloudAnalysisSettings(**cs["kw"]))
    assert (r.analysis_start_sample_index, r.analysis_length_samples, r.metric) == (cs["start"], cs["length"], cs["metric"])
    ref = g[f"{tag}/modal/points"]
    got = np.array([[p.centre_hz, p.rt60_seconds, p.r_squared] for p in r.points]).reshape(-1, 3)
    assert got.shape == ref.shape                      # same set of valid bins
    np.testing.assert_array_equal(got[:, 0], ref[:, 0])
    relerr = np.abs(got[:, 1] - ref[:, 1]) / np.abs(ref[:, 1])
    assert np.max(relerr) < 1e-4, np.max(relerr)       # north_star RT60 tolerance
    assert np.mean(relerr < 1e-6) > 0.95
    np.testing.assert_allclose(got[:, 2], ref[:, 2], rtol=0, atol=1e-6)
    assert mc.summarise_modal_cloud_results_text([r]) == cs["summary"]


def test_modal_and_waterfall_batch_vs_oracle():
    from audio_analysis_amd.analyse import modalcloud as mc, waterfall as wf
    from audio_analysis_amd.synth import synth_ir
    chans = [synth_ir(i, 0, 72000 + 501 * i, rt60_seconds=0.3 + 0.1 * i) for i in range(3)]
    names = ["a", "b", "c"]
    mres = mc.analyse_modal_cloud_batch(chans, SR, names, mc.ModalCloudAnalysisSettings())
    wres = wf.analyse_waterfall_batch(chans, SR, names, wf.WaterfallAnalysisSettings())
    for x, m, w in zip(chans, mres, wres):
        o = O.analyse_modal_cloud(x, SR)
        assert len(m.points) == len(o["points"])
        for p, q in zip(m.points, o["points"]):
            assert p.centre_hz == q[0] and _rel(p.rt60_seconds, q[1]) < 1e-4
        ow = O.analyse_waterfall(x, SR)
        np.testing.assert_array_equal(w.slice_times_seconds, ow["slice_times_seconds"])
        np.testing.assert_allclose(w.slice_magnitude_rel_db, ow["slice_rel_db"], rtol=0, atol=3e-5)
