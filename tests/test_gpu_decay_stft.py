"""GPU parity: decay (EDC + fits) and STFT/spectrogram vs the oracle and the golden vectors."""
import numpy as np
import pytest

from oracle import ira_oracle as O

pytestmark = pytest.mark.gpu
SR = 48000


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


@pytest.fixture(scope="module")
def amd():
    from audio_analysis_amd.analyse import decay, spectrogram
    return decay, spectrogram


def test_peak_index_exact(golden):
    g, c, _ = golden
    from audio_analysis_amd.engine import get_engine
    eng = get_engine()
    chans = [g[f"in/{k}"] for k in ("xa", "xb", "xb16", "xc", "xd", "xe", "xs_l", "xs_r")]
    rng = np.random.default_rng(3)
    ties = np.zeros(70001, dtype=np.float32); ties[[17, 40000, 69999]] = [-0.5, 0.5, 0.5]   # first max wins
    chans += [ties, np.zeros(100, np.float32), rng.standard_normal(1_000_003).astype(np.float32)]
    b = eng.upload(chans)
    pk = eng.peaks(b)
    for i, x in enumerate(chans):
        assert pk[i] == int(np.argmax(np.abs(x))), i
        assert b.peak_abs[i] == np.abs(x).max()


@pytest.mark.parametrize("tag,inp", [("xa", "xa"), ("xa_edt", "xa"), ("xa_ign", "xa"), ("xa_notrim", "xa"),
                                     ("xa_smooth", "xa"), ("xb", "xb"), ("xb16", "xb16"), ("xc", "xc")])
def test_decay_vs_golden(golden, amd, tag, inp):
    decay, _ = amd
    g, c, _ = golden
    case = c[f"{tag}/decay"]
    r = decay.analyse_decay_for_channel(g[f"in/{inp}"], SR, "mono", decay.DecayAnalysisSettings(**case["kw"]))
    assert r.analysis_start_sample_index == case["start"]            # bit-exact index
    ref = g[f"{tag}/decay/edc_db"]
    assert r.edc_db.shape == ref.shape and r.edc_db.dtype == np.float32
    if not case["kw"].get("edc_smoothing_window_samples"):
        assert r.edc_db[0] == 0.0          # exact: the kernel normalises by its own value at index 0
    # float64 scan order differs from numpy's sequential cumsum only in the last bits: <= 1 float32 ulp
    np.testing.assert_allclose(r.edc_db, ref, rtol=3e-7, atol=1e-6)
    assert (r.early_decay_10db_time_seconds is None) == (case["early"] is None)
    if case["early"] is not None:
        assert _rel(r.early_decay_10db_time_seconds, case["early"]) < 1e-6
    assert set(r.fits) == set(case["fits"])
    for k, f in r.fits.items():
        gold = case["fits"][k]
        assert _rel(f.rt60_seconds, gold[7]) < 1e-6, (k, f.rt60_seconds, gold[7])   # north_star bar is 1e-4
        assert _rel(f.slope_db_per_second, gold[4]) < 1e-6
        assert abs(f.r_squared - gold[6]) < 1e-9
        assert abs(f.start_time_seconds - gold[2]) < 1e-7 and abs(f.end_time_seconds - gold[3]) < 1e-7
    assert decay.summarise_decay_results_text([r]) == case["summary"]


def test_public_fit_function(golden, amd):
    decay, _ = amd
    g, c, _ = golden
    t, db, start = decay.compute_schroeder_edc_db(g["in/xb"], SR, decay.DecayAnalysisSettings())
    assert start == c["xb/decay"]["start"]
    f = decay.fit_decay_slope_over_db_range(t, db, (-5.0, -35.0), -80.0, "T30")
    assert _rel(f.rt60_seconds, c["xb/decay"]["fits"]["T30"][7]) < 1e-6
    assert decay.fit_decay_slope_over_db_range(t, db, (-5.0, -200.0), -300.0, "X") is None
    with pytest.raises(ValueError):
        decay.fit_decay_slope_over_db_range(t, db, (-25.0, -5.0), -80.0, "bad")
    with pytest.raises(ValueError):
        decay.compute_schroeder_edc_db(np.zeros((4, 2), np.float32), SR, decay.DecayAnalysisSettings())
    with pytest.raises(ValueError):
        decay.analyse_decay_for_channel(np.array([0, 0, 1.0], np.float32), SR, "m", decay.DecayAnalysisSettings())


STFT_F32_STATS = []


def _stft_check(got, ref, floor_db=-120.0):
    """float32-butterfly tolerance against the REFERENCE's values (golden), stated on the bins SURVEY.md section 8(d) names
    -- every bin whose reference value is > floor + 20 dB:
      * max |delta| <= 4e-3 dB, and >= 99.9 % of those bins within 1e-3 dB (section 8d's figure holds for all but a few
        bins in ten thousand: a float32 transform's error is ~2e-7 of the FRAME's rms, so the weakest bins of a frame carry
        the largest dB error);
      * 1e-3 dB on every bin within 50 dB of its frame's peak;
      * linear error below 3e-6 of the frame's peak everywhere (floor-clamped bins included)."""
    assert got.shape == ref.shape and got.dtype == np.float32
    peak = ref.max(axis=0, keepdims=True)
    named = ref > floor_db + 20.0
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    STFT_F32_STATS.append((int(named.sum()), float(err[named].max()), float(np.mean(err[named] <= 1e-3))))
    assert err[named].max() <= 4e-3, err[named].max()
    assert np.mean(err[named] <= 1e-3) >= 0.999
    strong = named & (ref > peak - 50.0)
    assert np.max(err[strong]) < 1e-3
    lin_err = np.abs(10.0 ** (got.astype(np.float64) / 20) - 10.0 ** (ref.astype(np.float64) / 20))
    assert np.max(lin_err / 10.0 ** (peak.astype(np.float64) / 20)) < 3e-6


@pytest.mark.parametrize("precision", [32, 64])
def test_stft_vs_golden(golden, precision):
    g, c, _ = golden
    from audio_analysis_amd.engine import get_engine
    eng = get_engine()
    for nfft in (4096, 8192):
        cs = c[f"stft{nfft}"]
        seg = g["in/xb"][cs["seg_start"] : cs["seg_start"] + cs["seg_len"]]
        b = eng.upload([seg])
        nfr = np.array([1 + (seg.size - nfft) // 512], dtype=np.int32)
        out, off, cols = eng.stft_mag_db(b.x, b.off, nfr, nfft, 512, True, -120.0, precision)
        got = out.cpu().numpy()[: (nfft // 2 + 1) * nfr[0]].reshape(nfft // 2 + 1, nfr[0])
        ref = g[f"stft{nfft}/mag_db"]
        if precision == 64:
            np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)     # f64 butterflies: float32 rounding only
            assert np.mean(got == ref) > 0.99
        else:
            _stft_check(got, ref)
    seg = g["in/xa"][240:6240]
    b = eng.upload([seg])
    nfr = np.array([1 + (seg.size - 1024) // 256], dtype=np.int32)
    out, _, _ = eng.stft_mag_db(b.x, b.off, nfr, 1024, 256, False, -100.0, precision)
    got = out.cpu().numpy()[: 513 * nfr[0]].reshape(513, nfr[0])
    if precision == 64:
        np.testing.assert_allclose(got, g["stft1024rect/mag_db"], rtol=0, atol=2e-5)
    else:
        _stft_check(got, g["stft1024rect/mag_db"], -100.0)


def test_zz_report_f32_stft_error_statistics():
    """Prints what the float32 STFT tests above measured (bins > floor + 20 dB: count, max |delta dB|, fraction within 1e-3)."""
    for n, mx, frac in STFT_F32_STATS:
        print(f"f32 STFT vs reference: {n} bins > floor+20 dB, max {mx:.2e} dB, {100 * frac:.4f} % within 1e-3 dB")
    assert all(mx <= 4e-3 for _, mx, _ in STFT_F32_STATS)


def test_spectrogram_vs_golden(golden, amd):
    _, spectrogram = amd
    g, c, _ = golden
    r = spectrogram.analyse_spectrogram_for_channel(g["in/xa"], SR, "mono", spectrogram.SpectrogramAnalysisSettings())
    cs = c["xa/spectrogram"]
    assert (r.analysis_start_sample_index, r.analysis_length_samples) == (cs["start"], cs["length"])
    _stft_check(r.magnitude_db, g["xa/spectrogram/mag_db"])
    assert spectrogram.summarise_spectrogram_results_text([r]) == cs["summary"]
    np.testing.assert_array_equal(r.frequency_hz, np.fft.rfftfreq(4096, 1 / 48000.0).astype(np.float32))
    r = spectrogram.analyse_spectrogram_for_channel(
        g["in/xb"], SR, "mono",
        spectrogram.SpectrogramAnalysisSettings(ignore_leading_seconds=0.01, analysis_duration_seconds=0.5))
    cs = c["xb_sel/spectrogram"]
    assert (r.analysis_start_sample_index, r.analysis_length_samples, list(r.magnitude_db.shape)) == (
        cs["start"], cs["length"], cs["shape"])
    assert spectrogram.summarise_spectrogram_results_text([r]) == cs["summary"]
    with pytest.raises(ValueError):
        spectrogram.analyse_spectrogram_for_channel(g["in/xa"][:3000], SR, "m", spectrogram.SpectrogramAnalysisSettings())


def test_batch_matches_oracle_ragged():
    """Ragged batch at a mid size, compared with the oracle run on the same seeded inputs."""
    from audio_analysis_amd.analyse import decay, spectrogram
    from audio_analysis_amd.synth import synth_ir
    chans = [synth_ir(i, 0, 96000 - 1000 * i) for i in range(5)]
    res = decay.analyse_decay_batch(chans, SR, [f"c{i}" for i in range(5)], decay.DecayAnalysisSettings(compute_edt=True))
    spec = spectrogram.analyse_spectrogram_batch(chans, SR, [f"c{i}" for i in range(5)],
                                                 spectrogram.SpectrogramAnalysisSettings())
    for x, r, s in zip(chans, res, spec):
        o = O.analyse_decay(x, SR, compute_edt=True)
        assert r.analysis_start_sample_index == o["start"]
        np.testing.assert_allclose(r.edc_db, o["edc_db"], rtol=3e-7, atol=1e-6)
        assert set(r.fits) == set(o["fits"])
        for k in r.fits:
            assert _rel(r.fits[k].rt60_seconds, o["fits"][k]["rt60"]) < 1e-6
        os_ = O.analyse_spectrogram(x, SR)
        _stft_check(s.magnitude_db, os_["magnitude_db"])


def test_frame_major_stft_is_the_exact_transpose(golden):
    """ira_stft_mag_db_tf (f32 / 4096): same arithmetic as the (F, T) kernel, stored frame by frame."""
    from audio_analysis_amd.analyse import spectrogram as sp
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    chans = [synth_ir(400 + i, 0, 30000 + 777 * i, rt60_seconds=0.3) for i in range(3)]
    b = eng.upload(chans)
    st = sp.SpectrogramAnalysisSettings()
    a = sp.spectrogram_results(sp.spectrogram_device(eng, b, SR, st), SR, list("abc"), st)
    dev_tf = sp.spectrogram_device(eng, b, SR, st, frame_major=True)
    assert dev_tf["frame_major"]
    t = sp.spectrogram_results(dev_tf, SR, list("abc"), st)
    for ra, rt in zip(a, t):
        np.testing.assert_array_equal(ra.magnitude_db, rt.magnitude_db)
    # unsupported configuration falls back to the reference layout
    st2 = sp.SpectrogramAnalysisSettings(n_fft=2048, hop_length=256)
    assert not sp.spectrogram_device(eng, b, SR, st2, frame_major=True)["frame_major"]


def test_fused_edc_fits_match_the_curve_path():
    """ira_edc_fits (crossings + fits straight from the samples, no EDC read back) against ira_edc_db + ira_curve_fits on the
    SAME device: crossing times bit-identical (same float32 curve, same first index), regression fields to 1e-9 (one
    sweep of shifted moments vs three centred passes), emitted curve bit-identical.  Ragged lengths around the
    4096-sample tile / 16384-sample chunk edges, degenerate segments included."""
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    rng = np.random.default_rng(11)
    chans = []
    for i, n in enumerate([4, 9, 100, 4095, 4096, 4097, 16383, 16384, 16385, 16384 * 3 + 5, 96000, 250001]):
        x = synth_ir(40 + i, 0, max(n, 300), rt60_seconds=0.02 + 0.25 * rng.random(), pre_delay=0)[:n]
        chans.append(x.astype(np.float32))
    chans.append(np.zeros(5000, np.float32))                                   # digital silence
    chans.append(np.full(20000, 0.25, np.float32))                             # DC
    last = np.zeros(30000, np.float32); last[-1] = 1.0                         # all the energy in the last sample
    chans.append(last)
    slow = synth_ir(77, 0, 480000, rt60_seconds=8.0, pre_delay=0)             # -35 dB is never reached in 10 s
    chans.append(slow)
    nan = synth_ir(78, 0, 50000, rt60_seconds=0.2, pre_delay=0).copy(); nan[1234] = np.nan
    chans.append(nan)
    # a 60 s file whose 0 .. -10 dB range spans ~370 tiles: more than 80 chunks of four, so the moments kernel takes
    # five tiles per chunk (edc_moments_kernel's K > FIT_MIN_CHUNK_TILES branch)
    chans.append(synth_ir(79, 0, 2_880_000, rt60_seconds=200.0, pre_delay=0))
    b = eng.upload(chans)
    ranges = [(0.0, -10.0), (-5.0, -25.0), (-5.0, -35.0)]
    cross = (0.0, -10.0)
    edc, edc_off = eng.edc_db(b.x, b.off, b.length, 1e-20, -120.0)
    f_ref, c_ref = eng.curve_fits(edc, edc_off, b.length, 1.0, 48000.0, ranges, 8, cross=cross)
    f_new, c_new, edc2, edc_off2 = eng.edc_fits(b.x, b.off, b.length, 1e-20, -120.0, 1.0, 48000.0, ranges, 8,
                                                cross=cross, want_edc=True)
    f_nc, c_nc, none, _ = eng.edc_fits(b.x, b.off, b.length, 1e-20, -120.0, 1.0, 48000.0, ranges, 8, cross=cross)
    assert none is None
    e1, e2 = edc.cpu().numpy(), edc2.cpu().numpy()
    tot = int(b.length.sum())
    np.testing.assert_array_equal(e1[:tot].view(np.uint32), e2[:tot].view(np.uint32))
    assert np.array_equal(edc_off, edc_off2)
    fr, fn, fq = f_ref.cpu().numpy(), f_new.cpu().numpy(), f_nc.cpu().numpy()
    cr, cn, cq = c_ref.cpu().numpy(), c_new.cpu().numpy(), c_nc.cpu().numpy()
    np.testing.assert_array_equal(fn.view(np.uint64), fq.view(np.uint64))      # writing the curve changes nothing
    np.testing.assert_array_equal(cn.view(np.uint64), cq.view(np.uint64))
    np.testing.assert_array_equal(np.isnan(cr), np.isnan(cn))
    np.testing.assert_array_equal(cr[~np.isnan(cr)], cn[~np.isnan(cn)])        # crossing times: bit-identical
    assert fr.shape == fn.shape == (len(chans), 3, 8)
    for i in range(len(chans)):
        for j in range(3):
            a, g = fn[i, j], fr[i, j]
            assert a[0] == g[0], (i, j, a, g)
            for k in (1, 2):                                                   # start / end times
                assert (np.isnan(a[k]) and np.isnan(g[k])) or a[k] == g[k], (i, j, k, a, g)
            assert (np.isnan(a[7]) and np.isnan(g[7])) or a[7] == g[7], (i, j, a, g)   # points in the mask
            if g[0] == 1.0:
                assert _rel(a[3], g[3]) < 1e-9 and _rel(a[6], g[6]) < 1e-9, (i, j, a, g)
                assert abs(a[4] - g[4]) < 1e-9 * max(1.0, abs(g[4])) and abs(a[5] - g[5]) < 1e-10, (i, j, a, g)
            elif not np.isnan(g[3]):
                assert _rel(a[3], g[3]) < 1e-9 or (a[3] == 0.0 and g[3] == 0.0), (i, j, a, g)


@pytest.mark.parametrize("n_fft,hop", [(6000, 512), (1000, 250), (32768, 4096), (4097, 1000)])
def test_stft_of_any_frame_size(n_fft, hop):
    """the reference takes any positive n_fft (numpy.fft.rfft of arbitrary length; `--nfft 6000` is a valid CLI
    call).  Frame sizes outside the STFT kernels' powers of two run on the arbitrary-length transforms: spectrogram,
    waterfall and modal cloud against the oracle."""
    from audio_analysis_amd.analyse import spectrogram, waterfall, modalcloud
    from audio_analysis_amd.synth import synth_ir
    x = synth_ir(55, 0, 96000, rt60_seconds=0.4)
    r = spectrogram.analyse_spectrogram_for_channel(x, SR, "m", spectrogram.SpectrogramAnalysisSettings(n_fft=n_fft, hop_length=hop))
    o = O.analyse_spectrogram(x, SR, n_fft=n_fft, hop_length=hop)
    assert r.magnitude_db.shape == o["magnitude_db"].shape and r.analysis_start_sample_index == o["start"]
    np.testing.assert_array_equal(r.frequency_hz, o["frequency_hz"])
    np.testing.assert_array_equal(r.time_seconds, o["time_seconds"])
    assert np.max(np.abs(r.magnitude_db - o["magnitude_db"])) <= 2e-5           # float64 transforms both sides
    if n_fft <= 8192:
        w = waterfall.analyse_waterfall_for_channel(x, SR, "m", waterfall.WaterfallAnalysisSettings(n_fft=n_fft, hop_length=hop))
        ow = O.analyse_waterfall(x, SR, n_fft=n_fft, hop_length=hop)
        np.testing.assert_array_equal(w.slice_times_seconds, ow["slice_times_seconds"])
        np.testing.assert_allclose(w.slice_magnitude_rel_db, ow["slice_rel_db"], rtol=0, atol=2e-5)
        mc = modalcloud.analyse_modal_cloud_for_channel(x, SR, "m", modalcloud.ModalCloudAnalysisSettings(n_fft=n_fft, hop_length=hop))
        om = O.analyse_modal_cloud(x, SR, n_fft=n_fft, hop_length=hop)
        assert len(mc.points) == len(om["points"])
        for p, q in zip(mc.points, om["points"]):
            assert p.centre_hz == q[0] and abs(p.rt60_seconds - q[1]) <= 1e-4 * q[1]
