"""
GPU, BASELINE.json's full size (48 kHz x 10 s IRs, the bench batch): size-independent properties of the whole
metrics-only report.  The oracle needs ~1.2 s per IR at this size, so parity here is structural:
  * batch-composition invariance: a channel's 128-double record does not depend on its neighbours, its position,
    the batch size, or on how transforms were paired (real-signal / band pairing) -- byte for byte;
  * amplitude invariance: scaling the input by a power of two leaves every index, RT60, radius and frequency unchanged;
  * delay covariance: prepending silence moves the start index by exactly that many samples and nothing else;
  * one full-size spot check against the oracle (decay + zplane radii + fr peak) on a single IR.
"""
import numpy as np
import pytest

from oracle import ira_oracle as O

pytestmark = pytest.mark.gpu
SR = 48000
N = 480000


@pytest.fixture(scope="module")
def setup():
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.pipeline import FullReport
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    chans = [synth_ir(500 + i, 0, N) for i in range(12)]
    return eng, FullReport(eng), chans


def _run(eng, rep, chans):
    return rep.run(eng.upload(chans))


def test_records_do_not_depend_on_batch_composition(setup):
    eng, rep, chans = setup
    full = _run(eng, rep, chans)                                     # 12 channels: pairs (0,1) (2,3) ...
    assert full.shape == (12, 128) and np.all(full[:, 0] == 0.0)
    odd = _run(eng, rep, chans[:5])                                  # channel 4 is now the unpaired leftover
    assert odd.tobytes() == full[:5].tobytes()
    rev = _run(eng, rep, chans[::-1])                                # different partners, different order
    assert rev[::-1].tobytes() == full.tobytes()
    single = _run(eng, rep, [chans[7]])
    assert single.tobytes() == full[7:8].tobytes()
    dup = _run(eng, rep, [chans[3], chans[3], chans[9], chans[3]])   # identical signals paired with each other
    assert dup[0].tobytes() == dup[1].tobytes() == dup[3].tobytes() == full[3].tobytes()


def test_amplitude_invariance_and_delay_covariance(setup):
    from audio_analysis_amd import pipeline as P
    eng, rep, chans = setup
    base = _run(eng, rep, chans[:2])
    half = _run(eng, rep, [c * np.float32(0.5) for c in chans[:2]])  # exact in float32
    level_free = [P.M_START, P.M_EARLY10, P.M_FR_PEAK, P.M_FR_CENTROID, P.M_SPEC_FRAMES, P.M_WF_SLICES, P.M_WF_BINS,
                  P.M_AR_POLES, P.M_AR_UNSTABLE, P.M_NBANDS]
    for col in level_free:
        np.testing.assert_allclose(half[:, col], base[:, col], rtol=1e-9, atol=0, equal_nan=True)
    for col in (P.M_FIT_T30 + 6, P.M_FIT_T20 + 6, P.M_AR_MAX_R, P.M_AR_MEDIAN_R):
        np.testing.assert_allclose(half[:, col], base[:, col], rtol=1e-6, equal_nan=True)
    nb = int(base[0, P.M_NBANDS])
    np.testing.assert_allclose(half[:, P.M_BANDS : P.M_BANDS + 3 * nb], base[:, P.M_BANDS : P.M_BANDS + 3 * nb],
                               rtol=1e-6, equal_nan=True)
    # 6.0206 dB lower filter magnitude at 1 kHz, same everything else
    np.testing.assert_allclose(half[:, P.M_FILT_1K], base[:, P.M_FILT_1K] - 20 * np.log10(2.0), atol=2e-5)

    pad = 4800
    delayed = _run(eng, rep, [np.concatenate([np.zeros(pad, np.float32), c[:-pad]]) for c in chans[:2]])
    assert np.all(delayed[:, P.M_START] == base[:, P.M_START] + pad)


def test_full_size_spot_check_against_oracle(setup):
    from audio_analysis_amd import pipeline as P
    eng, rep, chans = setup
    x = chans[0]
    m = _run(eng, rep, [x])[0]
    d = O.analyse_decay(x, SR)
    assert int(m[P.M_START]) == d["start"]
    f30 = d["fits"]["T30"]
    assert abs(m[P.M_FIT_T30 + 6] - f30["rt60"]) / f30["rt60"] < 1e-6
    fr = O.analyse_frequency_response(x, SR)
    assert m[P.M_FR_PEAK] == fr["peak_hz"]
    z = O.analyse_zplane(x, SR, ar_order=64)
    assert abs(m[P.M_AR_MAX_R] - z["max_radius"]) / z["max_radius"] < 1e-4
    assert abs(m[P.M_AR_MEDIAN_R] - z["median_radius"]) / z["median_radius"] < 1e-4
    assert int(m[P.M_AR_UNSTABLE]) == z["unstable"]


def test_baseline_config_batch256_two_second_irs_spectrogram_and_decay():
    """BASELINE.json configs[1]: batch 256 synthetic 2 s @ 48 kHz IRs, STFT spectrogram + Schroeder decay on one GPU.
    Every channel's record is checked for completeness; a sample of channels against the oracle."""
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    chans = [synth_ir(1000 + i, 0, 96000, rt60_seconds=0.25 + 0.002 * i) for i in range(256)]
    rep = P.FullReport(eng, P.FullReportSettings(run_rt60_bands=False, run_frequency_response=False, run_filter=False,
                                                 run_waterfall=False, run_modal_cloud=False, run_zplane=False))
    m = rep.run(eng.upload(chans))
    assert m.shape == (256, 128) and np.all(m[:, P.M_FIT_T30] == 1.0)          # every T30 fit valid
    assert np.all(np.diff(m[:, P.M_FIT_T30 + 6]) > -0.05)                       # RT60 follows the generator's ramp
    for i in (0, 101, 255):
        d = O.analyse_decay(chans[i], SR)
        assert int(m[i, P.M_START]) == d["start"]
        assert abs(m[i, P.M_FIT_T30 + 6] - d["fits"]["T30"]["rt60"]) <= 1e-6 * d["fits"]["T30"]["rt60"]
        sp = O.analyse_spectrogram(chans[i], SR)
        assert int(m[i, P.M_SPEC_FRAMES]) == sp["magnitude_db"].shape[1]
    # the spectrogram itself for one channel of the big batch, against the oracle
    out = rep.device_results["spectrogram"]
    sp = O.analyse_spectrogram(chans[101], SR)
    ref = sp["magnitude_db"]
    off, cols = int(out["mag_off"][101]), int(out["cols"][101])
    flat = out["mag"][off : off + ref.shape[0] * cols].cpu().numpy()
    got = flat.reshape(cols, ref.shape[0]).T if out.get("frame_major") else flat.reshape(ref.shape[0], cols)
    peak = ref.max(axis=0, keepdims=True)
    strong = (ref > peak - 50.0) & (ref > -100.0)
    assert np.max(np.abs(got - ref)[strong]) < 1e-3


def test_baseline_config_third_octave_bands_and_waterfall_ten_seconds():
    """BASELINE.json configs[2] at reduced batch: third-octave rt60bands + waterfall on 10 s IRs (the oracle needs a few
    seconds per channel for 30 bands), one channel checked band by band."""
    from audio_analysis_amd.analyse import rt60bands as rb, waterfall as wf
    from audio_analysis_amd.synth import synth_ir
    chans = [synth_ir(2000 + i, 0, N, rt60_seconds=0.8 + 0.1 * i) for i in range(4)]
    st = rb.Rt60BandsAnalysisSettings(band_mode="third")
    res = rb.analyse_rt60_bands_batch(chans, SR, list("abcd"), st)
    o = O.analyse_rt60_bands(chans[2], SR, band_mode="third")
    assert list(res[2].band_metrics_by_name) == [b["name"] for b in o["bands"]]
    for name, mm in res[2].band_metrics_by_name.items():
        w = o["metrics"][name]["t30"]
        assert (mm.rt60_t30_seconds is None) == (w is None), name
        if w is not None:
            assert abs(mm.rt60_t30_seconds - w) <= 1e-4 * abs(w), (name, mm.rt60_t30_seconds, w)
    wres = wf.analyse_waterfall_batch(chans, SR, list("abcd"), wf.WaterfallAnalysisSettings())
    ow = O.analyse_waterfall(chans[2], SR)
    np.testing.assert_array_equal(wres[2].slice_times_seconds, ow["slice_times_seconds"])
    np.testing.assert_allclose(wres[2].slice_magnitude_rel_db, ow["slice_rel_db"], rtol=0, atol=2e-5)


def test_baseline_config_zplane_and_modalcloud_sharded_over_ranks():
    """BASELINE.json config 4 (zplane AR(64) + modalcloud, per-IR shards): only those two blocks enabled; the records of
    the IRs analysed as one batch equal the concatenation of 2, 4 and 8 contiguous shards (dist.shard_files) byte for
    byte -- what the single gather delivers on N GPUs -- and agree with the oracle on one 10 s IR."""
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.dist import shard_files
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    s = P.FullReportSettings(run_decay=False, run_rt60_bands=False, run_frequency_response=False, run_filter=False,
                             run_spectrogram=False, run_waterfall=False)
    assert s.blocks() == ["modalcloud[8192/512]", "zplane[ar64]"]
    chans = [synth_ir(900 + i, 0, N) for i in range(9)]               # 9 files: ragged last shards
    full = P.FullReport(eng, s).run(eng.upload(chans))
    for world in (2, 4, 8):
        parts = []
        for r in range(world):
            lo, hi = shard_files(len(chans), r, world)
            if hi > lo:
                parts.append(P.FullReport(eng, s).run(eng.upload(chans[lo:hi])))
        assert np.concatenate(parts, axis=0).tobytes() == full.tobytes(), world
    z = O.analyse_zplane(chans[4], SR, ar_order=64)
    assert abs(full[4, P.M_AR_MAX_R] - z["max_radius"]) < 1e-8 and abs(full[4, P.M_AR_MEDIAN_R] - z["median_radius"]) < 1e-8
    assert int(full[4, P.M_AR_UNSTABLE]) == z["unstable"] and int(full[4, P.M_AR_POLES]) == 64
    mc = O.analyse_modal_cloud(chans[4], SR)
    rt = np.array([p[1] for p in mc["points"]])
    assert int(full[4, P.M_MODAL_POINTS]) == rt.size
    assert abs(full[4, P.M_MODAL_MEDIAN] - np.median(rt)) <= 1e-4 * np.median(rt)
    assert abs(full[4, P.M_MODAL_MAX] - rt.max()) <= 1e-4 * rt.max()
    assert np.all(np.isnan(full[:, P.M_FIT_T30 : P.M_FIT_T30 + 8]))     # disabled blocks leave their slots empty


def test_baseline_config_bundle_of_stereo_five_second_taps(tmp_path):
    """BASELINE.json config 5 (bundle report, stereo taps x 5 s, full pipeline) at a reduced tap count: taps in the
    recorder's on-disk format -> native ingest -> full report, several taps per step; records equal the float-upload
    path byte for byte and the oracle on one channel (which reads the file the way the reference would)."""
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.analyse import bundle
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    n = 240000
    root = tmp_path / "b"
    (root / "taps").mkdir(parents=True)
    names = [f"t{i}" for i in range(6)]
    for i, name in enumerate(names):
        st = np.stack([synth_ir(300 + i, c, n) for c in (0, 1)], axis=1)
        (root / "taps" / f"{name}.wav").write_bytes(O.recorder_wav_bytes(st))
    (root / "meta.json").write_text(O.recorder_meta_json(SR, n, names))
    labels, rec = bundle.run_bundle_metrics(root, taps_per_step=4)
    assert labels == [(t, ch) for t in names for ch in ("left", "right")] and rec.shape == (12, P.METRICS_WIDTH)
    _, raw = O.wav_pcm16_payload((root / "taps" / "t2.wav").read_bytes())
    f = O.pcm_to_float32(raw)
    chans = [x for _, x in O.analysis_channels(f, False)]
    eng = get_engine()
    direct = P.FullReport(eng).run(eng.upload(chans))
    assert direct.tobytes() == rec[4:6].tobytes()
    x = chans[1]
    d = O.analyse_decay(x)
    assert rec[5, P.M_START] == d["start"] and rec[5, P.M_NSAMPLES] == n
    assert abs(rec[5, P.M_FIT_T30 + 6] - d["fits"]["T30"]["rt60"]) <= 1e-6 * d["fits"]["T30"]["rt60"]
    b = O.analyse_rt60_bands(x)
    for k, band in enumerate(b["bands"]):
        ref = b["metrics"][band["name"]]["t30"]
        got = rec[5, P.M_BANDS + 3 * k]
        assert (ref is None) == bool(np.isnan(got))
        if ref is not None:
            assert abs(got - ref) <= 1e-4 * abs(ref)
    fr = O.analyse_frequency_response(x)
    assert rec[5, P.M_FR_PEAK] == fr["peak_hz"]
