"""
GPU, BASELINE.json's configurations at FULL size: the streaming path (pinned host batches ->
DeviceFeed -> FullReport, 256 channels per step) over
  * config 3: 4096 x 10 s IRs, third-octave rt60bands (26 bands) + waterfall,
  * config 4: 2048 x 10 s IRs (one GPU's shard of 16384), zplane AR(64) + modal cloud,
  * config 5: 2048 stereo 5 s PCM16 taps from files through bundle.run_bundle_metrics (native ingest ring),
and config 1's substitutes (the reference's own sample is not in its checkout): a 13 s mono sweep-like file
and a 24 s stereo IR of N = 1 151 844 samples (= 2^2 * 3 * 95987: every full-file transform goes through Bluestein at
M = 2^22).  The oracle needs seconds per channel at these sizes, so parity is
  - structural over the whole job: status ok, complete records, no NaN where a value must exist, records independent
    of how the job was chunked (a sample of channels re-analysed in other batch compositions, byte for byte), linked
    copies of a tap giving identical records wherever they land in the ingest ring,
  - and against the oracle on single channels.
"""
import os
from concurrent.futures import ThreadPoolExecutor
from dataclasses import replace

import numpy as np
import pytest

from oracle import ira_oracle as O

pytestmark = pytest.mark.gpu
SR = 48000
N = 480000


def _synth_many(first, count, n, channel=0):
    from audio_analysis_amd.synth import synth_ir
    workers = max(1, min(16, len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(max_workers=workers) as ex:
        return list(ex.map(lambda i: synth_ir(first + i, channel, n), range(count)))


@pytest.fixture(scope="module")
def big():
    """4096 distinct synthetic 10 s IRs in host memory (7.9 GB), shared by the config 3 and config 4 tests."""
    from audio_analysis_amd.engine import get_engine
    return get_engine(), _synth_many(7000, 4096, N)


def _stream(eng, settings, chans, per_step=256):
    from audio_analysis_amd.feed import DeviceFeed, HostBatch, run_pipelined
    from audio_analysis_amd.pipeline import FullReport
    rep = FullReport(eng, settings)
    feed = DeviceFeed(eng, per_step * max(c.size for c in chans), depth=4)
    rows = []

    def batches():
        for a in range(0, len(chans), per_step):
            yield HostBatch(eng, chans[a : a + per_step])

    steps = run_pipelined(rep, feed, batches(), rows.append)
    assert steps == -(-len(chans) // per_step)
    return rep, np.concatenate(rows)


def test_config3_4096_irs_third_octave_bands_and_waterfall(big):
    from audio_analysis_amd import pipeline as P
    eng, chans = big
    none = replace(P.FullReportSettings(), run_decay=False, run_frequency_response=False, run_filter=False,
                   run_spectrogram=False, run_modal_cloud=False, run_zplane=False)
    s = replace(none, rt60_bands=replace(none.rt60_bands, band_mode="third"))
    rep, m = _stream(eng, s, chans)
    assert m.shape == (4096, P.METRICS_WIDTH)
    assert np.all(m[:, P.M_STATUS] == 0.0) and np.all(m[:, P.M_NSAMPLES] == N) and np.all(m[:, P.M_NBANDS] == 26)
    t30 = m[:, P.M_BANDS : P.M_BANDS + 3 * 26 : 3]
    # mid and high bands of a 0.3-3 s broadband decay always have a T30 (low bands may not: their wrapped pre-ringing can keep
    # the EDC above -35 dB, and lifts some fitted values to hundreds of seconds -- reproduced, not "fixed": SURVEY a9);
    # the top octave is clean enough to sit inside the generator's RT60 range
    mid = t30[:, 10:]
    assert np.all(np.isfinite(mid)) and mid.min() > 0.2
    top = t30[:, -4:]
    assert top.max() < 4.0 and np.all(np.abs(np.median(top, axis=1) - np.median(t30[:, -8:-4], axis=1)) < 0.5)
    assert np.all(m[:, P.M_WF_SLICES] >= 2) and np.all(m[:, P.M_WF_BINS] == 1705)
    # the job's records do not depend on its chunking: channels from four different steps, re-analysed as one small batch
    pick = [3, 300, 1999, 4095]
    again = P.FullReport(eng, s).run(eng.upload([chans[i] for i in pick]))
    assert again.tobytes() == m[pick].tobytes()
    # one channel band by band against the oracle
    o = O.analyse_rt60_bands(chans[300], SR, band_mode="third")
    assert len(o["bands"]) == 26
    for k, band in enumerate(o["bands"]):
        ref = o["metrics"][band["name"]]["t30"]
        got = m[300, P.M_BANDS + 3 * k]
        assert (ref is None) == bool(np.isnan(got)), band["name"]
        if ref is not None:
            assert abs(got - ref) <= 1e-4 * abs(ref), (band["name"], got, ref)


def test_config4_2048_irs_zplane_and_modal_cloud(big):
    from audio_analysis_amd import pipeline as P
    eng, chans = big
    chans = chans[:2048]
    s = replace(P.FullReportSettings(), run_decay=False, run_rt60_bands=False, run_frequency_response=False,
                run_filter=False, run_spectrogram=False, run_waterfall=False)
    rep, m = _stream(eng, s, chans)
    assert m.shape == (2048, P.METRICS_WIDTH) and np.all(m[:, P.M_STATUS] == 0.0)
    assert np.all(m[:, P.M_AR_POLES] == 64)
    assert np.all(np.isfinite(m[:, P.M_AR_MAX_R])) and np.all(m[:, P.M_AR_MAX_R] < 1.01) and np.all(m[:, P.M_AR_MEDIAN_R] > 0.3)
    assert np.all(m[:, P.M_MODAL_POINTS] > 100)                       # white-noise decays: nearly every log bin fits
    assert np.all(np.isfinite(m[:, P.M_MODAL_MEDIAN])) and m[:, P.M_MODAL_MEDIAN].min() > 0.2 and m[:, P.M_MODAL_MEDIAN].max() < 3.5
    pick = [0, 257, 1024, 2047]
    again = P.FullReport(eng, s).run(eng.upload([chans[i] for i in pick]))
    assert again.tobytes() == m[pick].tobytes()
    z = O.analyse_zplane(chans[257], SR, ar_order=64)
    assert abs(m[257, P.M_AR_MAX_R] - z["max_radius"]) <= 1e-4 * z["max_radius"]
    assert abs(m[257, P.M_AR_MEDIAN_R] - z["median_radius"]) <= 1e-4 * z["median_radius"]
    assert int(m[257, P.M_AR_UNSTABLE]) == z["unstable"]
    mc = O.analyse_modal_cloud(chans[257], SR)
    rt = np.array([p[1] for p in mc["points"]])
    assert int(m[257, P.M_MODAL_POINTS]) == rt.size
    assert abs(m[257, P.M_MODAL_MEDIAN] - np.median(rt)) <= 1e-4 * np.median(rt)


def test_config5_one_gpus_share_of_the_bundle_through_the_ingest_ring(tmp_path):
    """BASELINE config 5 at its own size: 65 536 stereo 5 s taps over 8 GPUs = 8 192 taps (16 384 channels) per GPU, read from
    files in the recorder's format (256 distinct taps, the rest hard links to them: the same bytes analysed in 32 different
    steps must give the same records), 128 taps per step."""
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.analyse import bundle
    n = 240000
    distinct, copies = 256, 32
    root = tmp_path / "bundle"
    (root / "taps").mkdir(parents=True)
    left, right = _synth_many(9000, distinct, n, 0), _synth_many(9000, distinct, n, 1)
    names = []
    for c in range(copies):
        for i in range(distinct):
            name = f"tap_{c:02d}_{i:03d}"               # (the recorder's meta.json lists taps in std::map = lexicographic order)
            names.append(name)
            path = root / "taps" / f"{name}.wav"
            if c == 0:
                path.write_bytes(O.recorder_wav_bytes(np.stack([left[i], right[i]], axis=1)))
            else:
                os.link(root / "taps" / f"tap_00_{i:03d}.wav", path)
    (root / "meta.json").write_text(O.recorder_meta_json(SR, n, names))
    labels, rec = bundle.run_bundle_metrics(root, taps_per_step=128)
    assert len(labels) == 2 * distinct * copies and rec.shape == (16384, P.METRICS_WIDTH)
    assert labels[:2] == [(names[0], "left"), (names[0], "right")] and labels[-1] == (names[-1], "right")
    assert np.all(rec[:, P.M_STATUS] == 0.0) and np.all(rec[:, P.M_NSAMPLES] == n)
    first = rec[: 2 * distinct]
    for c in range(1, copies):                                        # the same file analysed in 32 different steps
        assert rec[c * 2 * distinct : (c + 1) * 2 * distinct].tobytes() == first.tobytes(), c
    # one channel against the oracle, read from the file the way the reference's loader would
    _, raw = O.wav_pcm16_payload((root / "taps" / "tap_00_017.wav").read_bytes())
    x = [v for _, v in O.analysis_channels(O.pcm_to_float32(raw), False)][1]
    row = rec[2 * 17 + 1]
    d = O.analyse_decay(x)
    assert row[P.M_START] == d["start"]
    assert abs(row[P.M_FIT_T30 + 6] - d["fits"]["T30"]["rt60"]) <= 1e-6 * d["fits"]["T30"]["rt60"]
    fr = O.analyse_frequency_response(x)
    assert row[P.M_FR_PEAK] == fr["peak_hz"] and abs(row[P.M_FR_CENTROID] - fr["centroid_hz"]) <= 1e-6 * fr["centroid_hz"]


def _check_against_oracle(row, x, what):
    from audio_analysis_amd import pipeline as P
    d = O.analyse_decay(x)
    assert int(row[P.M_START]) == d["start"], what
    for name, slot in (("T20", P.M_FIT_T20), ("T30", P.M_FIT_T30)):
        f = d["fits"].get(name)
        assert (f is None) == (row[slot] != 1.0), (what, name)
        if f is not None:
            assert abs(row[slot + 6] - f["rt60"]) <= 1e-6 * abs(f["rt60"]), (what, name)
    b = O.analyse_rt60_bands(x)
    for k, band in enumerate(b["bands"]):
        ref = b["metrics"][band["name"]]["t30"]
        got = row[P.M_BANDS + 3 * k]
        assert (ref is None) == bool(np.isnan(got)), (what, band["name"])
        if ref is not None:
            assert abs(got - ref) <= 1e-4 * abs(ref), (what, band["name"], got, ref)
    fr = O.analyse_frequency_response(x)
    assert row[P.M_FR_PEAK] == fr["peak_hz"], what
    assert abs(row[P.M_FR_CENTROID] - fr["centroid_hz"]) <= 1e-6 * fr["centroid_hz"], what
    fl = O.analyse_filter_response(x)
    assert abs(row[P.M_FILT_1K] - fl["mag_1k_db"]) <= 2e-5, what
    sp = O.analyse_spectrogram(x)
    assert int(row[P.M_SPEC_FRAMES]) == sp["magnitude_db"].shape[1], what
    mc = O.analyse_modal_cloud(x)
    assert int(row[P.M_MODAL_POINTS]) == len(mc["points"]), what
    if mc["points"]:
        rt = np.array([p[1] for p in mc["points"]])
        assert abs(row[P.M_MODAL_MEDIAN] - np.median(rt)) <= 1e-4 * np.median(rt), what
    z = O.analyse_zplane(x, ar_order=64)
    assert abs(row[P.M_AR_MAX_R] - z["max_radius"]) <= 1e-4 * z["max_radius"], what
    assert int(row[P.M_AR_UNSTABLE]) == z["unstable"], what


def test_config1_substitutes_long_files_through_bluestein():
    """13 s mono sweep-like file (N = 624000 = 2^7 3 5^3 13: not smooth) and the 24 s stereo IR with the sample count of the
    reference's example report (plots/example/verb_report.md:6, N = 1 151 844): full report, both against the oracle."""
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    n13 = 13 * SR
    t = np.arange(n13, dtype=np.float64) / SR
    k = np.log(20000.0 / 20.0) / 12.0
    sweep = 0.5 * np.sin(2.0 * np.pi * 20.0 * (np.exp(k * np.minimum(t, 12.0)) - 1.0) / k)
    sweep[t > 12.0] *= np.exp(-(t[t > 12.0] - 12.0) * 12.0)                    # one second of decaying tail
    sweep = sweep.astype(np.float32)
    n24 = 1_151_844
    left, right = synth_ir(0, 0, n24), synth_ir(0, 1, n24)
    assert eng.smooth_split(n13) is None and eng.smooth_split(n24) is None        # both go through Bluestein
    m = P.FullReport(eng).run(eng.upload([sweep, left, right]))
    assert m.shape == (3, P.METRICS_WIDTH) and np.all(m[:, P.M_STATUS] == 0.0)
    assert np.array_equal(m[:, P.M_NSAMPLES], [n13, n24, n24])
    _check_against_oracle(m[0], sweep, "13 s sweep")
    _check_against_oracle(m[1], left, "24 s stereo IR, left")
    _check_against_oracle(m[2], right, "24 s stereo IR, right")
