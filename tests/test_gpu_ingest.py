"""
GPU: native tap ingest.  int16 taps are uploaded as they lie on disk and converted on
the device; the float32 channels must be BIT-IDENTICAL to what the reference's loader + channel policy produce
(goldens from the reference reading a bundle the reference's own C++ recorder wrote; oracle on seeded extremes).
"""
import json
import shutil
import struct
from pathlib import Path

import numpy as np
import pytest
from scipy.io import wavfile

from oracle import ira_oracle as O

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"
TAPS = ["early", "late_hot"]
SR = 48000


def _host(batch):
    x = batch.x.cpu().numpy()
    return [x[o : o + n] for o, n in zip(batch.off, batch.length)]


@pytest.mark.parametrize("mono", [False, True])
def test_golden_bundle_channels_are_bit_exact(mono):
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.ingest import ingest_taps
    z = np.load(GOLD / "bundle_expected.npz")
    batch, labels = ingest_taps(get_engine(), [GOLD / "bundle" / "taps" / f"{t}.wav" for t in TAPS], mono)
    want = [(i, ch) for i in range(2) for ch in (["mono"] if mono else ["left", "right"])]
    assert labels == want
    for (i, ch), got in zip(labels, _host(batch)):
        ref = z[f"{TAPS[i]}/{'mix' if mono else 'split'}/{ch}"]
        assert got.dtype == np.float32 and got.tobytes() == ref.tobytes()


def test_extreme_values_ragged_lengths_and_other_encodings(tmp_path):
    """All 65536 int16 codes, odd frame counts (alignment of the next file's payload), a mono file, an empty file and a
    float32 file (decoded by the Python reader) in ONE batch; both channel policies."""
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.ingest import ingest_taps
    rng = np.random.default_rng(11)
    codes = np.arange(-32768, 32768, dtype=np.int16)
    files = {
        "all_codes": np.stack([codes, codes[::-1]], axis=1),                       # 65536 frames stereo
        "odd": rng.integers(-32768, 32768, size=(1001, 2)).astype(np.int16),
        "mono_odd": rng.integers(-32768, 32768, size=777).astype(np.int16),
        "one": np.array([[-32768, 32767]], dtype=np.int16),
        "f32": (rng.standard_normal((500, 2)) * 0.7).astype(np.float32),           # values beyond +-1 get clipped
        "tail": rng.integers(-32768, 32768, size=(4096, 2)).astype(np.int16),
    }
    paths = []
    for name, a in files.items():
        p = tmp_path / f"{name}.wav"
        wavfile.write(str(p), SR, a)
        paths.append(p)
    empty = tmp_path / "empty.wav"
    empty.write_bytes(b"RIFF" + struct.pack("<I", 36) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, SR, SR * 4, 4, 16)
                      + b"data" + struct.pack("<I", 0))
    paths.insert(3, empty)
    order = ["all_codes", "odd", "mono_odd", None, "one", "f32", "tail"]
    for mono in (False, True):
        batch, labels = ingest_taps(get_engine(), paths, mono)
        got = _host(batch)
        k = 0
        for i, name in enumerate(order):
            if name is None:
                chans = [("mono", np.zeros(0, np.float32))] if mono else [("left", np.zeros(0, np.float32)),
                                                                         ("right", np.zeros(0, np.float32))]
            else:
                a = files[name]
                chans = O.analysis_channels(O.pcm_to_float32(a if a.ndim == 2 else a.reshape(-1, 1)), mono)
            for ch, ref in chans:
                assert labels[k] == (i, ch)
                assert got[k].tobytes() == np.ascontiguousarray(ref, dtype=np.float32).tobytes(), (name, ch, mono)
                k += 1
        assert k == len(labels) == batch.count


def test_ingest_rejects_wrong_rate(tmp_path):
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.ingest import ingest_taps
    p = tmp_path / "r.wav"
    wavfile.write(str(p), 44100, np.zeros((100, 2), np.int16))
    with pytest.raises(ValueError, match="Expected sample rate 48000 Hz, but got 44100 Hz"):
        ingest_taps(get_engine(), [p])


def test_bundle_metrics_equal_the_float_upload_path(tmp_path):
    """run_bundle_metrics (native ingest, several taps per step, pipelined) == FullReport over the same channels
    uploaded as float32 from the oracle's reading of the files: byte-identical records."""
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.analyse import bundle
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    root = tmp_path / "b"
    (root / "taps").mkdir(parents=True)
    names = [f"tap{i:02d}" for i in range(5)]
    chans = []
    for i, name in enumerate(names):
        st = np.stack([synth_ir(70 + i, c, 30000, rt60_seconds=0.12 + 0.02 * i) for c in (0, 1)], axis=1)
        (root / "taps" / f"{name}.wav").write_bytes(O.recorder_wav_bytes(st))           # the recorder's format
        f = O.pcm_to_float32(O.recorder_float_to_pcm16(st))
        chans += [x for _, x in O.analysis_channels(f, False)]
    (root / "meta.json").write_text(O.recorder_meta_json(SR, 30000, names))
    labels, rec = bundle.run_bundle_metrics(root, taps_per_step=2)
    assert labels == [(n, ch) for n in names for ch in ("left", "right")]
    eng = get_engine()
    want = P.FullReport(eng).run(eng.upload(chans))
    assert rec.shape == want.shape == (10, P.METRICS_WIDTH)
    assert rec.tobytes() == want.tobytes()
    # the reference-written golden bundle runs end to end too
    dst = tmp_path / "g"
    shutil.copytree(GOLD / "bundle", dst)
    labels, rec = bundle.run_bundle_metrics(dst, use_mono_downmix_for_stereo=True)
    assert labels == [("early", "mono"), ("late_hot", "mono")] and rec.shape[0] == 2
    assert json.loads((dst / "meta.json").read_text())["taps"] == TAPS
    assert np.all(rec[:, P.M_NSAMPLES] == 12000) and np.all(rec[:, P.M_START] >= 240)


def test_bundle_edge_cases_empty_and_mono(tmp_path, capsys):
    """A bundle without taps gives an index and empty records; mono PCM16 taps (not what the recorder writes, but what
    the reference's loader accepts) run through both bundle paths; the CLI's bundle command works with the batched-path
    knobs taken from the environment."""
    import os
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.analyse import bundle, cli, report as rp
    from audio_analysis_amd.synth import synth_ir
    empty = tmp_path / "empty"
    (empty / "taps").mkdir(parents=True)
    (empty / "meta.json").write_text(json.dumps({"sample_rate_hz": SR, "length_samples": 0, "taps": []}))
    labels, rec = bundle.run_bundle_metrics(empty)
    assert labels == [] and rec.shape == (0, P.METRICS_WIDTH)
    idx = bundle.run_bundle_report(empty)
    assert idx.read_text().startswith("# IR Bundle Report\n") and "## Taps" in idx.read_text()

    mono = tmp_path / "mono"
    (mono / "taps").mkdir(parents=True)
    x = synth_ir(5, 0, 20000, rt60_seconds=0.1)
    pcm = O.recorder_float_to_pcm16(x)
    for name in ("m1", "m2"):
        wavfile.write(str(mono / "taps" / f"{name}.wav"), SR, pcm)
    (mono / "meta.json").write_text(json.dumps({"sample_rate_hz": SR, "length_samples": 20000, "taps": ["m1", "m2"]}))
    labels, rec = bundle.run_bundle_metrics(mono)
    assert labels == [("m1", "mono"), ("m2", "mono")] and rec[0].tobytes() == rec[1].tobytes()
    d = O.analyse_decay(O.pcm_to_float32(pcm))
    assert rec[0, P.M_START] == d["start"]
    os.environ["IRA_TAPS_PER_BATCH"] = "2"
    try:
        # PNG rendering is inline here (the reference's behaviour); keep the run short: two 0.4 s taps
        cli.main(["bundle", "--input", str(mono), "--reports-subdir", "rep2"])
    finally:
        os.environ.pop("IRA_TAPS_PER_BATCH", None)
    assert capsys.readouterr().out.strip() == f"Wrote bundle report index: {mono / 'rep2' / 'bundle_report.md'}"
    md = (mono / "rep2" / "m1" / "m1_report.md").read_text()
    single = rp.run_report_from_wav_file(mono / "taps" / "m1.wav", tmp_path / "s" / "m1",
                                         rp.ReportSettings(render_plots=False))
    assert md.replace(str(mono / "taps" / "m1.wav"), "@") == single.summary_markdown.replace(str(mono / "taps" / "m1.wav"), "@")
    assert (mono / "rep2" / "m2" / "m2_spectrogram_mono.png").stat().st_size > 1000


def test_host_pull_upload_is_bit_exact_and_feeds_the_pipeline():
    """ira_host_pull (the batch upload as a kernel reading pinned host memory) against plain copies: float32 bit for bit,
    PCM16 converted exactly like ira_pcm16_to_channels (all 65536 codes, odd counts: the scalar tail), and DeviceFeed /
    run_pipelined hand the same records back for pull and copy-engine uploads and both wire formats' own references."""
    import torch
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.feed import DeviceFeed, HostBatch, run_pipelined
    from audio_analysis_amd.pipeline import FullReport, FullReportSettings
    from audio_analysis_amd.synth import synth_ir
    from dataclasses import replace
    eng = get_engine()
    rng = np.random.default_rng(9)
    for count in (1, 3, 4, 7, 8, 1000, 65536 + 5):
        f = rng.standard_normal(count).astype(np.float32)
        hp = torch.empty(count, dtype=torch.float32, pin_memory=True); hp.numpy()[:] = f
        out = eng.empty(count + 8, torch.float32)
        out.fill_(7.0)
        assert eng.lib.ira_host_pull(int(hp.data_ptr()), count, 0, int(out.data_ptr()), 0, eng.stream) == 0
        got = out.cpu().numpy()
        assert np.array_equal(got[:count].view(np.uint32), f.view(np.uint32)) and np.all(got[count:] == 7.0)
    codes = np.arange(-32768, 32768, dtype=np.int16)
    pcm = np.concatenate([codes, codes[::-1], codes[:5]])
    hp = torch.empty(pcm.size, dtype=torch.int16, pin_memory=True); hp.numpy()[:] = pcm
    out = eng.empty(pcm.size, torch.float32)
    assert eng.lib.ira_host_pull(int(hp.data_ptr()), pcm.size, 1, int(out.data_ptr()), 3, eng.stream) == 0
    ref = np.clip(pcm.astype(np.float32) / np.float32(32768.0), -1.0, 1.0).astype(np.float32)
    assert np.array_equal(out.cpu().numpy().view(np.uint32), ref.view(np.uint32))
    # not pinned / not host memory -> refused, no launch
    assert eng.lib.ira_host_pull(int(out.data_ptr()) + 4, 16, 0, int(out.data_ptr()), 0, eng.stream) in (-2, -3)
    # the feed: ragged batches, pull vs copy engine, float32 and PCM16
    settings = replace(FullReportSettings(), run_rt60_bands=False, run_modal_cloud=False, run_zplane=False,
                       run_waterfall=False, run_filter=False)
    rep = FullReport(eng, settings)
    chans = [[synth_ir(300 + 4 * b + i, 0, 30000 + 1111 * i + 7 * b, rt60_seconds=0.08) for i in range(4)] for b in range(5)]
    rows = {}
    for pull in (True, False):
        for pcm16 in (False, True):
            feed = DeviceFeed(eng, 200000, depth=4, pull=pull)
            hbs = [HostBatch(eng, [(c * np.float32(32767.0)).astype(np.int16) for c in cs] if pcm16 else cs, pcm16=pcm16)
                   for cs in chans]
            got = []
            assert run_pipelined(rep, feed, hbs, got.append) == len(chans)
            rows[(pull, pcm16)] = np.concatenate(got)
    for pcm16 in (False, True):
        a, b = rows[(True, pcm16)], rows[(False, pcm16)]
        assert a.shape == (20, rows[(True, False)].shape[1]) and a.tobytes() == b.tobytes()
    direct = np.concatenate([rep.run(eng.upload(cs)) for cs in chans])
    assert direct.tobytes() == rows[(True, False)].tobytes()
