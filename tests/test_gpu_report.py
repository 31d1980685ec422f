"""GPU: report Markdown vs the reference's (golden), bundle runner, CLI commands, pipeline shard invariance."""
import json
from pathlib import Path

import numpy as np
import pytest
from scipy.io import wavfile

from oracle import ira_oracle as O

pytestmark = pytest.mark.gpu
SR = 48000


def _write(tmp_path, golden, name):
    g, _, _ = golden
    pcm = g[f"report/{name}/pcm"]
    p = tmp_path / f"{name}.wav"
    wavfile.write(str(p), SR, pcm[:, 0] if (pcm.ndim == 2 and pcm.shape[1] == 1) else pcm)
    return p


@pytest.mark.parametrize("name,variant", [("stereo16", "default"), ("stereo16", "monomix"), ("mono16", "default"),
                                          ("stereof32", "default"), ("stereof32", "monomix"),
                                          ("stereo16", "full"), ("stereo16", "fullmix"), ("mono16", "full"),
                                          ("stereo16", "full+png"), ("mono16", "full+png")])
def test_report_markdown_matches_reference(tmp_path, golden, name, variant):
    """`full*` = the reference's literal default report minus the IR waveform plots: group delay and diffusion on.
    Without PNGs the text comes from the small device records (the curves and matrices stay in HBM); `+png` runs the
    path that brings the arrays to the host for plotting -- both must give the reference's Markdown."""
    from audio_analysis_amd.analyse import report as rp
    _, c, _ = golden
    wav = _write(tmp_path, golden, name)
    png = variant.endswith("+png")
    variant = variant.replace("+png", "")
    kw = dict(run_impulse_response_plots=False, render_plots=png)
    if variant in ("default", "monomix"):
        kw.update(run_group_delay=False, run_diffusion=False)
    if variant in ("monomix", "fullmix"):
        kw.update(common_use_mono_downmix_for_stereo=True, common_ignore_leading_seconds=0.003)
    res = rp.run_report_from_wav_file(wav, tmp_path / f"out_{variant}" / "rep", rp.ReportSettings(**kw))
    want = c["report"][f"{name}/{variant}"]["markdown"].replace("{WAV}", str(wav))
    assert res.summary_markdown == want
    assert res.summary_markdown_path.read_text() == want


@pytest.mark.parametrize("name,tag,kw", [
    ("stereo16", "literal", {}), ("stereo16", "literal_mono", dict(common_use_mono_downmix_for_stereo=True)),
    ("mono16", "literal", {})])
def test_literal_default_report_markdown_and_png_set(tmp_path, name, tag, kw):
    """`ReportSettings()` exactly as `python -m analyse.cli report` builds it: every block on, impulse-response plots
    included, PNGs rendered.  Markdown string-identical to the reference's (tests/golden/report_literal.json, made by
    tests/golden/make_literal_report_golden.py) and the SAME SET of PNG files written."""
    import json as _json
    from audio_analysis_amd.analyse import report as rp
    lit = _json.loads((Path(__file__).resolve().parent / "golden" / "report_literal.json").read_text())
    g = np.load(Path(__file__).resolve().parent / "golden" / "goldens.npz")
    pcm = g[f"report/{name}/pcm"]
    wav = tmp_path / f"{name}.wav"
    wavfile.write(str(wav), SR, pcm[:, 0] if (pcm.ndim == 2 and pcm.shape[1] == 1) else pcm)
    res = rp.run_report_from_wav_file(wav, tmp_path / "out" / "rep", rp.ReportSettings(**kw))
    assert res.summary_markdown == lit[f"{name}/{tag}"].replace("{WAV}", str(wav))
    pngs = sorted(p.name for p in (tmp_path / "out").glob("*.png"))
    assert pngs == lit[f"{name}/{tag}/pngs"]
    assert all((tmp_path / "out" / p).stat().st_size > 1000 for p in pngs)
    assert "Skipped blocks" not in res.summary_markdown


def test_ir_command_writes_the_three_views(tmp_path, golden, capsys):
    from audio_analysis_amd.analyse import cli
    wav = _write(tmp_path, golden, "stereo16")
    cli.main(["ir", "--input", str(wav), "--output", str(tmp_path / "v" / "take.1"), "--no_show"])
    assert capsys.readouterr().out == ""                                  # the reference prints nothing for `ir`
    # the reference derives the names with Path.with_suffix: a basename with a dot loses what follows it
    assert sorted(p.name for p in (tmp_path / "v").glob("*.png")) == ["take.png", "take_early.png", "take_tail.png"]


def test_zplane_and_filter_commands_match_reference(tmp_path, golden, capsys):
    from audio_analysis_amd.analyse import cli
    g, c, _ = golden
    wav = _write(tmp_path, golden, "stereo16")
    cli.main(["zplane", "--input", str(wav), "--ar-order", "32", "--no-show"])
    out = capsys.readouterr().out
    assert out.strip() == c["report"]["stereo16/zplane32"]["summary"]
    cli.main(["filter", "--input", str(wav), "--no_show"])
    out = capsys.readouterr().out
    assert out.strip() == c["report"]["stereo16/filter"]["summary"]
    cli.main(["groupdelay", "--input", str(wav), "--no-show"])
    assert capsys.readouterr().out.strip() == c["report"]["stereo16/groupdelay"]["summary"]
    cli.main(["diffusion", "--input", str(wav), "--no_show"])
    assert capsys.readouterr().out.strip() == c["report"]["stereo16/diffusion"]["summary"]
    cli.main(["decay", "--input", str(wav), "--no_show"])          # CLI default: compute_edt=True
    out = capsys.readouterr().out
    assert "[left] analysis_start_sample_index=245" in out and "EDT:" in out


def test_bundle_runner(tmp_path, golden):
    from audio_analysis_amd.analyse import bundle, report as rp
    g, _, _ = golden
    root = tmp_path / "bundle"
    (root / "taps").mkdir(parents=True)
    pcm = g["report/stereo16/pcm"]
    for tap in ("in", "out"):
        wavfile.write(str(root / "taps" / f"{tap}.wav"), SR, pcm if tap == "in" else (pcm // 2).astype(np.int16))
    (root / "meta.json").write_text(json.dumps({"sample_rate_hz": SR, "length_samples": int(pcm.shape[0]),
                                                "taps": ["in", "out"]}))
    rs = rp.ReportSettings(run_impulse_response_plots=False, run_group_delay=False, run_diffusion=False, render_plots=False)
    idx = bundle.run_bundle_report(root, bundle.BundleRunSettings(report_settings=rs))
    text = idx.read_text()
    assert text.startswith("# IR Bundle Report\n") and "- [in](reports/in/in_report.md)" in text
    assert (root / "reports" / "out" / "out_report.md").exists()


def test_pipeline_records_are_shard_invariant():
    """the same files analysed as one batch or as shards cut ANYWHERE -- odd positions included --
    give byte-identical metric records, whoever the neighbours are: a DC channel and a silent one sit between ordinary
    responses (round 2 paired two channels in one complex transform, so a channel's last bits depended on its partner and
    these two were the adversarial partners; now a channel's transforms are its own), lengths are mixed (equal smooth
    lengths that used to pair, odd and even data-dependent ones)."""
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.dist import shard_files
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    n = 60000
    chans = [synth_ir(i, 0, n, rt60_seconds=0.3 + 0.05 * i) for i in range(3)]
    chans += [np.full(n, 0.25, np.float32), np.zeros(n, np.float32)]                    # DC, digital silence
    chans += [synth_ir(10 + i, 0, n, rt60_seconds=0.2 + 0.1 * i, pre_delay=240 + i) for i in range(3)]
    chans += [synth_ir(20, 0, 48001, rt60_seconds=0.2), synth_ir(21, 0, 50000, rt60_seconds=0.25)]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        full = P.FullReport(eng).run(eng.upload(chans))
        assert np.all(full[:, P.M_STATUS] == 0)
        for cuts in ([1], [3], [4], [5], [2, 7], [1, 2, 3, 4, 5, 6, 7, 8, 9]):
            edges = [0] + cuts + [len(chans)]
            parts = [P.FullReport(eng).run(eng.upload(chans[a:b])) for a, b in zip(edges[:-1], edges[1:])]
            assert np.concatenate(parts, axis=0).tobytes() == full.tobytes(), cuts
        # a different ORDER of the same channels: every record unchanged
        perm = [4, 0, 9, 3, 7, 1, 8, 2, 6, 5]
        shuffled = P.FullReport(eng).run(eng.upload([chans[i] for i in perm]))
        assert shuffled.tobytes() == full[perm].tobytes()
        # the number of HIP streams the report blocks are dealt onto (one, the default two, round 2's three) is scheduling
        # only: the same bytes
        saved = eng.num_lanes
        try:
            for lanes in (1, 2, 3):
                eng.num_lanes = lanes
                assert P.FullReport(eng).run(eng.upload(chans)).tobytes() == full.tobytes(), lanes
        finally:
            eng.num_lanes = saved
    parts = []
    for r in range(2):
        lo, hi = shard_files(len(chans), r, 2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            parts.append(P.FullReport(eng).run(eng.upload(chans[lo:hi])))
    assert np.concatenate(parts, axis=0).tobytes() == full.tobytes()
    # and the records agree with the oracle
    for i in (0, 1, 2, 5):
        x = chans[i]
        d = O.analyse_decay(x)
        assert full[i, P.M_START] == d["start"]
        assert abs(full[i, P.M_FIT_T30 + 6] - d["fits"]["T30"]["rt60"]) < 1e-6 * d["fits"]["T30"]["rt60"]
        z = O.analyse_zplane(x, ar_order=64)
        assert abs(full[i, P.M_AR_MEDIAN_R] - z["median_radius"]) < 1e-8


# ---------------------------------------------------------------------------------------- section 8f rank 3
def test_batched_reports_equal_reference_markdown_for_a_mixed_group(tmp_path, golden):
    """One device batch holding a stereo PCM16 file, a mono PCM16 file and a float32 file (Python-decoded): every
    file's Markdown is the reference's literal default report (group delay + diffusion on), string for string."""
    from audio_analysis_amd.analyse import report as rp
    _, c, _ = golden
    names = ["stereo16", "mono16", "stereof32"]
    wavs = [_write(tmp_path, golden, n) for n in names]
    rs = rp.ReportSettings(run_impulse_response_plots=False, render_plots=False)
    out = rp.run_reports_batched([(w, tmp_path / "o" / n / "rep") for w, n in zip(wavs, names)], rs)
    for n, w, r in zip(names, wavs, out):
        key = f"{n}/full"
        if key in c["report"]:
            assert r.summary_markdown == c["report"][key]["markdown"].replace("{WAV}", str(w)), n
        single = rp.run_report_from_wav_file(w, tmp_path / "s" / n / "rep", rs)
        assert r.summary_markdown == single.summary_markdown
    # --mono policy across the same group
    rs = rp.ReportSettings(run_impulse_response_plots=False, render_plots=False, common_use_mono_downmix_for_stereo=True,
                           common_ignore_leading_seconds=0.003)
    out = rp.run_reports_batched([(w, tmp_path / "m" / n / "rep") for w, n in zip(wavs, names)], rs)
    for n, w, r in zip(names, wavs, out):
        key = f"{n}/fullmix"
        if key in c["report"]:
            assert r.summary_markdown == c["report"][key]["markdown"].replace("{WAV}", str(w)), n


def test_bundle_batched_with_plot_workers_and_reference_written_taps(tmp_path):
    """The bundle the reference's C++ recorder wrote: batched analysis + PNGs rendered by worker processes give the
    same Markdown as tap-by-tap inline rendering, and every linked PNG that the writer produces exists."""
    import shutil
    from audio_analysis_amd.analyse import bundle, report as rp
    gold = Path(__file__).resolve().parent / "golden" / "bundle"
    a, b = tmp_path / "a", tmp_path / "b"
    shutil.copytree(gold, a)
    shutil.copytree(gold, b)
    rs = rp.ReportSettings(run_impulse_response_plots=False, run_waterfall=False)
    bundle.run_bundle_report(a, bundle.BundleRunSettings(report_settings=rs, taps_per_batch=8, plot_workers=2))
    bundle.run_bundle_report(b, bundle.BundleRunSettings(report_settings=rs, taps_per_batch=1, plot_workers=0))
    for tap in ("early", "late_hot"):
        ta = (a / "reports" / tap / f"{tap}_report.md").read_text()
        tb = (b / "reports" / tap / f"{tap}_report.md").read_text()
        assert ta.replace(str(a), "@") == tb.replace(str(b), "@")
        for suffix in ("_decay", "_rt60bands", "_fr", "_groupdelay_left", "_spectrogram_right", "_diffusion",
                       "_modalcloud_left"):
            assert (a / "reports" / tap / f"{tap}{suffix}.png").stat().st_size > 1000, suffix
    assert (a / "reports" / "bundle_report.md").read_text().replace(str(a), "@") == \
        (b / "reports" / "bundle_report.md").read_text().replace(str(b), "@")


def test_bundle_aborts_at_the_first_bad_tap_like_the_reference(tmp_path):
    from audio_analysis_amd.analyse import bundle, report as rp
    from audio_analysis_amd.synth import synth_ir
    root = tmp_path / "bundle"
    (root / "taps").mkdir(parents=True)
    good = np.stack([synth_ir(3, c, 20000, rt60_seconds=0.1) for c in (0, 1)], axis=1)
    for tap, st in (("a_ok", good), ("b_short", good[:3000]), ("c_ok", good)):
        (root / "taps" / f"{tap}.wav").write_bytes(O.recorder_wav_bytes(st))
    (root / "meta.json").write_text(O.recorder_meta_json(SR, 20000, ["a_ok", "b_short", "c_ok"]))
    rs = rp.ReportSettings(run_impulse_response_plots=False, render_plots=False)
    with pytest.raises(ValueError):                           # 3000 samples < n_fft of the spectrogram
        bundle.run_bundle_report(root, bundle.BundleRunSettings(report_settings=rs, taps_per_batch=8))
    assert (root / "reports" / "a_ok" / "a_ok_report.md").exists()
    assert not (root / "reports" / "c_ok" / "c_ok_report.md").exists()
    assert not (root / "reports" / "bundle_report.md").exists()


def test_per_ir_status_one_bad_channel_does_not_poison_the_batch():
    """a channel the reference would refuse (ValueError: too few samples for a block) comes back with the
    status code of the FIRST block that refuses it and NaN metrics; every other channel's record is byte-identical to
    what it is in a batch without the bad ones."""
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    good = [synth_ir(600 + i, 0, 30000 + 500 * i, rt60_seconds=0.1) for i in range(3)]
    short_stft = synth_ir(610, 0, 3000, rt60_seconds=0.02, pre_delay=10)        # fine for decay/bands/fr, < n_fft 4096
    tiny = np.array([0.0, 0.1, 0.2, 1.0, 0.5], np.float32)                      # 2 samples after the peak: decay refuses
    seven = np.array([1.0, 0.5, 0.25, 0.1, 0.05, 0.02, 0.01], np.float32)       # decay takes it, rt60bands (< 8) refuses
    rep = P.FullReport(eng)
    mixed = rep.run(eng.upload([good[0], short_stft, good[1], tiny, seven, good[2]]))
    alone = rep.run(eng.upload(good))
    assert list(mixed[:, P.M_STATUS]) == [0, P.ST_SPECTROGRAM_TOO_SHORT, 0, P.ST_DECAY_TOO_SHORT, P.ST_BANDS_TOO_SHORT, 0]
    assert mixed[[0, 2, 5]].tobytes() == alone.tobytes()
    for row in (1, 3, 4):
        assert np.all(np.isnan(mixed[row, 2:])) and mixed[row, P.M_NSAMPLES] in (3000, 5, 7)
    assert P.STATUS_MESSAGES[int(mixed[3, P.M_STATUS])].startswith("Not enough samples after trimming/ignoring")
    # a batch of ONLY refused channels launches nothing and still reports
    none = rep.run(eng.upload([tiny, seven]))
    assert list(none[:, P.M_STATUS]) == [P.ST_DECAY_TOO_SHORT, P.ST_BANDS_TOO_SHORT]
    # the drop-in single-channel API keeps the reference's exception
    from audio_analysis_amd.analyse import spectrogram
    with pytest.raises(ValueError, match="Not enough samples"):
        spectrogram.analyse_spectrogram_for_channel(short_stft, 48000, "m", spectrogram.SpectrogramAnalysisSettings())


def _degenerate_inputs():
    from audio_analysis_amd.synth import synth_ir
    n = 48000
    base = synth_ir(11, 0, n, rt60_seconds=0.2)
    nan = base.copy(); nan[1234] = np.nan
    inf = base.copy(); inf[20000] = np.inf
    last = np.zeros(n, np.float32); last[-1] = 1.0
    return dict(zeros=np.zeros(n, np.float32), dc=np.full(n, 0.25, np.float32), last=last, nan=nan, inf=inf)


def _num(v):
    return None if v is None else float(v)          # "nan" / "inf" strings of the golden file parse as floats


def test_degenerate_inputs_match_the_reference():
    """Digital silence, a DC offset, all the energy in the last sample, a NaN and an infinite sample: what the REFERENCE
    returns or raises for each block (tests/golden/degenerate.json, written by make_degenerate_goldens.py) against the
    device path -- drop-in functions for the curves, the batched pipeline for records and per-IR status."""
    import warnings
    from dataclasses import replace
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.analyse import decay, spectrogram, waterfall
    from audio_analysis_amd.engine import get_engine
    gold = json.loads((Path(__file__).resolve().parent / "golden" / "degenerate.json").read_text())
    xs = _degenerate_inputs()
    eng = get_engine()
    names = ["zeros", "dc", "nan", "inf"]
    rep = P.FullReport(eng, replace(P.FullReportSettings(), run_zplane=False,
                                    decay=replace(P.FullReportSettings().decay, compute_edt=True)))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = rep.run(eng.upload([xs[k] for k in names]))
    for i, tag in enumerate(names):
        g = gold[tag]
        row = m[i]
        assert row[P.M_STATUS] == 0, tag
        assert int(row[P.M_START]) == g["decay"]["start"], tag
        early = _num(g["decay"]["early"])
        assert (early is None) == bool(np.isnan(row[P.M_EARLY10])), tag
        if early is not None:
            assert abs(row[P.M_EARLY10] - early) <= 1e-6 * early, tag
        for name, slot in (("EDT", P.M_FIT_EDT), ("T20", P.M_FIT_T20), ("T30", P.M_FIT_T30)):
            ref = g["decay"]["fits"].get(name)
            assert (ref is None) == (row[slot] != 1.0), (tag, name, row[slot : slot + 8])
            if ref is not None:
                assert abs(row[slot + 6] - ref[0]) <= 1e-6 * abs(ref[0]), (tag, name)
                assert abs(row[slot + 5] - ref[2]) <= 1e-9, (tag, name)
        for k, band in enumerate(("Low", "Mid", "High")):
            for j in range(3):
                ref = _num(g["bands"][band][j])
                got = row[P.M_BANDS + 3 * k + j]
                assert (ref is None) == bool(np.isnan(got)), (tag, band, j, got)
                if ref is not None:
                    assert abs(got - ref) <= 1e-4 * abs(ref), (tag, band, j)
        assert row[P.M_FR_PEAK] == _num(g["fr"]["peak"]), tag
        assert abs(row[P.M_FR_CENTROID] - _num(g["fr"]["centroid"])) <= 1e-9 * _num(g["fr"]["centroid"]), tag
        m1k = _num(g["filter"]["mag_1k"])
        assert (np.isnan(m1k) and np.isnan(row[P.M_FILT_1K])) or abs(row[P.M_FILT_1K] - m1k) <= 2e-5, tag
        assert int(row[P.M_SPEC_FRAMES]) == g["spectrogram"]["shape"][1], tag
        assert int(row[P.M_MODAL_POINTS]) == g["modal"]["points"], tag
    # "last": one sample after the peak -- the reference's decay block raises, so does every later block
    only = rep.run(eng.upload([xs["last"], xs["dc"]]))
    assert only[0, P.M_STATUS] == P.ST_DECAY_TOO_SHORT and np.all(np.isnan(only[0, 2:]))
    # (round 2: "the DC channel shared a transform with the silent one above and runs alone here ... hence a tolerance
    # instead of bytes"; a channel's transforms are its own now)
    assert only[1].tobytes() == m[1].tobytes()
    with pytest.raises(ValueError, match=gold["last"]["decay"]["message"][:40]):
        decay.analyse_decay_for_channel(xs["last"], 48000, "m", decay.DecayAnalysisSettings())
    # curves: NaN / infinity patterns of the EDC, the spectrogram and the waterfall slices
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tag in ("zeros", "nan", "inf"):
            g = gold[tag]
            r = decay.analyse_decay_for_channel(xs[tag], 48000, "m", decay.DecayAnalysisSettings(compute_edt=True))
            e = r.edc_db
            assert e.size == g["decay"]["edc_len"] and int(np.isnan(e).sum()) == g["decay"]["edc_nan"], tag
            first, lastv = _num(g["decay"]["edc_first"]), _num(g["decay"]["edc_last"])
            assert (np.isnan(first) and np.isnan(e[0])) or e[0] == first, tag
            assert (np.isnan(lastv) and np.isnan(e[-1])) or e[-1] == lastv, tag
            sp = spectrogram.analyse_spectrogram_for_channel(xs[tag], 48000, "m", spectrogram.SpectrogramAnalysisSettings())
            mm = sp.magnitude_db
            assert list(mm.shape) == g["spectrogram"]["shape"] and int(np.isnan(mm).sum()) == g["spectrogram"]["nan"], tag
            if g["spectrogram"]["nan"]:
                assert np.all(np.isnan(mm[:, 0])) and not np.any(np.isnan(mm[:, 1:])), tag     # the frame holding the bad sample
            assert abs(np.nanmax(mm) - _num(g["spectrogram"]["max"])) <= 1e-3, tag
            wf = waterfall.analyse_waterfall_for_channel(xs[tag], 48000, "m", waterfall.WaterfallAnalysisSettings())
            assert list(wf.slice_magnitude_rel_db.shape) == g["waterfall"]["shape"], tag
            assert int(np.isnan(wf.slice_magnitude_rel_db).sum()) == g["waterfall"]["nan"], tag


def test_ragged_batches_through_the_feed_on_one_stream():
    """Round-2 advisor finding: on ONE stream (Engine.num_lanes = 1) a DeviceFeed batch's offset / length tables live in
    a block allocated on the feed's copy stream; run_pipelined drops the batch before finish(), so the next push could
    recycle that block under the step's queued kernels.  Ragged batches make that visible: every step's records must
    equal the records of the same channels analysed on their own."""
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.feed import DeviceFeed, HostBatch, run_pipelined
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    groups = [[synth_ir(900 + 10 * g + i, 0, 20000 + 1777 * ((3 * g + i) % 5), rt60_seconds=0.12) for i in range(3 + g % 2)]
              for g in range(6)]
    rep = P.FullReport(eng)
    want = [rep.run(eng.upload(chs)) for chs in groups]
    lanes = eng.num_lanes
    eng.num_lanes = 1
    try:
        feed = DeviceFeed(eng, max(sum(c.size for c in chs) for chs in groups), depth=4)
        got = []
        run_pipelined(rep, feed, [HostBatch(eng, chs) for chs in groups], got.append)
    finally:
        eng.num_lanes = lanes
    assert len(got) == len(want)
    for g, (a, b) in enumerate(zip(got, want)):
        assert a.tobytes() == b.tobytes(), g


def test_feed_times_its_uploads_and_picks_an_upload_method():
    """Round 5: DeviceFeed.timing brackets every upload piece with events on its copy stream (what bench.py reports as
    upload_ms_per_step / h2d_*_GBps), and DeviceFeed.autotune tries two pieces, one piece and the pull kernel under the job's
    own kernels, keeps one of them -- and the records do not depend on which (the upload method moves bytes, nothing else)."""
    import time
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.feed import DeviceFeed, HostBatch, run_pipelined
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    chans = [synth_ir(1300 + i, 0, 48000, rt60_seconds=0.2) for i in range(4)]
    hb = [HostBatch(eng, np.stack(chans)) for _ in range(2)]
    rep = P.FullReport(eng)
    want = rep.run(eng.upload(chans))
    feed = DeviceFeed(eng, 4 * 48000, depth=4)
    feed.split_bytes = 1 << 10                           # (these small batches would go as one copy otherwise)
    feed.timing = []
    got = []
    run_pipelined(rep, feed, [hb[i % 2] for i in range(4)], got.append)
    eng.torch.cuda.synchronize()
    up = feed.upload_times()
    assert up["uploads"] == 4 and up["pieces_per_upload"] == 2 and up["bytes_per_upload"] == 4 * 48000 * 4
    assert up["ms_per_upload"] > 0.0 and up["GBps"] > 0.01 and 0.0 < up["busy_fraction_of_stretch"] <= 1.0
    feed.timing = None
    assert all(r.tobytes() == want.tobytes() for r in got)

    def steps(c):
        t0 = time.perf_counter()
        run_pipelined(rep, feed, [hb[i % 2] for i in range(c)], got.append)
        eng.torch.cuda.synchronize()
        return time.perf_counter() - t0

    tried = feed.autotune(steps, alone_GBps=1e9, steps=2)             # an unreachable link rate: every arm is tried
    assert len(tried) == 3 and sum(1 for r in tried if r["chosen"]) == 1
    assert {r["mode"] for r in tried} == {"copy engine, 2 piece(s)", "copy engine, 1 piece(s)", "pull kernel"}
    assert feed.mode() == next(r["mode"] for r in tried if r["chosen"])
    steps(2)
    assert all(r.tobytes() == want.tobytes() for r in got)


def test_filter_block_reads_the_raw_spectrum_when_fr_smoothing_is_on():
    """Round-2 advisor finding: with frequency-response log smoothing on, the fr block smooths its dB curve in place on
    the device; the filter block (reference filterplot.py: no smoothing) must not share that curve."""
    from dataclasses import replace
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    chans = [synth_ir(700 + i, 0, 30000 + 999 * i, rt60_seconds=0.15) for i in range(3)]
    base = P.FullReportSettings()
    st = replace(base, frequency_response=replace(base.frequency_response, smoothing_log_bins=9))
    m = P.FullReport(eng, st).run(eng.upload(chans))
    plain = P.FullReport(eng, base).run(eng.upload(chans))
    for i, x in enumerate(chans):
        f = O.analyse_filter_response(x)
        assert abs(m[i, P.M_FILT_1K] - f["mag_1k_db"]) <= 2e-5 and m[i, P.M_FILT_PEAK] == f["peak_hz"]
        r = O.analyse_frequency_response(x, smoothing_log_bins=9)
        assert m[i, P.M_FR_PEAK] == r["peak_hz"]
        assert abs(m[i, P.M_FR_CENTROID] - r["centroid_hz"]) <= 1e-6 * r["centroid_hz"]
    assert np.array_equal(m[:, P.M_FILT_1K], plain[:, P.M_FILT_1K])
    assert not np.array_equal(m[:, P.M_FR_CENTROID], plain[:, P.M_FR_CENTROID])
