"""
CPU tests for the native tap ingest: the oracle's restatement of the recorder's on-disk
format against bytes the reference's own C++ recorder wrote (tests/golden/bundle, made by
tests/golden/make_bundle_fixture.py), and libira's HOST-side RIFF walker / PCM16 reader (no device calls).
"""
import struct
from pathlib import Path

import numpy as np
import pytest

from oracle import ira_oracle as O

GOLD = Path(__file__).resolve().parent / "golden"
TAPS = ["early", "late_hot"]


@pytest.fixture(scope="module")
def expected():
    return np.load(GOLD / "bundle_expected.npz")


def test_oracle_recorder_format_is_byte_exact(expected):
    for name in TAPS:
        blob = (GOLD / "bundle" / "taps" / f"{name}.wav").read_bytes()
        assert O.recorder_wav_bytes(expected[f"{name}/input"]) == blob
    assert O.recorder_meta_json(48000, 12000, reversed(TAPS)) == (GOLD / "bundle" / "meta.json").read_text()


def test_oracle_reader_and_channel_policy_match_reference_loader(expected):
    for name in TAPS:
        sr, raw = O.wav_pcm16_payload((GOLD / "bundle" / "taps" / f"{name}.wav").read_bytes())
        assert sr == 48000 and raw.shape == (12000, 2) and raw.dtype == np.int16
        f = O.pcm_to_float32(raw)
        for mono, key in ((False, "split"), (True, "mix")):
            for ch, x in O.analysis_channels(f, mono):
                ref = expected[f"{name}/{key}/{ch}"]
                assert x.dtype == np.float32 and np.array_equal(x, ref)
    # the hot tap really exercises the recorder's clamp
    _, raw = O.wav_pcm16_payload((GOLD / "bundle" / "taps" / "late_hot.wav").read_bytes())
    assert raw.max() == 32767 and raw.min() == -32767


def test_native_probe_and_read_match_oracle():
    from audio_analysis_amd.ingest import probe_tap, read_tap_pcm16
    for name in TAPS:
        p = GOLD / "bundle" / "taps" / f"{name}.wav"
        info = probe_tap(p)
        assert (info.native, info.sample_rate_hz, info.channels, info.frames, info.data_offset) == (True, 48000, 2, 12000, 44)
        _, raw = O.wav_pcm16_payload(p.read_bytes())
        assert np.array_equal(read_tap_pcm16(info), raw)


def _wav(fmt_body: bytes, data: bytes, extra_before: bytes = b"", extra_after: bytes = b"") -> bytes:
    body = b"WAVE" + extra_before + b"fmt " + struct.pack("<I", len(fmt_body)) + fmt_body + extra_after
    body += b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", len(body)) + body


def test_native_probe_edge_cases(tmp_path):
    from audio_analysis_amd.ingest import probe_tap, read_tap_pcm16
    pcm = np.arange(-7, 8, dtype="<i2")                                   # 15 mono samples
    fmt16 = struct.pack("<HHIIHH", 1, 1, 48000, 96000, 2, 16)
    # extra chunks around fmt, one of odd size (word-aligned padding), mono
    odd = b"LIST" + struct.pack("<I", 5) + b"abcde" + b"\0"
    p = tmp_path / "mono_list.wav"
    p.write_bytes(_wav(fmt16, pcm.tobytes(), extra_before=odd, extra_after=b"fact" + struct.pack("<II", 4, 15)))
    info = probe_tap(p)
    assert (info.native, info.channels, info.frames, info.sample_rate_hz) == (True, 1, 15, 48000)
    assert np.array_equal(read_tap_pcm16(info).reshape(-1), pcm)
    # WAVE_FORMAT_EXTENSIBLE carrying PCM16
    ext = struct.pack("<HHIIHH", 0xFFFE, 2, 44100, 44100 * 4, 4, 16) + struct.pack("<HHI", 22, 16, 3)
    ext += struct.pack("<H", 1) + b"\x00\x00\x00\x00\x10\x00\x80\x00\x00\xaa\x00\x38\x9b\x71"
    p = tmp_path / "ext.wav"
    p.write_bytes(_wav(ext, np.arange(8, dtype="<i2").tobytes()))
    info = probe_tap(p)
    assert (info.native, info.channels, info.frames, info.sample_rate_hz) == (True, 2, 4, 44100)
    # float32 file: valid WAV, not native -> header facts still reported for validation
    f32 = struct.pack("<HHIIHH", 3, 2, 48000, 48000 * 8, 8, 32)
    p = tmp_path / "float.wav"
    p.write_bytes(_wav(f32, np.zeros(10, "<f4").tobytes()))
    info = probe_tap(p)
    assert (info.native, info.channels, info.frames) == (False, 2, 5)
    with pytest.raises(ValueError):
        read_tap_pcm16(info)
    # empty data chunk, truncated payload, garbage, missing file
    p = tmp_path / "empty.wav"
    p.write_bytes(_wav(fmt16, b""))
    assert probe_tap(p).frames == 0
    p = tmp_path / "short.wav"
    blob = _wav(fmt16, pcm.tobytes())
    p.write_bytes(blob[:-6])
    info = probe_tap(p)                                                   # truncated payload: the reference's reader returns
    assert info.frames == 12                                              # the samples the file holds (golden test below)
    assert np.array_equal(read_tap_pcm16(info).reshape(-1), pcm[:12])
    p = tmp_path / "junk.wav"
    p.write_bytes(b"not a wav file at all, sorry")
    with pytest.raises(ValueError):                                       # not RIFF/WAVE: scipy's reader raises ValueError
        probe_tap(p)
    with pytest.raises(FileNotFoundError):
        probe_tap(tmp_path / "missing.wav")
    # data before fmt is malformed
    p = tmp_path / "nofmt.wav"
    p.write_bytes(b"RIFF" + struct.pack("<I", 12 + 8) + b"WAVE" + b"data" + struct.pack("<I", 0))
    with pytest.raises((ValueError, OSError)):
        probe_tap(p)


def test_truncated_taps_read_like_the_reference_reader(tmp_path):
    """tests/golden/truncated_wav.json (made by make_truncated_wav_golden.py with the reference's load_wav_file): a data
    chunk shorter than its header says yields the whole frames the file holds; samples that make no whole stereo frame
    raise ValueError (scipy's reshape)."""
    import json
    from audio_analysis_amd.ingest import probe_tap, read_tap_pcm16
    gold = json.loads((GOLD / "truncated_wav.json").read_text())
    for name, rec in gold.items():
        if name.startswith("_"):
            continue
        p = tmp_path / f"{name}.wav"
        p.write_bytes(bytes.fromhex(rec["file_hex"]))
        if "raises" in rec:
            with pytest.raises({"ValueError": ValueError, "OSError": OSError}[rec["raises"]]):
                probe_tap(p)
            continue
        info = probe_tap(p)
        assert [info.frames, info.channels] == rec["shape"], name
        want = np.frombuffer(bytes.fromhex(rec["samples_f32_hex"]), dtype="<f4").reshape(rec["shape"])
        got = O.pcm_to_float32(read_tap_pcm16(info)) if info.frames else np.zeros(rec["shape"], np.float32)
        assert np.array_equal(got, want), name


def test_ingest_validation_messages(tmp_path):
    """Wrong rate / channel count raise the reference's ValueError text (io.py:161-178) before any device work."""
    from audio_analysis_amd import ingest
    fmt = struct.pack("<HHIIHH", 1, 2, 44100, 44100 * 4, 4, 16)
    p = tmp_path / "r.wav"
    p.write_bytes(_wav(fmt, np.zeros(8, "<i2").tobytes()))
    with pytest.raises(ValueError, match=r"Expected sample rate 48000 Hz, but got 44100 Hz for file "):
        ingest._validate(ingest.probe_tap(p), 48000)
    fmt = struct.pack("<HHIIHH", 1, 3, 48000, 48000 * 6, 6, 16)
    p = tmp_path / "c.wav"
    p.write_bytes(_wav(fmt, np.zeros(9, "<i2").tobytes()))
    with pytest.raises(ValueError, match=r"Expected mono or stereo \(1 or 2 channels\) but got 3 channels for file "):
        ingest._validate(ingest.probe_tap(p), 48000)
    assert ingest.channel_names(2, False) == ["left", "right"] and ingest.channel_names(2, True) == ["mono"]
    assert ingest.channel_names(1, False) == ["mono"] and ingest.channel_names(1, True) == ["mono"]


def test_group_calls_equal_the_per_file_calls(tmp_path):
    """ira_wav_probe_batch / ira_wav_read_pcm16_batch (a whole group of taps per library call, threads inside the call):
    the same infos and payload bytes as the per-file entry points, for more files than threads, in group order; the first
    offending file of a group raises what it raises on its own (a missing tap FileNotFoundError, a non-RIFF file
    ValueError); an int32 file comes back as not native."""
    from audio_analysis_amd import ingest
    rng = np.random.default_rng(3)
    paths, payloads = [], []
    for i in range(37):
        frames, ch = int(rng.integers(1, 3000)), int(rng.integers(1, 3))
        pcm = rng.integers(-32768, 32768, size=(frames, ch), dtype=np.int16)
        p = tmp_path / f"t{i:02d}.wav"
        p.write_bytes(_wav16(pcm))
        paths.append(p); payloads.append(pcm)
    infos = ingest.probe_taps(paths)
    assert [i.path for i in infos] == paths
    for info, p, pcm in zip(infos, paths, payloads):
        assert info == ingest.probe_tap(p) and info.native and (info.frames, info.channels) == pcm.shape
    sizes = [(i.frames * i.channels + 1) & ~1 for i in infos]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
    buf = np.full(int(sum(sizes)), 12345, dtype=np.int16)
    ingest.read_taps_pcm16(infos, buf, offs.tolist())
    for info, o, pcm in zip(infos, offs, payloads):
        assert np.array_equal(buf[o : o + pcm.size].reshape(pcm.shape), pcm)
        assert np.array_equal(ingest.read_tap_pcm16(info), pcm)
    # errors: per-file types, first offender in group order
    bad = tmp_path / "notwav.wav"
    bad.write_bytes(b"this is not a RIFF file at all")
    with pytest.raises(FileNotFoundError):
        ingest.probe_taps(paths[:3] + [tmp_path / "missing.wav", bad])
    with pytest.raises(ValueError):
        ingest.probe_taps(paths[:3] + [bad, tmp_path / "missing.wav"])
    with pytest.raises(ValueError):
        ingest.read_taps_pcm16(infos[:2], np.zeros(4, dtype=np.int16), [0, 2])      # too small
    assert ingest.probe_taps([]) == []


def _wav16(pcm: np.ndarray) -> bytes:
    data = pcm.astype("<i2").tobytes()
    ch = pcm.shape[1]
    return (b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, ch, 48000, 48000 * 2 * ch, 2 * ch, 16)
            + b"data" + struct.pack("<I", len(data)) + data)
