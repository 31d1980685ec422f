"""
CPU-only tests (run with -m "not gpu"): the C-ABI library loads and exports every declared symbol, host-side
integer/index logic matches the oracle and the goldens, the CLI surface matches the reference's, the product
path refuses to run without a GPU, and the multi-rank metrics gather is rank-count invariant (gloo, world 2).
"""
import ctypes
import json
import os
import re
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

from oracle import ira_oracle as O

REPO = Path(__file__).resolve().parent.parent
SR = 48000


# ---------------------------------------------------------------------------------------- C-ABI
def _declared():
    text = (REPO / "include" / "ira.h").read_text()
    return sorted(set(re.findall(r"\b(ira_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    from audio_analysis_amd import _lib, build
    path = build.build()
    assert path.exists()
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ira.h but not exported"
        assert n in _lib.PROTOTYPES, f"{n} has no ctypes prototype"
    assert set(_lib.PROTOTYPES) == set(names)
    assert lib.ira_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define IRA_ABI_VERSION (\d+)", (REPO / "include" / "ira.h").read_text()).group(1))
    assert lib.ira_error_string(0) == b"ok" and b"NULL" in lib.ira_error_string(-1)
    assert lib.ira_ar_partial_doubles(64, 480000) == 59 * (64 * 64 + 64)   # ceil((480000-64)/8192) chunks
    assert lib.ira_ar_partial_doubles(64, 10) == 0


def test_stale_library_is_named_as_such(tmp_path):
    """a libira.so left over from an older ABI lacks the newer symbols.  The loader checks the version BEFORE it
    binds the prototypes, so the failure is the "rebuild it" message and not a bare AttributeError; the same for a library
    of the right version that lacks a symbol."""
    src = tmp_path / "stale.c"
    for tag, body in (("old", "int ira_abi_version(void) { return 1; }"),
                      ("holes", "int ira_abi_version(void) { return %d; }" % __import__("audio_analysis_amd._lib", fromlist=["x"]).ABI_VERSION),
                      ("none", "int unrelated(void) { return 0; }")):
        src.write_text(body + "\n")
        so = tmp_path / f"libira_{tag}.so"
        subprocess.run(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
        code = ("from audio_analysis_amd import _lib\n"
                "try:\n    _lib.load()\nexcept _lib.IraError as e:\n    print('IraError:', e)\n")
        env = dict(os.environ, IRA_TUNING="1", IRA_LIBRARY=str(so), PYTHONPATH=str(REPO))
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(REPO))
        assert r.returncode == 0, r.stderr[-2000:]
        assert "IraError:" in r.stdout and "audio_analysis_amd.build" in r.stdout, (tag, r.stdout, r.stderr[-500:])
        if tag == "old":
            assert "ABI version 1" in r.stdout
        if tag == "holes":
            assert "does not export" in r.stdout


def test_argument_validation_without_gpu():
    """Entry points reject bad arguments before touching the device (no compute calls here)."""
    from audio_analysis_amd import _lib
    lib = _lib.load()
    assert lib.ira_peak_index(0, 0, 0, 1, 16, 0, 0, 0) == -1                      # IRA_E_NULL
    assert lib.ira_stft_mag_db(1, 1, 1, 1, 1, 1000, 512, 1, 1, 32, -120.0, 1, 1, 0, 0, 0) == -2   # n_fft not a power of 2
    assert lib.ira_stft_mag_db(1, 1, 1, 1, 1, 4096, 512, 1, 1, 16, -120.0, 1, 1, 0, 0, 0) == -3  # precision
    assert lib.ira_poly_roots(1, 1, 5000, 1e-14, 1, 1, 0) == -2
    assert lib.ira_ar_gram(1, 0, 1, 1, 0, 1, 100, 2000, 1, 0, 0) == -2


def test_bluestein_convolution_sizes_and_their_four_step_split():
    """The sizes the host picks (2^k or 3 * 2^k) are sizes the library splits (n1 * n2 = M, n2 a power of two, n1 one or
    three times one); everything else is refused.  ira_fft_split is host-only code: no GPU needed."""
    import ctypes
    from audio_analysis_amd import _lib
    from audio_analysis_amd.engine import conv_size
    lib = _lib.load()
    a, b = ctypes.c_int32(0), ctypes.c_int32(0)
    for need in [1, 16, 17, 97 + 48, 1000, 4099 * 3 // 2, 65535 + 32767, 65537 + 32768, 2 * 49151 - 1, 2 * 239750 - 1,
                 479501 + 239750, 959999, (1 << 22) - 5]:
        for three in (True, False):
            m = conv_size(need, three)
            assert m >= max(need, 16)
            assert m & (m - 1) == 0 or (three and m % 3 == 0 and (m // 3) & (m // 3 - 1) == 0)
            if three and m & (m - 1) == 0 and m >= 128:
                assert 3 * (m >> 2) < need                     # the smaller 3 * 2^k really was too small
            assert lib.ira_fft_split(m, ctypes.byref(a), ctypes.byref(b)) == 0
            assert a.value * b.value == m and b.value & (b.value - 1) == 0 and b.value <= 8192
            q = a.value // 3 if a.value % 3 == 0 else a.value
            assert q & (q - 1) == 0
    assert conv_size(479501 + 239750) == 3 << 18 and conv_size(479501 + 239750, False) == 1 << 20
    for bad in (0, 15, 48, 5 << 10, 7 << 12, (1 << 22) + 1, 3 << 21, 100000):
        assert lib.ira_fft_split(bad, ctypes.byref(a), ctypes.byref(b)) == -2, bad     # IRA_E_SIZE


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from audio_analysis_amd._lib import IraError
    from audio_analysis_amd.analyse import decay
    with pytest.raises(IraError):
        decay.analyse_decay_for_channel(np.zeros(100, np.float32), SR, "m", decay.DecayAnalysisSettings())


def test_library_reads_no_environment_and_keeps_no_state():
    """the product library is a pure function of its arguments -- no getenv, no function-local
    statics in the sources; tuning knobs only exist under IRA_TUNING_BUILD (ira_common.h)."""
    import re
    for src in sorted((REPO / "audio_analysis_amd" / "csrc").glob("*.hip")) + sorted((REPO / "audio_analysis_amd" / "csrc").glob("*.h")):
        text = src.read_text()
        if src.name == "ira_common.h":
            body = text.split("#ifdef IRA_TUNING_BUILD")[1].split("#else")[0]
            assert "getenv" in body and "getenv" not in text.replace(body, "")
            continue
        assert "getenv" not in text, src.name
        for line in text.splitlines():
            code = line.split("//")[0]
            assert not re.search(r"^\s+static\s+(?!constexpr|inline|_assert)", code), (src.name, line)
    lib_path = REPO / "audio_analysis_amd" / "csrc" / "libira.so"
    if lib_path.exists():
        import subprocess
        syms = subprocess.run(["nm", "-D", "--undefined-only", str(lib_path)], capture_output=True, text=True).stdout
        assert "getenv" not in syms


def test_product_never_imports_the_oracle():
    for py in (REPO / "audio_analysis_amd").rglob("*.py"):
        text = py.read_text()
        assert "import oracle" not in text and "from oracle" not in text, py


# ---------------------------------------------------------------------------------------- host logic
def test_segment_bounds_matches_oracle():
    from audio_analysis_amd.analyse._common import segment_bounds
    rng = np.random.default_rng(5)
    for _ in range(500):
        n = int(rng.integers(1, 100000)); pk = int(rng.integers(0, n))
        trim = bool(rng.integers(0, 2)); ign = float(rng.choice([0.0, 0.001, 0.0105, 0.5, 3.0]))
        dur = rng.choice([None, 0.0, 0.01, 0.25, 5.0])
        dur = None if dur is None else float(dur)
        assert segment_bounds(n, pk, SR, trim, ign, dur) == O.select_segment(n, pk, SR, trim, ign, dur)


def test_io_conversion_and_channel_policy(golden):
    from audio_analysis_amd.analyse import io
    g, _, _ = golden
    for k in ("i16", "i32", "f32"):
        np.testing.assert_array_equal(io.convert_wav_samples_to_float32(g[f"io/{k}"]), g[f"io/{k}_f32"])
    st = np.stack([g["in/xs_l"], g["in/xs_r"]], axis=1)
    la = io.LoadedAudio(samples=st, sample_rate_hz=SR, file_path=Path("x.wav"))
    ch = io.get_analysis_channels(la, True)
    assert ch[0][0] == "mono"
    np.testing.assert_array_equal(ch[0][1], g["io/downmix"])
    assert [n for n, _ in io.get_analysis_channels(la, False)] == ["left", "right"]
    with pytest.raises(ValueError):
        io.validate_audio_format(io.LoadedAudio(st, 44100, Path("x.wav")), 48000, "stereo")
    with pytest.raises(ValueError):
        io.convert_wav_samples_to_float32(np.zeros(4, np.int8))


def test_wav_round_trip(tmp_path, golden):
    from scipy.io import wavfile
    from audio_analysis_amd.analyse import io
    g, _, _ = golden
    pcm = g["report/stereo16/pcm"]
    wavfile.write(str(tmp_path / "s.wav"), SR, pcm)
    la = io.load_wav_file(tmp_path / "s.wav", expected_channel_mode="mono_or_stereo", allow_mono_and_upmix_to_stereo=False)
    np.testing.assert_array_equal(la.samples, O.pcm_to_float32(pcm))
    wavfile.write(str(tmp_path / "m.wav"), SR, pcm[:, 0])
    la = io.load_wav_file(tmp_path / "m.wav")                      # default: stereo expected, mono upmixed
    assert la.samples.shape[1] == 2 and np.array_equal(la.samples[:, 0], la.samples[:, 1])
    with pytest.raises(ValueError):
        io.load_wav_file(tmp_path / "m.wav", expected_channel_mode="stereo", allow_mono_and_upmix_to_stereo=False)


def test_band_tables_and_mask_records(golden):
    from audio_analysis_amd.analyse import rt60bands as rb
    _, c, _ = golden
    for mode in ("three", "octave", "third"):
        got = rb._build_band_definitions(rb.Rt60BandsAnalysisSettings(band_mode=mode), SR)
        assert [[b.name, b.centre_hz, b.kind, b.low_edge_hz, b.high_edge_hz] for b in got] == c[f"xb/rt60bands/{mode}"]["bands"]
    assert len(rb._build_band_definitions(rb.Rt60BandsAnalysisSettings(band_mode="third"), SR)) == 26
    rec = rb.band_mask_record(rb.BandDefinition("Low", 70.0, "lowpass", high_edge_hz=250.0), 1 / 6, 24000.0)
    assert rec[0] == 1.0 and rec[3] == 250.0 and rec[4] == 250.0 * 2.0 ** (1 / 6)
    rec = rb.band_mask_record(rb.BandDefinition("x", 1.0, "bandpass", 3000.0, 2000.0), 1 / 6, 24000.0)
    assert rec[0] == 0.0                                            # inverted edges -> zero mask


def test_slice_selection_and_log_bins(golden):
    from audio_analysis_amd.analyse import modalcloud as mc, waterfall as wf
    g, c, _ = golden
    for T, modes in c["slice_select"].items():
        ft = (np.arange(int(T), dtype=np.float32) * 512.0 / 48000.0).astype(np.float32)
        assert wf._select_slice_frame_indices(ft, wf.WaterfallAnalysisSettings()).tolist() == modes["auto"]
        assert wf._select_slice_frame_indices(
            ft, wf.WaterfallAnalysisSettings(slice_mode="uniform_frames", num_slices=7)).tolist() == modes["uniform_frames"]
    np.testing.assert_array_equal(mc._build_log_bins(20.0, 20000.0, 24, 24), g["modal/edges"])
    freq = np.fft.rfftfreq(8192, 1 / 48000.0).astype(np.float32)
    sel = freq[(freq >= 20.0) & (freq <= 20000.0)]
    cen, first, count = mc.log_bin_rows(sel, g["modal/edges"])
    cen_o, first_o, count_o = O.log_bin_membership(sel, g["modal/edges"])
    np.testing.assert_array_equal(count, count_o)
    np.testing.assert_array_equal(first[count > 0], first_o[count_o > 0])


def test_synth_is_deterministic_and_matches_goldens(golden):
    from audio_analysis_amd.synth import synth_ir
    g, _, _ = golden
    np.testing.assert_array_equal(synth_ir(0, 0, 24000, rt60_seconds=0.15), g["in/xa"])
    np.testing.assert_array_equal(synth_ir(1, 0, 48000, rt60_seconds=0.32, pcm16_round_trip=True), g["in/xb16"])
    x = synth_ir(7, 0, 9600)
    assert x.dtype == np.float32 and abs(np.abs(x).max() - 0.95) < 1e-6 and int(np.argmax(np.abs(x))) == 247


def test_summaries_format_without_gpu(golden):
    """Text formatting of results is pure host code: feed it the golden numbers."""
    from audio_analysis_amd.analyse import decay
    _, c, _ = golden
    case = c["xb/decay"]
    fits = {k: decay.LinearDecayFit(k, (v[0], v[1]), v[2], v[3], v[4], v[5], v[6], v[7]) for k, v in case["fits"].items()}
    r = decay.ChannelDecayAnalysis("mono", SR, case["start"], np.zeros(1, np.float32), np.zeros(1, np.float32),
                                   case["early"], fits)
    assert decay.summarise_decay_results_text([r]) == case["summary"]


# ---------------------------------------------------------------------------------------- CLI surface
IN_SCOPE = ("zplane", "bundle", "decay", "rt60bands", "fr", "filter", "spectrogram", "waterfall", "modalcloud", "report",
            "groupdelay", "diffusion", "deconvolve", "ir")


def test_cli_surface_matches_reference(golden):
    import argparse
    from audio_analysis_amd.analyse import cli
    _, c, _ = golden
    sub = [a for a in cli.build_parser()._actions if isinstance(a, argparse._SubParsersAction)][0]
    for name in IN_SCOPE:
        mine = {}
        for act in sub.choices[name]._actions:
            if act.dest != "help":
                mine[act.dest] = act
        ref_rows = c["cli_surface"][name]
        assert set(mine) == {r["dest"] for r in ref_rows}, name
        for r in ref_rows:
            act = mine[r["dest"]]
            assert list(act.option_strings) == r["flags"], (name, r["dest"])
            assert type(act).__name__ == r["kind"], (name, r["dest"])
            assert act.default == r["default"], (name, r["dest"])
            assert bool(act.required) == r["required"]
            assert (list(act.choices) if act.choices else None) == r["choices"]
            assert getattr(act.type, "__name__", None) == r["type"]
    assert set(sub.choices) == set(c["cli_surface"])                       # all 14 commands, nothing extra


# ---------------------------------------------------------------------------------------- multi-rank
def test_shard_files_partitions_everything():
    from audio_analysis_amd.dist import shard_files
    for f in (0, 1, 7, 8, 9, 65536, 65537):
        for w in (1, 2, 3, 4, 8):
            blocks = [shard_files(f, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == f
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            assert max(hi - lo for lo, hi in blocks) <= -(-f // w) if f else True


def test_balanced_assignment_for_ragged_bundles():
    """ragged bundles are sorted by size and dealt round-robin; gathered records go back to file order."""
    from audio_analysis_amd.dist import balanced_assignment, restore_order
    rng = np.random.default_rng(5)
    for f in (0, 1, 5, 64, 1001):
        sizes = rng.integers(1000, 2_000_000, size=f)
        for w in (1, 2, 3, 8):
            parts = balanced_assignment(sizes, w)
            assert len(parts) == w
            flat = np.concatenate(parts) if f else np.zeros(0, np.int64)
            assert sorted(flat.tolist()) == list(range(f))                       # a partition
            assert all(np.all(np.diff(p) > 0) for p in parts if p.size > 1)       # bundle order inside a rank
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
            if f >= 8 * w:
                loads = np.array([sizes[p].sum() for p in parts], dtype=np.float64)
                contiguous = np.array([sizes[lo:hi].sum() for lo, hi in
                                       [(r * -(-f // w), min(f, (r + 1) * -(-f // w))) for r in range(w)]], dtype=np.float64)
                assert loads.max() / loads.mean() <= max(1.05, contiguous.max() / contiguous.mean())
            # gathered in rank order -> back to file order
            gathered = np.concatenate([np.asarray(p) for p in parts]) if f else np.zeros(0, np.int64)
            assert np.array_equal(gathered[restore_order(parts)], np.arange(f))
            assert all(np.array_equal(a, b) for a, b in zip(parts, balanced_assignment(sizes, w)))   # deterministic


def test_ingest_errors_have_the_reference_reader_types(tmp_path):
    """A missing tap is FileNotFoundError, a non-RIFF file ValueError -- what scipy's reader raises behind the reference's
    load_wav_file (io.py:200); the bundle runner's abort semantics key on them.  A data chunk shorter than its header
    says is NOT an error for the reference (scipy warns and returns what the file holds): the tap is analysed."""
    from audio_analysis_amd import ingest
    with pytest.raises(FileNotFoundError):
        ingest.probe_tap(tmp_path / "nope.wav")
    bad = tmp_path / "bad.wav"
    bad.write_bytes(b"this is not a wave file at all, not even close")
    with pytest.raises(ValueError):
        ingest.probe_tap(bad)
    import struct
    hdr = b"RIFF" + struct.pack("<I", 36 + 4000) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, 48000, 192000, 4, 16)
    short = tmp_path / "short.wav"
    short.write_bytes(hdr + b"data" + struct.pack("<I", 4000) + b"\0" * 100)      # payload shorter than the header says
    info = ingest.probe_tap(short)
    assert info.native and info.frames == 25                       # the 100 bytes the file holds (tests/golden/truncated_wav.json)
    assert ingest.read_tap_pcm16(info).shape == (25, 2)


_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np
from audio_analysis_amd import dist as D
rank, local, world = D.init_process_group("gloo")
full = np.random.default_rng(3).standard_normal((11, 128))
full[2, 5] = np.nan
lo, hi = D.shard_files(11, rank, world)
got = D.gather_metrics(full[lo:hi])
same = D.gather_metrics(full[5 * rank : 5 * rank + 5], equal_rows=True)      # fixed shard sizes: ONE collective, no counts
if rank == 0:
    assert same.tobytes() == full[:10].tobytes(), "equal-rows gather is not byte-identical"
t = D.max_over_ranks(1.0 + rank)
assert D.any_rank_true(rank == 1) is True and D.any_rank_true(False) is False
D.barrier()
if rank == 0:
    assert got.shape == full.shape and got.tobytes() == full.tobytes(), "gather is not byte-identical"
    assert t == float(world)
    print("GATHER_OK")
else:
    assert got is None
"""


def test_gloo_world2_gather_is_byte_identical(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, REPO=str(REPO), MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` outside torchrun (how the driver launches the N = 1 run) starts two rank processes
    itself; rank 0's JSON line reports n_gpus 2 (launch + rendezvous check over gloo, no GPU involved)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(REPO / "bench.py"), "--gpus", "2", "--spawn-probe"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert (line["n_gpus"], line["ranks"], line["rows"]) == (2, [0.0, 1.0], 6)
    # every rank narrows itself to its own CPUs before anything allocates (the cores of its GPU's NUMA
    # node where sysfs shows the topology, an even split of the allowed CPUs here): disjoint sets, whole set covered
    if line["cpus_before"] >= 2:
        assert line["disjoint"] and all(len(a) >= 1 for a in line["affinity"]), line
        assert sum(len(a) for a in line["affinity"]) <= line["cpus_before"]


def test_rank_affinity_follows_the_gpu_numa_nodes():
    """dist.rank_cpu_affinity on a made-up two-socket, eight-GPU host: four ranks per NUMA node, each gets a quarter of its
    node's CPUs, hardware threads of one core stay together when sysfs names the siblings (not here: identity order),
    slices are disjoint and stay inside the rank's node; fewer CPUs than ranks or no topology -> documented fallbacks."""
    from audio_analysis_amd import dist as D
    node0 = list(range(0, 64)) + list(range(128, 192))
    node1 = list(range(64, 128)) + list(range(192, 256))
    gpus = [(f"0000:{b:02x}:00.0", 0 if i < 4 else 1, node0 if i < 4 else node1) for i, b in enumerate(range(5, 85, 10))]
    got = [D.rank_cpu_affinity(r, 8, list(range(256)), gpus) for r in range(8)]
    sets = [set(c) for c, _ in got]
    assert all(len(x) == 32 for x in sets)
    assert all(sets[i].isdisjoint(sets[j]) for i in range(8) for j in range(i))
    assert all(sets[r] <= set(node0 if r < 4 else node1) for r in range(8))
    assert "numa node 1" in got[5][1]
    # the lease allows only some of the host's CPUs: slices come from the allowed ones
    c, how = D.rank_cpu_affinity(1, 2, list(range(0, 16)), gpus)
    assert set(c) <= set(range(16)) and len(c) == 8
    # no topology: even split; one rank: untouched
    c, how = D.rank_cpu_affinity(2, 4, list(range(12)), [])
    assert c == [6, 7, 8] and "even split" in how
    assert D.rank_cpu_affinity(0, 1, [3, 4, 5], [])[0] == [3, 4, 5]
    assert D._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]


def test_visible_devices_reorder_the_gpu_list_before_ranks_index_it():
    """local rank r maps to the r-th GPU HIP will ENUMERATE -- ROCR_VISIBLE_DEVICES filters the PCI-ordered sysfs
    list first, HIP_VISIBLE_DEVICES indexes what is left; an unparsable value (UUIDs) gives no topology (-> even split)."""
    from audio_analysis_amd import dist as D
    node0, node1 = list(range(0, 8)), list(range(8, 16))
    gpus = [(f"0000:{b:02x}:00.0", 0 if i < 2 else 1, node0 if i < 2 else node1) for i, b in enumerate((5, 15, 25, 35))]
    assert D.visible_gpus(gpus, {}) == gpus
    assert [g[0] for g in D.visible_gpus(gpus, {"HIP_VISIBLE_DEVICES": "3,0"})] == ["0000:23:00.0", "0000:05:00.0"]
    assert [g[0] for g in D.visible_gpus(gpus, {"ROCR_VISIBLE_DEVICES": "2,3", "HIP_VISIBLE_DEVICES": "1"})] == ["0000:23:00.0"]
    assert [g[0] for g in D.visible_gpus(gpus, {"HIP_VISIBLE_DEVICES": "1,7,2"})] == ["0000:0f:00.0"]     # stops at the bad index
    assert D.visible_gpus(gpus, {"HIP_VISIBLE_DEVICES": "GPU-abcdef"}) == []
    # rank 0 of two ranks under HIP_VISIBLE_DEVICES=3,0 sits on GPU 3's node (node 1), rank 1 on node 0
    vis = D.visible_gpus(gpus, {"HIP_VISIBLE_DEVICES": "3,0"})
    c0, how0 = D.rank_cpu_affinity(0, 2, list(range(16)), vis)
    c1, _ = D.rank_cpu_affinity(1, 2, list(range(16)), vis)
    assert set(c0) <= set(node1) and set(c1) <= set(node0) and "0000:23:00.0" in how0


def test_keep_packed_without_the_interleave_table_is_refused():
    """ira_rfft_any(keep_packed=1) leaves Z = DFT(x1 + i x2) as the last pass wrote it -- right for interleaved
    elements, silently wrong for paired ones.  Argument validation only (no kernel is launched, no pointer dereferenced)."""
    from audio_analysis_amd import _lib
    lib = _lib.load()
    fake = 64                                            # any non-null address: the call returns before touching memory
    args = [fake, fake, fake, 4, 1, 1 << 19, fake, fake, fake, fake, fake, fake, fake, fake,
            fake, fake, fake, fake, 1000, None, None, None, None]
    assert lib.ira_rfft_any(*args, None, 1, None) == -3                   # x2off given, no interleave table, keep_packed


def test_bench_parity_report_reads_the_record_layout():
    """bench.py's cpu_baseline workers return the oracle's values for the files they time; rank 0 compares them with the
    gathered records (RT60 and pole radii as max and fraction within 1e-4).  Here: records filled FROM the
    oracle's values must compare as exact, a perturbed RT60 must show up, a None / NaN mismatch must be counted."""
    sys.path.insert(0, str(REPO))
    import bench
    from audio_analysis_amd import pipeline as P
    blocks = ("decay", "rt60bands", "fr", "filter", "spectrogram", "waterfall", "modalcloud", "zplane")
    secs, vals, specs = bench._cpu_one((5, 0.5, blocks, "three", False, False, True))
    assert secs > 0 and len(vals) == 1 and len(specs) == 1 and specs[0].shape[0] == 2049
    v = vals[0]
    rec = np.full((1, P.METRICS_WIDTH), np.nan)
    rec[0, P.M_STATUS] = 0.0
    rec[0, P.M_START] = v["start"]; rec[0, P.M_EARLY10] = v["early10"]
    rec[0, P.M_FIT_T20 + 6], rec[0, P.M_FIT_T30 + 6] = v["t20_rt60"], v["t30_rt60"]
    for k, t in enumerate(v["bands_t30"]):
        rec[0, P.M_BANDS + 3 * k] = np.nan if t is None else t
    rec[0, P.M_FR_PEAK], rec[0, P.M_FR_CENTROID] = v["fr_peak_hz"], v["fr_centroid_hz"]
    rec[0, P.M_FILT_PEAK], rec[0, P.M_FILT_1K] = v["filter_peak_hz"], v["filter_1k_db"]
    rec[0, P.M_SPEC_FRAMES], rec[0, P.M_WF_SLICES], rec[0, P.M_WF_BINS] = v["spec_frames"], v["wf_slices"], v["wf_bins"]
    rec[0, P.M_MODAL_POINTS] = v["modal_points"]
    if v["modal_points"]:
        rec[0, P.M_MODAL_MEDIAN], rec[0, P.M_MODAL_P90], rec[0, P.M_MODAL_MAX] = v["modal_median"], v["modal_p90"], v["modal_max"]
    rec[0, P.M_AR_MAX_R], rec[0, P.M_AR_MEDIAN_R], rec[0, P.M_AR_UNSTABLE] = v["ar_max_radius"], v["ar_median_radius"], v["ar_unstable"]
    rep = bench.parity_report(rec, vals, 3)
    for name, e in rep.items():
        if "max_rel" in e:
            assert e["max_rel"] == 0.0 and e["frac_within_1e_4"] == 1.0, (name, e)
        if "exact_matches" in e:
            assert e["exact_matches"] == e["n"], (name, e)
        if "none_pattern_matches" in e:
            assert e["none_pattern_matches"] == e["of"], (name, e)
    assert rep["t30_rt60_s"]["n"] == 1 and rep["ar_max_radius"]["n"] == 1 and rep["band_t30_rt60_s"]["of"] == 3
    rec2 = rec.copy()
    rec2[0, P.M_FIT_T30 + 6] *= 1.0 + 3e-4
    rec2[0, P.M_FIT_T20 + 6] = np.nan
    rec2[0, P.M_START] += 1
    rep2 = bench.parity_report(rec2, vals, 3)
    assert 2.9e-4 < rep2["t30_rt60_s"]["max_rel"] < 3.1e-4 and rep2["t30_rt60_s"]["frac_within_1e_4"] == 0.0
    assert rep2["t20_rt60_s"]["none_pattern_matches"] == 0 and rep2["start_index"]["exact_matches"] == 0
    sp = bench.spectrogram_parity([specs[0] + np.float32(5e-4)], specs, -120.0)
    assert 4e-4 < sp["max_abs_err_db"] < 6e-4 and sp["fraction_within_1e-3_db"] == 1.0


# ---------------------------------------------------------------------------------------- report host helpers
def test_vectorised_row_quantiles_equal_numpys_nan_functions():
    """pipeline._row_median_p90_max replaces numpy.nanmedian / nanpercentile(90) / nanmax along rows (they go row by row
    through apply_along_axis): the same float64 values bit for bit, NaN holes, ties and single-value rows included."""
    from audio_analysis_amd.pipeline import _row_median_p90_max
    rng = np.random.default_rng(0)
    for trial in range(200):
        n, k = int(rng.integers(1, 40)), int(rng.integers(1, 241))
        a = rng.standard_normal((n, k)) * rng.choice([1.0, 1e-3, 1e3])
        if k > 3:
            a[:, 1] = a[:, 0]                                     # ties
        a[rng.random((n, k)) < rng.random()] = np.nan
        for i in range(n):
            if np.all(np.isnan(a[i])):
                a[i, rng.integers(0, k)] = rng.standard_normal()
        med, p90, mx = _row_median_p90_max(a)
        assert np.array_equal(med, np.nanmedian(a, axis=1)), trial
        assert np.array_equal(p90, np.nanpercentile(a, 90, axis=1)), trial
        assert np.array_equal(mx, np.nanmax(a, axis=1)), trial


def test_report_grouping_helpers_and_group_delay_quantiles():
    """Host-side pieces of the batched / text-only report path: per-file grouping of per-channel items, the group-delay
    summary joiner, and the order-statistic ranks that reproduce numpy.median / numpy.percentile."""
    from audio_analysis_amd.analyse import group_delay as gd
    from audio_analysis_amd.analyse import report as rp
    labels = [(0, "left"), (0, "right"), (1, "mono"), (2, "left"), (2, "right")]
    assert rp._group(list("abcde"), labels, 3) == [["a", "b"], ["c"], ["d", "e"]]
    assert rp._texts_per_file(["L0", "R0", "M1", "L2", "R2"], labels, 3) == ["L0\nR0", "M1", "L2\nR2"]
    assert rp._group([], [], 2) == [[], []]
    assert gd.join_group_delay_summary(["- left: x", None, "- right: y"]) == "Group delay summary:\n- left: x\n- right: y"
    assert gd.join_group_delay_summary([None, None]) == "No group delay results."
    rng = np.random.default_rng(3)
    for m in (1, 2, 3, 10, 11, 1000, 1001):
        v = rng.standard_normal(m)
        ranks, gam = gd.quantile_ranks(m)
        s = np.sort(v)
        vals = s[ranks][None, :]
        got = gd.finish_summary_statistics(vals, gam[None, :], np.array([m]))[0]
        assert got[0] == np.median(v) and got[1] == np.percentile(v, 10) and got[2] == np.percentile(v, 90), m
    assert np.all(np.isnan(gd.finish_summary_statistics(np.zeros((1, 6)), np.zeros((1, 3)), np.array([0]))))


def test_bundle_settings_defaults_reproduce_the_reference_behaviour():
    from audio_analysis_amd.analyse.bundle import BundleRunSettings
    s = BundleRunSettings()
    assert s.reports_subdir == "reports" and s.report_settings is None
    assert s.plot_workers == 0 and s.taps_per_batch >= 1           # inline rendering unless asked otherwise


def test_band_pairing_leaves_the_narrowest_band_alone():
    """Engine._pair_bands (host logic of the band filter bank): entries with equal keys ride one inverse transform two by two,
    in order; a group of odd size leaves its NARROWEST band alone (it then is a narrow job of the half-length inverse and
    skips the first pass, ira.h job_info_dev) -- the first of them on a tie -- and every entry appears exactly once."""
    from audio_analysis_amd.engine import Engine
    keys = np.array([0, 0, 0, 1, 1, 2, 2, 2, 2, 2, 3])
    width = np.array([280.0, 1800.0, np.inf, 5.0, 7.0, 9.0, 3.0, 3.0, 8.0, 4.0, 1.0])
    j1, j2 = Engine._pair_bands(keys, width)
    pairs = list(zip(j1.tolist(), j2.tolist()))
    assert pairs == [(1, 2), (0, -1), (3, 4), (5, 7), (8, 9), (6, -1), (10, -1)]
    used = [i for p in pairs for i in p if i >= 0]
    assert sorted(used) == list(range(keys.size))
    # the three bands of the default report: low-pass 250 Hz (280 Hz wide with its ramp), 0.5-2 kHz, high-pass 4 kHz
    from audio_analysis_amd.analyse import rt60bands as rb
    st = rb.Rt60BandsAnalysisSettings()
    recs = np.stack([rb.band_mask_record(b, st.transition_width_octaves, 24000.0) for b in rb._build_band_definitions(st, 48000)])
    kind, w = recs[:, 0], np.full(3, np.inf)
    w[kind == 1.0] = recs[kind == 1.0, 4]
    w[kind == 3.0] = recs[kind == 3.0, 4] - recs[kind == 3.0, 1]
    j1, j2 = Engine._pair_bands(np.zeros(3, dtype=np.int64), w)
    assert list(zip(j1.tolist(), j2.tolist())) == [(1, 2), (0, -1)]
