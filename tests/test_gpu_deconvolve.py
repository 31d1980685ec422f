"""
GPU: sweep deconvolution against the reference's outputs (tests/golden/deconvolve.npz,
made by tests/golden/make_deconvolve_goldens.py) and the oracle.

Tolerance: the result is a float32 rounding of a float64 computation (rFFT, division, irFFT) whose float64 error is
~1e-15 of the response's peak on both sides, so the device and the reference agree to float32 rounding:
|dh| <= 2e-7 * peak(|h|) + 2e-7 * |h| (one float32 ulp is 1.2e-7 relative).
"""
from pathlib import Path

import numpy as np
import pytest
from scipy.io import wavfile

from oracle import ira_oracle as O

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"
SR = 48000


def _close(got, ref, what):
    assert got.shape == ref.shape and got.dtype == np.float32, what
    peak = float(np.max(np.abs(ref)))
    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
    assert np.all(err <= 2e-7 * peak + 2e-7 * np.abs(ref)), (what, float(err.max()), peak)


@pytest.fixture(scope="module")
def z():
    return np.load(GOLD / "deconvolve.npz")


@pytest.mark.parametrize("case,kw", [
    ("default", {}),
    ("raw", dict(normalise_peak=False, remove_dc=False)),
    ("full", dict(output_length_mode="full_fft", regularization_relative=1e-6, target_peak=0.5)),
])
def test_stereo_recording_vs_reference(z, case, kw):
    from audio_analysis_amd.analyse import deconvolve as D
    got = D.deconvolve_impulse_response(z["stereo/recorded_pcm16"], z["stereo/sweep"], SR, D.DeconvolveSettings(**kw))
    _close(got, z[f"stereo/{case}"], case)


def test_mono_float_recording_shorter_than_the_sweep(z):
    from audio_analysis_amd.analyse import deconvolve as D
    got = D.deconvolve_impulse_response(z["mono/recorded_f32"], z["mono/sweep"], SR, D.DeconvolveSettings())
    _close(got, z["mono/default"], "mono")


def test_file_api_and_cli(z, tmp_path, capsys):
    from audio_analysis_amd.analyse import cli, deconvolve as D
    wavfile.write(str(tmp_path / "rec.take1.wav"), SR, z["stereo/recorded_pcm16"])
    wavfile.write(str(tmp_path / "sweep.wav"), SR, z["file/sweep_pcm16"])
    res = D.deconvolve_from_wav_files(tmp_path / "rec.take1.wav", tmp_path / "sweep.wav", None, tmp_path / "o" / "ir.wav")
    _close(res.samples, z["file/ir"], "file")
    rate, written = wavfile.read(str(tmp_path / "o" / "ir.wav"))
    assert rate == SR and written.dtype == np.float32 and np.array_equal(written, res.samples)
    assert D.default_output_ir_path("/x/y/rec.take1.wav").name == str(z["file/default_name"])
    cli.main(["deconvolve", "--recorded_wav_file_path", str(tmp_path / "rec.take1.wav"),
              "--sweep_wav_file_path", str(tmp_path / "sweep.wav")])
    out = capsys.readouterr().out.strip().splitlines()
    assert out == [f"Wrote IR WAV: {tmp_path / 'rec.take1_ir.wav'}", "  sample_rate_hz=48000", "  channels=2",
                   "  length_seconds=0.312"]
    _, written = wavfile.read(str(tmp_path / "rec.take1_ir.wav"))
    _close(written, z["file/ir"], "cli")


def test_errors_like_the_reference(z):
    from audio_analysis_amd.analyse import deconvolve as D
    with pytest.raises(ValueError, match="at least a few samples"):
        D.deconvolve_impulse_response(z["stereo/recorded_pcm16"][:5], z["stereo/sweep"], SR, D.DeconvolveSettings())
    with pytest.raises(ValueError, match="Unknown output_length_mode: nope"):
        D.deconvolve_impulse_response(z["stereo/recorded_pcm16"], z["stereo/sweep"], SR,
                                      D.DeconvolveSettings(output_length_mode="nope"))


def test_batched_device_path_recovers_known_responses_at_full_size():
    """Size-independent property at a realistic size (n_fft = 2^20): recordings made by convolving a 10 s sweep with
    known impulse responses deconvolve back to those responses (band-limited by the sweep); several recordings with
    DIFFERENT sweeps share one device batch and one call; per-file results equal the one-file-at-a-time results to
    float32 rounding (no cross-file coupling), and a spot check against the oracle."""
    from audio_analysis_amd.analyse import deconvolve as D
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    n_sw = 480000
    t = np.arange(n_sw, dtype=np.float64) / SR
    sweeps = []
    for f0, f1 in ((20.0, 20000.0), (30.0, 18000.0)):
        k = np.log(f1 / f0)
        s = 0.5 * np.sin(2 * np.pi * f0 * (n_sw / SR) / k * (np.exp(t / (n_sw / SR) * k) - 1.0))
        s[:480] *= np.linspace(0, 1, 480)
        s[-480:] *= np.linspace(1, 0, 480)
        sweeps.append(s.astype(np.float32))
    irs = [synth_ir(200 + i, 0, 48000, rt60_seconds=0.25, pre_delay=100 + 7 * i) for i in range(3)]
    n_fft = 1 << 20
    recs = []
    for i, h in enumerate(irs):
        sw = sweeps[i % 2].astype(np.float64)
        y = np.fft.irfft(np.fft.rfft(sw, n_fft) * np.fft.rfft(h.astype(np.float64), n_fft), n_fft)[:600000]
        recs.append((y / np.max(np.abs(y)) * 0.9).astype(np.float32))
    s = D.DeconvolveSettings()
    batch, swb = eng.upload(recs), eng.upload(sweeps)
    dev = D.deconvolve_device(eng, batch, [0, 1, 2], swb, [0, 1, 0], SR, s)
    host = dev["h"].cpu().numpy()
    assert list(dev["n_fft"]) == [n_fft] * 3 and list(dev["n_out"]) == [600000] * 3
    for i, h in enumerate(irs):
        got = host[int(dev["off"][i]) : int(dev["off"][i]) + 600000]
        # peak position = the response's pre-delay, and the shape matches the known response in the sweep's band
        assert int(np.argmax(np.abs(got))) == int(np.argmax(np.abs(h)))
        G, R = np.fft.rfft(got[:48000].astype(np.float64)), np.fft.rfft(h.astype(np.float64))
        band = slice(200, 15000)                              # 200 Hz .. 15 kHz at 1 Hz per bin
        gain = np.vdot(R[band], G[band]).real / np.vdot(R[band], R[band]).real      # the peak normalisation's factor
        assert gain > 0 and np.linalg.norm(G[band] - gain * R[band]) < 0.02 * np.linalg.norm(gain * R[band])
        one = D.deconvolve_device(eng, eng.upload([recs[i]]), [0], eng.upload([sweeps[i % 2]]), [0], SR, s)
        # two recordings of a batch ride ONE complex transform (cross-talk ~1e-16 of the larger): float32 roundings of
        # a few samples may differ between batch compositions, nothing more
        _close(one["h"].cpu().numpy()[:600000], got, "alone vs in a batch")
    _close(host[:600000].reshape(-1, 1), O.deconvolve(recs[0], sweeps[0]), "oracle spot check")


def test_input_forms_the_reference_accepts(z):
    """1-D (mono) recordings, int32 PCM and float64 inputs go through convert_wav_samples_to_float32 like the reference's
    entry point; results against the oracle."""
    from audio_analysis_amd.analyse import deconvolve as D
    sweep = z["stereo/sweep"]
    rec16 = z["stereo/recorded_pcm16"][:, 0]
    cases = [rec16, (rec16.astype(np.int32) << 16), rec16.astype(np.float64) / 32768.0]
    for rec in cases:
        got = D.deconvolve_impulse_response(rec, sweep, SR, D.DeconvolveSettings())
        assert got.shape == (15000, 1)
        _close(got, O.deconvolve(rec, sweep), str(rec.dtype))
