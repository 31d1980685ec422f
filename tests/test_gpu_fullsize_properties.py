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
