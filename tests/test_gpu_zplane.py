"""GPU parity: AR least squares (MFMA Gram + Cholesky), Aberth roots, FIR numerator vs golden vectors."""
from pathlib import Path

import numpy as np
import pytest

from oracle import ira_oracle as O

pytestmark = pytest.mark.gpu
SR = 48000


def _sorted(z):
    z = np.asarray(z)
    return z[np.lexsort((np.round(z.imag, 9), np.round(z.real, 9)))]


def _match_roots(got, ref):
    """Greedy nearest matching (root order is unspecified in numpy.roots); returns max |difference|."""
    assert got.size == ref.size
    ref = list(ref)
    worst = 0.0
    for g in got:
        d = [abs(g - r) for r in ref]
        j = int(np.argmin(d))
        worst = max(worst, d[j])
        ref.pop(j)
    return worst


def test_poly_roots_known():
    from audio_analysis_amd.analyse import zplane as zp
    rts = np.array([0.5, -0.25, 0.9 * np.exp(1j * 0.7), 0.9 * np.exp(-1j * 0.7), 1.1j, -1.1j, 0.99, -0.98])
    c = np.real(np.poly(rts))
    got = zp._roots_from_poly_descending(c)
    assert _match_roots(got, rts) < 1e-12
    # trailing tiny coefficient is dropped (degree falls), trailing exact zeros become roots at 0, leading zeros strip
    got = zp._roots_from_poly_descending(np.array([1.0, -1.5, 0.5, 1e-15]))
    assert _match_roots(got, np.array([1.0, 0.5])) < 1e-12
    got = zp._roots_from_poly_descending(np.array([0.0, 2.0, -3.0, 1.0, 0.0]))
    assert _match_roots(got, O.poly_roots(np.array([0.0, 2.0, -3.0, 1.0, 0.0]))) < 1e-12
    assert zp._roots_from_poly_descending(np.array([3.0])).size == 0


@pytest.mark.parametrize("tag,inp", [("xa_p8", "xa"), ("xa_p64", "xa"), ("xa_p256", "xa"), ("xa_p64_ridge", "xa"),
                                     ("xe_p64", "xe"), ("xb16_p64", "xb16"), ("xc_p32", "xc")])
def test_zplane_vs_golden(golden, tag, inp):
    from audio_analysis_amd.analyse import zplane as zp
    g, c, _ = golden
    cs = c["zplane"][tag]
    s = zp.ZPlaneAnalysisSettings(ar_order=cs["order"], ridge_lambda=cs["ridge"], derive_zeros=True)
    r = zp.analyse_zplane_batch([g[f"in/{inp}"]], SR, ["mono"], s)[0]
    ref_poles = g[f"{tag}/zplane/poles"]
    assert r.poles.size == ref_poles.size
    rad, rad_ref = np.sort(np.abs(r.poles)), np.sort(np.abs(ref_poles))
    assert np.max(np.abs(rad - rad_ref) / rad_ref) < 1e-4            # north_star: pole radii within 1e-4
    assert _match_roots(r.poles, ref_poles) < 1e-6
    assert abs(np.max(rad) - cs["max_r"]) < 1e-8 and abs(np.median(rad) - cs["med_r"]) < 1e-8
    assert int(np.sum(rad >= 1.0)) == cs["unstable"]                  # integer: exact
    assert zp.summarise_zplane_results_text([r]) == cs["summary"]
    ref_zeros = g[f"{tag}/zplane/zeros"]
    assert r.zeros.size == ref_zeros.size
    assert _match_roots(r.zeros, ref_zeros) < 1e-6 * max(1.0, np.max(np.abs(ref_zeros)))


def test_ar_helpers_vs_golden(golden):
    from audio_analysis_amd.analyse import zplane as zp
    g, c, _ = golden
    x = g["in/xa"]
    st = c["zplane"]["xa_p64"]["start"]
    seg = x[st:].astype(np.float64)
    seg = seg / np.max(np.abs(seg))
    a = zp._fit_ar_least_squares(seg, 64)
    ref = g["xa_p64/zplane/a"]
    assert np.max(np.abs(a - ref)) / np.max(np.abs(ref)) < 1e-10
    a = zp._fit_ar_least_squares(seg, 64, 1e-6)
    assert np.max(np.abs(a - g["xa_p64_ridge/zplane/a"])) / np.max(np.abs(ref)) < 1e-10
    b = zp._derive_fir_numerator_from_ar(ref, seg, 64)
    np.testing.assert_allclose(b, g["xa_p64/zplane/b"], rtol=1e-12, atol=1e-14)
    assert zp._fit_ar_least_squares(seg, 0).tolist() == [1.0]
    assert zp._fit_ar_least_squares(seg[:5], 64).size == 5           # order reduced to N-1
    assert zp._rt60_from_pole_radius(1.0, SR) == float("inf")
    assert abs(zp._rt60_from_pole_radius(0.999, SR) - O.rt60_from_radius(0.999, SR)) < 1e-15


def test_zplane_batch_vs_oracle():
    from audio_analysis_amd.analyse import zplane as zp
    from audio_analysis_amd.synth import synth_ir
    chans = [synth_ir(i, 0, 40000 + 333 * i, rt60_seconds=0.2 + 0.1 * i) for i in range(3)]
    chans.append(synth_ir(9, 0, 30000, rt60_seconds=0.3, lowpass_pole=0.9))       # strongly coloured
    res = zp.analyse_zplane_batch(chans, SR, list("abcd"), zp.ZPlaneAnalysisSettings(ar_order=48))
    for x, r in zip(chans, res):
        o = O.analyse_zplane(x, SR, ar_order=48)
        rad, rad_ref = np.sort(np.abs(r.poles)), np.sort(np.abs(o["poles"]))
        assert np.max(np.abs(rad - rad_ref) / rad_ref) < 1e-4
        assert int(np.sum(rad >= 1.0)) == o["unstable"]


def test_ar_lag_path_matches_dense_mfma_gram():
    """The O(pN) lag-sum normal equations (default) against the dense MFMA Gram (flags = IRA_AR_DENSE_GRAM): same coefficients."""
    import os
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    chans = [synth_ir(70 + i, 0, 30000 + 1234 * i, rt60_seconds=0.3 + 0.1 * i, lowpass_pole=0.2 * i) for i in range(4)]
    b = eng.upload(chans)
    for order in (1, 7, 64, 130, 300):
        lens = b.length.astype(np.int32)
        c_lag, info_lag = eng.ar_fit(b.x, b.off, lens, None, order)
        try:
            eng.ar_dense_gram = True                       # IRA_AR_DENSE_GRAM flag of the AR entry points
            c_dense, info_dense = eng.ar_fit(b.x, b.off, lens, None, order)
        finally:
            eng.ar_dense_gram = False
        a, d = c_lag.cpu().numpy().reshape(len(chans), order + 1), c_dense.cpu().numpy().reshape(len(chans), order + 1)
        assert np.all(a[:, 0] == 1.0)
        # both solve the same normal equations; the difference is cond(G) * 1e-16
        scale = np.max(np.abs(d), axis=1, keepdims=True)
        assert np.max(np.abs(a - d) / scale) < 1e-7, (order, np.max(np.abs(a - d) / scale))
        # and against a float64 lstsq on the explicit matrix for the small orders
        if order <= 64:
            for i, x in enumerate(chans):
                s = x.astype(np.float64)
                n = np.arange(order, s.size)
                A = np.stack([s[n - k] for k in range(1, order + 1)], axis=1)
                ref = np.linalg.lstsq(A, -s[n], rcond=None)[0]
                assert np.max(np.abs(a[i, 1:] - ref)) / np.max(np.abs(ref)) < 1e-6, (order, i)


def test_blocked_solver_above_order_128_matches_lstsq():
    """Round 5: above order 128 the Gram matrix lives in global scratch and is factored in PANELS (32 columns up to order 384,
    16 above; ira_ar.hip: chol_factor_blocked / chol_solve_blocked).  Orders on both sides of every boundary (129: a one-column
    last panel; 160, 256; 384 / 385: the panel width changes; 500: ragged last panel of 16-wide panels) against numpy's lstsq
    on the explicit matrix (float64), for well-conditioned responses: coefficients to 1e-7 of the largest one, the condition
    estimate finite and the status 'analysed' (0) or 'refined' (2)."""
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    chans = [synth_ir(90 + i, 0, 9000 + 700 * i, rt60_seconds=0.25 + 0.05 * i) for i in range(3)]
    b = eng.upload(chans)
    lens = b.length.astype(np.int32)
    for order in (129, 160, 256, 384, 385, 500):
        c, info = eng.ar_fit(b.x, b.off, lens, None, order)
        a = c.cpu().numpy().reshape(len(chans), order + 1)
        inf = info.cpu().numpy().reshape(len(chans), -1)
        assert np.all(a[:, 0] == 1.0)
        assert np.all((inf[:, 0] == 0.0) | (inf[:, 0] == 2.0)) and np.all(np.isfinite(inf[:, 3])), (order, inf[:, :4])
        for i, x in enumerate(chans):
            s = x.astype(np.float64)
            n = np.arange(order, s.size)
            A = np.stack([s[n - k] for k in range(1, order + 1)], axis=1)
            ref = np.linalg.lstsq(A, -s[n], rcond=None)[0]
            assert np.max(np.abs(a[i, 1:] - ref)) / np.max(np.abs(ref)) < 1e-7, (order, i, float(np.max(np.abs(a[i, 1:] - ref))))


def _band_limited(seed, b, a, n=48000):
    from scipy.signal import lfilter
    from audio_analysis_amd.synth import synth_ir
    x = lfilter(b, a, synth_ir(seed, 0, n, rt60_seconds=0.4).astype(np.float64))
    return (x / np.max(np.abs(x))).astype(np.float32)


def test_ar_refinement_recovers_lstsq_accuracy_on_ill_conditioned_irs():
    """Recordings band-limited just below Nyquist (anti-alias filters): cond(A) ~ 2e5, i.e. cond(G) ~ 5e10.  The plain
    normal equations are off by ~1e-5 there; with the corrected-semi-normal-equation steps (ira_ar_refine, default) the
    coefficients agree with the reference's SVD-based lstsq to 1e-8 and the pole radii to 1e-8 relative.  Well-conditioned
    channels of the same batch are not touched (info[0] stays 0)."""
    from scipy.signal import butter, ellip
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.analyse import zplane as zp
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    order = 64
    chans = [_band_limited(1, *butter(8, 18000 / 24000)), synth_ir(2, 0, 40000, rt60_seconds=0.3),
             _band_limited(3, *ellip(8, 0.1, 100, 20000 / 24000))]
    ref = [O.fit_ar(x.astype(np.float64), order) for x in chans]
    b = eng.upload(chans)
    lens = b.length.astype(np.int32)

    def fit():
        c, info = eng.ar_fit(b.x, b.off, lens, None, order)
        return c.cpu().numpy().reshape(3, order + 1), info.cpu().numpy().reshape(3, 4)

    def err(c):
        return [float(np.max(np.abs(c[i] - ref[i])) / np.max(np.abs(ref[i]))) for i in range(3)]

    c, info = fit()
    e = err(c)
    assert e[0] < 1e-8 and e[2] < 1e-8 and e[1] < 1e-12, e
    assert list(info[:, 0]) == [2.0, 0.0, 2.0]                    # refined / untouched / refined
    conds = [np.linalg.cond(np.stack([x.astype(np.float64)[np.arange(order, x.size) - k] for k in range(1, order + 1)],
                                     axis=1)) ** 2 for x in chans]
    for est, true in zip(info[:, 3], conds):                      # the estimate brackets cond(G) from above within p
        assert 0.5 * true < est < 2.0 * order * true, (est, true)
    saved = eng.ar_refine_steps
    try:
        eng.ar_refine_steps = 0
        c0, info0 = fit()
    finally:
        eng.ar_refine_steps = saved
    e0 = err(c0)
    assert e0[0] > 1e-7 and e0[2] > 1e-7, e0                      # the test has teeth: unrefined fits are visibly off
    assert np.array_equal(c0[1], c[1]) and list(info0[:, 0]) == [0.0, 0.0, 0.0]
    # end to end through the drop-in API: pole radii against the oracle
    res = zp.analyse_zplane_batch(chans, SR, list("abc"), zp.ZPlaneAnalysisSettings(ar_order=order))
    for x, r in zip(chans, res):
        o = O.analyse_zplane(x, SR, ar_order=order)
        rad, rad_ref = np.sort(np.abs(r.poles)), np.sort(np.abs(o["poles"]))
        assert np.max(np.abs(rad - rad_ref) / rad_ref) < 1e-7
        assert int(np.sum(rad >= 1.0)) == o["unstable"]


def test_rank_deficient_fits_return_the_minimum_norm_solution():
    """a singular Gram matrix used to come back pivot-patched.  For segments whose design matrix is rank
    deficient -- a constant, a few taps followed by digital silence, a short exact AR(2) sequence, silence -- the
    coefficients now equal numpy.linalg.lstsq's MINIMUM-NORM solution (the oracle's fit_ar is the reference's call), the
    solver status says so, and a well-conditioned channel in the same batch is untouched."""
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    n = 6000
    dc = np.full(n, 1.0, np.float32)
    taps = np.zeros(n, np.float32); taps[:4] = [1.0, 0.5, -0.25, 0.125]
    ar2 = np.zeros(n, np.float64); ar2[0] = 1.0; ar2[1] = 1.2
    for i in range(2, n):
        ar2[i] = 1.2 * ar2[i - 1] - 0.72 * ar2[i - 2]
    ar2 = ar2.astype(np.float32)                                        # two poles, radius 0.85: rank 2 (+ rounding dust)
    zeros = np.zeros(n, np.float32)
    good = synth_ir(31, 0, n, rt60_seconds=0.05, pre_delay=0)
    chans = [dc, taps, good, zeros]
    b = eng.upload(chans)
    for order in (8, 64):
        co, info = eng.ar_fit(b.x, b.off, b.length.astype(np.int32), None, order)
        co, info = co.cpu().numpy(), info.cpu().numpy()
        assert list(info[:, 0]) == [4.0, 4.0, 0.0, 4.0], (order, info[:, 0])
        assert info[0, 3] == 1.0 and info[3, 3] == 0.0 and 1 <= info[1, 3] <= 4        # ranks: constant 1, silence 0
        for i, x in enumerate(chans):
            ref = O.fit_ar(x.astype(np.float64), order)
            scale = max(1.0, float(np.abs(ref).max()))
            assert np.abs(co[i] - ref).max() <= 1e-8 * scale, (order, i, np.abs(co[i] - ref).max())
    # the AR(2) sequence at order 8: rank 2 plus float32 rounding dust (singular values 3, 1.9, then six around 1e-8).
    # lstsq keeps every direction down to 1e-10 of sigma_max, so the REFERENCE fits the dust too and returns six more
    # poles, four of them outside the generating pair.  Round 2's eigenvalue cut dropped those directions (and returned
    # only the generating poles on top); the double-double normal equations keep them like lstsq: all eight radii agree.
    b2 = eng.upload([ar2])
    co, info = eng.ar_fit(b2.x, b2.off, b2.length.astype(np.int32), None, 8)
    assert info.cpu().numpy()[0, 0] == 5.0
    roots, cnt = eng.poly_roots(co, 1, 9, 1e-14)
    r = roots.cpu().numpy()[0, : int(cnt.cpu().numpy()[0])]
    rad = np.sort(np.hypot(r[:, 0], r[:, 1]))[::-1]
    rad_ref = np.sort(np.abs(np.roots(O.fit_ar(ar2.astype(np.float64), 8))))[::-1]
    assert rad.size == 8 and np.max(np.abs(rad - rad_ref) / rad_ref) < 1e-6, (rad, rad_ref)
    assert np.sum(np.abs(rad - np.sqrt(0.72)) < 1e-6) == 2                  # the generating pair is among them
    # NaN input: the reference's lstsq raises LinAlgError; the batch API reports status 3, the drop-in function raises
    bad = good.copy(); bad[100] = np.nan
    b3 = eng.upload([bad, good])
    co, info = eng.ar_fit(b3.x, b3.off, b3.length.astype(np.int32), None, 16)
    info = info.cpu().numpy()
    assert info[0, 0] == 3.0 and info[1, 0] in (0.0, 2.0) and np.all(np.isnan(co.cpu().numpy()[0, 1:]))
    from audio_analysis_amd.analyse import zplane
    with pytest.raises(np.linalg.LinAlgError):
        zplane._fit_ar_least_squares(bad.astype(np.float64), 16)


def test_singular_gram_flagged_by_its_condition_estimate_gets_the_minimum_norm_solution():
    """exactly periodic segments make the Gram matrix exactly singular (columns of the Hankel matrix
    repeat), but the float64 Cholesky pivots of such a matrix can stay positive by rounding noise: the element then reaches
    the double-double solver with status 0, on its condition estimate alone.  When the double-double factorisation finds it
    singular it must SAY so (status 1 -> ira_ar_minnorm -> status 4 + rank), not leave a status 0 behind that sends the
    element through refinement of a meaningless float64 factor and reports noise coefficients as a solved fit."""
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    n = 6000
    rng = np.random.default_rng(77)
    chans, periods = [], (3, 4, 8, 12, 20)
    for per in periods:
        chans.append(np.tile(rng.standard_normal(per).astype(np.float32), n // per + 1)[:n].copy())
    good = synth_ir(32, 0, n, rt60_seconds=0.05, pre_delay=0)
    chans.append(good)
    b = eng.upload(chans)
    for order in (32, 64):
        co, info = eng.ar_fit(b.x, b.off, b.length.astype(np.int32), None, order)
        co, info = co.cpu().numpy(), info.cpu().numpy()
        assert list(info[:-1, 0]) == [4.0] * len(periods), (order, info[:, 0])
        assert info[-1, 0] in (0.0, 2.0)
        for i, per in enumerate(periods):
            assert 1 <= info[i, 3] <= per, (order, per, info[i])                # the rank lstsq sees: at most the period
            ref = O.fit_ar(chans[i].astype(np.float64), order)
            scale = max(1.0, float(np.abs(ref).max()))
            assert np.abs(co[i] - ref).max() <= 1e-8 * scale, (order, per, np.abs(co[i] - ref).max())
    # the same through the float64 sample path (deconvolved responses are float64 on the device)
    x64 = np.concatenate([c.astype(np.float64) for c in chans[:3]])
    d64 = eng.to_dev(x64)
    off = np.arange(3, dtype=np.int64) * n
    co, info = eng.ar_fit(d64, off, np.full(3, n, np.int32), None, 64, x_is_f64=True)
    info = info.cpu().numpy()
    assert list(info[:, 0]) == [4.0, 4.0, 4.0], info[:, 0]


def test_rank_cut_of_the_double_double_solver_is_lstsqs():
    """numpy.linalg.lstsq(rcond=None) drops singular values below eps * max(rows, columns) * sigma_max
    (1.3e-12 for 6000 rows).  Three sinusoids plus noise at 3e-13 of their level: six directions carry the signal, the other
    26 sit BETWEEN the double-double solver's old absolute cut (singular-value ratio 1e-13) and lstsq's -- the reference returns
    the rank-6 minimum-norm solution, and so must the device (status 4, rank 6), not a full-rank fit to the noise."""
    from audio_analysis_amd.engine import get_engine
    eng = get_engine()
    n, order = 6000, 32
    t = np.arange(n, dtype=np.float64)
    rng = np.random.default_rng(5)
    x = np.sin(0.3 * t) + 0.7 * np.sin(1.1 * t + 1.0) + 0.5 * np.sin(2.0 * t + 2.0) + 3e-13 * rng.standard_normal(n)
    A = np.stack([x[order - k - 1 : n - k - 1] for k in range(order)], axis=1)
    sv = np.linalg.svd(A, compute_uv=False)
    rcond = np.finfo(np.float64).eps * max(A.shape)
    assert sv[5] > 1e-3 * sv[0] and 1e-13 * sv[0] < sv[6] < rcond * sv[0], (sv[:8] / sv[0], rcond)   # the case is in the gap
    ref = O.fit_ar(x, order)
    co, info = eng.ar_fit(eng.to_dev(x), np.zeros(1, np.int64), np.full(1, n, np.int32), None, order, x_is_f64=True)
    co, info = co.cpu().numpy(), info.cpu().numpy()
    assert info[0, 0] == 4.0 and info[0, 3] == 6.0, info[0]
    assert np.abs(co[0] - ref).max() <= 1e-6 * max(1.0, float(np.abs(ref).max())), np.abs(co[0] - ref).max()


def test_one_wave_solver_gives_the_workgroup_solvers_bits():
    """Round 4: for order <= 64 the normal equations are factored and solved by ONE WAVE per element (left-looking Cholesky in
    the wave's own LDS, the right-hand side one value per lane) instead of a 256-thread workgroup that spends its time at
    ~1000 barriers.  Same operations in the same order: coefficients, status and the info record (pivots, condition estimate)
    are IDENTICAL to the workgroup kernel's (flags = IRA_AR_WORKGROUP_SOLVE), for well-conditioned, refined (cond(G) ~ 1e10),
    double-double and rank-deficient channels in one batch, at orders 1, 8, 33 and 64."""
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    n = 20000
    rng = np.random.default_rng(8)
    chans = [synth_ir(40 + i, 0, n, rt60_seconds=0.1, pre_delay=0) for i in range(3)]
    chans.append(synth_ir(50, 0, n, rt60_seconds=0.2, pre_delay=0, lowpass_pole=0.97))        # refinement range
    chans.append(synth_ir(51, 0, n, rt60_seconds=0.2, pre_delay=0, lowpass_pole=0.995))       # double-double range
    chans.append(np.full(n, 0.25, np.float32))                                                # rank 1
    chans.append(np.tile(rng.standard_normal(6).astype(np.float32), n // 6 + 1)[:n].copy())   # rank <= 6
    b = eng.upload(chans)
    try:
        for order in (1, 8, 33, 64):
            out = {}
            for wg in (False, True):
                eng.ar_workgroup_solve = wg
                co, info = eng.ar_fit(b.x, b.off, b.length.astype(np.int32), None, order)
                out[wg] = (co.cpu().numpy().copy(), info.cpu().numpy().copy())
            assert np.array_equal(out[False][1], out[True][1], equal_nan=True), (order, out[False][1], out[True][1])
            assert np.array_equal(out[False][0], out[True][0], equal_nan=True), order
            st = out[False][1][:, 0]
            assert set(st[:3]) <= {0.0}, (order, st)
            if order >= 8:
                assert st[5] == 4.0 and st[6] == 4.0, (order, st)             # the constant and the period-6 signal: rank deficient
    finally:
        eng.ar_workgroup_solve = False


def _match_poles(got, ref):
    """Greedy nearest-neighbour matching of two pole sets (numpy.roots' order is unspecified): max |got - ref| over pairs."""
    got = list(got)
    worst = 0.0
    for r in ref:
        j = int(np.argmin([abs(g - r) for g in got]))
        worst = max(worst, abs(got[j] - r))
        got.pop(j)
    return worst


@pytest.mark.parametrize("tag", ["lp500_p64", "lp500_p256", "lp2k_p64"])
def test_ill_conditioned_fits_match_the_reference_svd(tag):
    """SURVEY.md section 7, hard part 1: float32 responses low-passed at 500 Hz / 2 kHz, cond(A) ~ 1e8..4e8, cond(A^T A) ~
    1e16..1e17 (tests/golden/ar_illcond.npz: the REFERENCE's _fit_ar_least_squares + poles).  float64 normal equations are
    off by tens of percent there; the double-double path (ira_ar_exact) matches the reference's SVD solve: coefficients 1e-6
    of the largest one, pole radii 1e-6 relative (north star: 1e-4), every pole matched to 1e-6, same unstable count."""
    from audio_analysis_amd.analyse import zplane as zp
    from audio_analysis_amd.engine import get_engine
    g = np.load(Path(__file__).resolve().parent / "golden" / "ar_illcond.npz")
    x = g[f"{tag}/x"]
    order = int(g[f"{tag}/order_rank"][0])
    assert int(g[f"{tag}/order_rank"][1]) == order                    # lstsq kept every singular value: a full-rank fit
    ref_a, ref_p = g[f"{tag}/coeffs"], g[f"{tag}/poles"]
    eng = get_engine()
    b = eng.upload([x])
    co, info = eng.ar_fit(b.x, b.off, b.length.astype(np.int32), None, order)
    a = co.cpu().numpy()[0]
    info = info.cpu().numpy()[0]
    assert info[0] == 5.0, info                                        # solved by the double-double normal equations
    assert np.max(np.abs(a - ref_a)) / np.max(np.abs(ref_a)) < 1e-6
    r = zp.analyse_zplane_batch([x], 48000, ["m"], zp.ZPlaneAnalysisSettings(ar_order=order, trim_to_peak=False,
                                                                             normalise_segment=False))[0]
    assert r.poles.size == ref_p.size
    rg, rr = np.sort(np.abs(r.poles)), np.sort(np.abs(ref_p))
    assert np.max(np.abs(rg - rr) / rr) < 1e-6
    assert _match_poles(r.poles, ref_p) < 1e-6
    assert int(np.sum(rg >= 1.0)) == int(np.sum(rr >= 1.0))
    # without the double-double path the same input is visibly wrong (the test has teeth)
    saved = eng.ar_exact_cond
    try:
        eng.ar_exact_cond = 0.0
        co0, _ = eng.ar_fit(b.x, b.off, b.length.astype(np.int32), None, order)
    finally:
        eng.ar_exact_cond = saved
    a0 = co0.cpu().numpy()[0]
    assert not (np.max(np.abs(a0 - ref_a)) / np.max(np.abs(ref_a)) < 1e-4)


def test_condition_estimate_travels_in_the_metrics_record():
    """pipeline.M_AR_COND: the caller sees how ill-conditioned every channel's pole fit was (and so which solver ran)."""
    from dataclasses import replace
    from audio_analysis_amd import pipeline as P
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    g = np.load(Path(__file__).resolve().parent / "golden" / "ar_illcond.npz")
    eng = get_engine()
    none = replace(P.FullReportSettings(), run_decay=False, run_rt60_bands=False, run_frequency_response=False,
                   run_filter=False, run_spectrogram=False, run_waterfall=False, run_modal_cloud=False)
    m = P.FullReport(eng, none).run(eng.upload([synth_ir(5, 0, 24000, rt60_seconds=0.3), g["lp500_p64/x"]]))
    assert m[0, P.M_AR_COND] < 1e9 < 1e13 < m[1, P.M_AR_COND]
    assert np.all(m[:, P.M_STATUS] == 0)
    z = O.analyse_zplane(g["lp500_p64/x"], ar_order=64)
    assert abs(m[1, P.M_AR_MAX_R] - z["max_radius"]) < 1e-6 * z["max_radius"]
