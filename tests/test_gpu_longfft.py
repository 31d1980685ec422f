"""GPU parity: arbitrary-length float64 rFFT (Bluestein), fr / filter modules, RT60 band filter bank."""
import numpy as np
import pytest

from oracle import ira_oracle as O

pytestmark = pytest.mark.gpu
SR = 48000


def _rel(a, b):
    return abs(a - b) / max(abs(b), 1e-300)


def _spec(eng, x, L_list, hann):
    b = eng.upload([x])
    offs = np.zeros(len(L_list), dtype=np.int64)
    spec, off = eng.rfft_any(b.x, offs, np.array(L_list, np.int32), hann)
    h = spec.cpu().numpy()
    return [h[2 * o : 2 * (o + L // 2 + 1)].reshape(-1, 2) for o, L in zip(off, L_list)]


def test_rfft_any_lengths():
    """Awkward lengths (prime, 2*prime, power of two, 2^a 3^b 5^c, tiny) against numpy.fft.rfft in float64."""
    from audio_analysis_amd.engine import get_engine
    eng = get_engine()
    rng = np.random.default_rng(11)
    x = (rng.standard_normal(70000) * np.exp(-np.arange(70000) / 9000.0)).astype(np.float32)
    lengths = [32, 33, 97, 1000, 1024, 4099, 23003, 2 * 37 * 311, 65536, 69997, 48000]
    for hann in (False, True):
        got = _spec(eng, x, lengths, hann)
        for L, g in zip(lengths, got):
            seg = x[:L].astype(np.float64)
            if hann:
                seg = seg * np.hanning(L)
            ref = np.fft.rfft(seg)
            err = np.max(np.abs(g[:, 0] + 1j * g[:, 1] - ref)) / np.max(np.abs(ref))
            assert err < 5e-14, (L, hann, err)
            assert g[0, 1] == 0.0


def test_rfft_any_pairs_equal_lengths():
    """Equal-length signals share one complex transform (x1 + i*x2): same spectra as unpaired, odd counts, DC/Nyquist."""
    from audio_analysis_amd.engine import get_engine
    eng = get_engine()
    rng = np.random.default_rng(5)
    n = 30011
    # very different levels on purpose: cross-talk stays at 1e-16 of the LARGER partner
    chans = [(rng.standard_normal(n) * np.exp(-np.arange(n) / 4000.0) * s).astype(np.float32) for s in (1.0, 1e-3, 0.3, 2.0, 0.7)]
    b = eng.upload(chans)
    for L_list in ([30000, 30000, 30000, 30000, 30000], [30011, 30000, 30011, 30000, 30011], [2, 2, 1, 1, 3], [4096] * 5):
        for hann in (False, True):
            lens = np.array(L_list, np.int32)
            assert eng.pair_real_ffts
            spec, off = eng.rfft_any(b.x, b.off, lens, hann)
            h = spec.cpu().numpy()
            peak = 0.0
            refs = []
            for c, L in zip(chans, L_list):
                seg = c[:L].astype(np.float64)
                if hann:
                    seg = seg * np.hanning(L)
                refs.append(np.fft.rfft(seg))
                peak = max(peak, float(np.max(np.abs(refs[-1]))))
            for o, L, ref in zip(off, L_list, refs):
                g = h[2 * o : 2 * (o + L // 2 + 1)].reshape(-1, 2)
                err = np.max(np.abs(g[:, 0] + 1j * g[:, 1] - ref)) / max(peak, 1e-300)
                assert err < 5e-14, (L_list, hann, err)
                assert g[0, 1] == 0.0
                if L % 2 == 0:
                    assert g[-1, 1] == 0.0
    # A/B: unpaired path gives the same numbers to rounding
    try:
        eng.pair_real_ffts = False
        lens = np.array([30000] * 5, np.int32)
        s1, off = eng.rfft_any(b.x, b.off, lens, True)
    finally:
        eng.pair_real_ffts = True
    s2, _ = eng.rfft_any(b.x, b.off, lens, True)
    a1, a2 = s1.cpu().numpy(), s2.cpu().numpy()
    assert np.max(np.abs(a1 - a2)) / np.max(np.abs(a1)) < 1e-14


def test_even_lengths_use_the_half_size_transform_and_match_unpacked():
    """Even, non-smooth L: one complex transform of L/2 on x[2m] + i*x[2m+1]; same spectrum as the full-size path."""
    from audio_analysis_amd.engine import get_engine
    eng = get_engine()
    rng = np.random.default_rng(8)
    n = 70000
    chans = [(rng.standard_normal(n) * np.exp(-np.arange(n) / 9000.0)).astype(np.float32) for _ in range(4)]
    b = eng.upload(chans)
    lens = np.array([2 * 34613, 69994, 2 * 34613, 4 * 17 * 1021], np.int32)       # even, none of them smooth
    assert all(eng.smooth_split(int(v)) is None for v in lens) and eng.half_real_ffts
    for hann in (False, True):
        spec, off = eng.rfft_any(b.x, b.off, lens, hann)
        h = spec.cpu().numpy()
        try:
            eng.half_real_ffts = False
            spec2, _ = eng.rfft_any(b.x, b.off, lens, hann)
        finally:
            eng.half_real_ffts = True
        h2 = spec2.cpu().numpy()
        for c, o, L in zip(chans, off, lens):
            seg = c[:L].astype(np.float64)
            ref = np.fft.rfft(seg * np.hanning(L) if hann else seg)
            g = h[2 * o : 2 * (o + L // 2 + 1)].reshape(-1, 2)
            err = np.max(np.abs(g[:, 0] + 1j * g[:, 1] - ref)) / np.max(np.abs(ref))
            assert err < 5e-14, (L, hann, err)
            assert g[0, 1] == 0.0 and g[-1, 1] == 0.0
            g2 = h2[2 * o : 2 * (o + L // 2 + 1)].reshape(-1, 2)
            assert np.max(np.abs(g - g2)) / np.max(np.abs(ref)) < 1e-13


def test_untangling_fused_into_the_producing_and_consuming_kernels():
    """the separate split passes over a half-length transform are gone.  Smooth lengths with an even
    n1: the second pass of ira_rfft_smooth forms X[k] from Z[k] and Z[n-k], which its mirror-pair tiles bring together
    (240 000 = 480 x 500, 2^18 = 512 x 512, 48 000 = 200 x 240; 50 625 = 225 x 225 keeps round 3's split pass).  Bluestein
    lengths: the spectrum stays packed and ira_spectrum_mag_phase untangles as it reads.  Both against numpy.fft.rfft and
    against the unfused path (Engine.fuse_half_split = False), zero-padded / windowed variants included."""
    from audio_analysis_amd.engine import get_engine
    eng = get_engine()
    rng = np.random.default_rng(23)
    n = 530000
    x = (rng.standard_normal(n) * np.exp(-np.arange(n) / 60000.0)).astype(np.float32)
    b = eng.upload([x, x[::-1].copy()])
    assert eng.fuse_half_split
    smooth = [480000, 1 << 19, 96000, 101250, 2 * 3 * 5 * 128, 128]
    for L in smooth:
        assert eng.smooth_split(L // 2) is not None, L
    try:
        for hann in (False, True):
            for L in smooth:
                lens = np.array([L, L], np.int32)
                ref = [np.fft.rfft(c[:L].astype(np.float64) * (np.hanning(L) if hann else 1.0)) for c in (x, x[::-1])]
                out = {}
                for fused in (True, False):
                    eng.fuse_half_split = fused
                    spec, off = eng.rfft_any(b.x, b.off, lens, hann)
                    h = spec.cpu().numpy().reshape(-1, 2)
                    out[fused] = [h[o : o + L // 2 + 1, 0] + 1j * h[o : o + L // 2 + 1, 1] for o in off]
                for i in range(2):
                    scale = np.max(np.abs(ref[i]))
                    assert np.max(np.abs(out[True][i] - ref[i])) < 5e-14 * scale, (L, hann, i)
                    assert np.max(np.abs(out[True][i] - out[False][i])) < 2e-15 * scale, (L, hann, i)
                    assert out[True][i][0].imag == 0.0 and out[True][i][-1].imag == 0.0
        # zero-padded with its own window length (the group-delay transform): 479 500 samples under hanning(479 500) in 2^19
        eng.fuse_half_split = True
        d, L = 479500, 1 << 19
        spec, off = eng.rfft_any(b.x, b.off[:1], np.array([L], np.int32), True, data_len=np.array([d], np.int32),
                                 win_len=np.array([d], np.int32))
        h = spec.cpu().numpy().reshape(-1, 2)
        ref = np.fft.rfft(x[:d].astype(np.float64) * np.hanning(d), n=L)
        assert np.max(np.abs(h[: L // 2 + 1, 0] + 1j * h[: L // 2 + 1, 1] - ref)) < 5e-14 * np.max(np.abs(ref))
        # Bluestein, even lengths: packed spectra through the dB / phase kernel
        lens = np.array([479254, 95998], np.int32)
        res = {}
        for fused in (True, False):
            eng.fuse_half_split = fused
            spec, off, packed = eng.rfft_any_packed(b.x, b.off, lens, True)
            assert (packed is not None and list(packed) == [1, 1]) if fused else packed is None
            mag, ph = eng.spectrum_mag_phase(spec, off, lens, -120.0, want_phase=True, packed=packed)
            res[fused] = (mag.cpu().numpy(), ph.cpu().numpy(), off)
        for i, L in enumerate(lens):
            o, nb = int(res[True][2][i]), int(L) // 2 + 1
            ref = np.fft.rfft((x if i == 0 else x[::-1])[:L].astype(np.float64) * np.hanning(L))
            ref_db = (20.0 * np.log10(np.maximum(np.abs(ref), 1e-6))).astype(np.float32)
            assert np.max(np.abs(res[True][0][o : o + nb] - ref_db)) < 2e-5
            assert np.max(np.abs(res[True][0][o : o + nb] - res[False][0][o : o + nb])) < 1e-5
            dphi = np.abs(res[True][1][o : o + nb] - np.angle(ref))
            dphi = np.minimum(dphi, 2 * np.pi - dphi)                     # a bin on the negative real axis may land on either side
            strong = np.abs(ref) > 1e-9 * np.max(np.abs(ref))
            assert np.max(dphi[strong]) < 1e-7, np.max(dphi[strong])
    finally:
        eng.fuse_half_split = True


def test_unwrap_equals_numpy_unwrap_on_winding_phases():
    """The unwrap correction no longer calls the library fmod (phases from atan2 keep dd + pi inside [-pi, 3 pi], where the
    floor-mod is a compare and an exact subtraction): against numpy.unwrap(numpy.angle(.)) on spectra whose phase winds
    thousands of times, with exact 0 / pi / 0 steps, bin counts from 2 to 240 k in one batch, float32 degrees and float64
    radians, and with the unwrap switched off."""
    from audio_analysis_amd.engine import get_engine
    eng = get_engine()
    rng = np.random.default_rng(3)
    lengths = np.array([2, 126, 128, 130, 254, 256, 8190, 200001, 479518], np.int32)
    bins = lengths.astype(np.int64) // 2 + 1
    off = (np.cumsum(bins) - bins).astype(np.int64)
    spec = np.empty((int(bins.sum()), 2))
    refs = []
    for L, o, nb in zip(lengths, off, bins):
        k = np.arange(nb)
        ph = -2.0 * np.pi * k * (237.3 / max(int(L), 1)) + 0.3 * rng.standard_normal(nb)
        z = (1.0 + rng.random(nb)) * np.exp(1j * ph)
        if nb > 70:
            z[64] = -abs(z[64]); z[63] = abs(z[63]); z[65] = abs(z[65])          # phases 0, pi, 0
        spec[o : o + nb, 0], spec[o : o + nb, 1] = z.real, z.imag
        refs.append(np.unwrap(np.angle(z)))
    d_spec = eng.to_dev(spec.reshape(-1))
    mag, ph = eng.spectrum_mag_phase(d_spec, off, lengths, -120.0, want_phase=True)
    rad = eng.phase_unwrap(ph, off, lengths, True, False, as_float64=True).cpu().numpy()
    deg = eng.phase_unwrap(ph, off, lengths, True, True).cpu().numpy()
    flat = eng.phase_unwrap(ph, off, lengths, False, False, as_float64=True).cpu().numpy()
    assert np.array_equal(flat, ph.cpu().numpy())
    for o, nb, ref in zip(off, bins, refs):
        assert np.max(np.abs(rad[o : o + nb] - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), nb
        assert np.allclose(deg[o : o + nb], np.rad2deg(ref).astype(np.float32), rtol=3e-7, atol=1e-4)


def test_convolution_sizes_three_times_a_power_of_two():
    """
    Bluestein over M = 3 * 2^k (radix-3 column stage) and the reduced size rule for single real signals (M >= L + L/2: the
    wrap-around never reaches bins k <= L/2): lengths sitting exactly on both sides of the size boundaries, against
    numpy.fft.rfft and against the power-of-two sizes of round 2.
    """
    from audio_analysis_amd.engine import get_engine
    eng = get_engine()
    rng = np.random.default_rng(21)
    n = 100000
    x = (rng.standard_normal(n) * np.exp(-np.arange(n) / 12000.0)).astype(np.float32)
    assert eng.three_pow2_sizes
    # the rule itself
    assert eng.conv_size_for(65535 + 65535 // 2) == 3 << 15          # odd single, tight: 98302 lags in M = 98304
    assert eng.conv_size_for(65537 + 65537 // 2) == 1 << 17          # one sample more: next size
    assert eng.conv_size_for(2 * 49151 - 1) == 3 << 15               # half-length job of L = 98302
    assert eng.conv_size_for(2 * 239750 - 1) == 1 << 19 and eng.conv_size_for(479501 + 479501 // 2) == 3 << 18
    assert eng.conv_size_for(97 + 48) == 192 and eng.conv_size_for(20) == 32
    lengths = [65535, 65537, 98302, 97, 33, 4099, 23003, 69997, 2 * 37 * 311, 99991, 32771, 49153, 2 * 24571]
    assert all(eng.smooth_split(int(v)) is None for v in lengths)
    for hann in (False, True):
        got = _spec(eng, x, lengths, hann)
        try:
            eng.three_pow2_sizes = False
            old = _spec(eng, x, lengths, hann)
        finally:
            eng.three_pow2_sizes = True
        for L, g, g2 in zip(lengths, got, old):
            seg = x[:L].astype(np.float64)
            if hann:
                seg = seg * np.hanning(L)
            ref = np.fft.rfft(seg)
            scale = np.max(np.abs(ref))
            err = np.max(np.abs(g[:, 0] + 1j * g[:, 1] - ref)) / scale
            assert err < 5e-14, (L, hann, err)
            assert g[0, 1] == 0.0 and (L % 2 == 1 or g[-1, 1] == 0.0)
            assert np.max(np.abs(g - g2)) / scale < 1e-13, (L, hann)
    # two signals per transform (all L outputs wanted): the full 2L - 1 rule, still on 3 * 2^k where it fits
    chans = [x[:40000].copy(), x[30000:70000].copy()]
    b = eng.upload(chans)
    try:
        eng.pair_across_channels = True
        lens = np.array([24571, 24571], np.int32)                     # 2 L - 1 = 49141 -> M = 49152 = 3 * 2^14
        spec, off = eng.rfft_any(b.x, b.off, lens, True)
    finally:
        eng.pair_across_channels = False
    h = spec.cpu().numpy()
    for c, o in zip(chans, off):
        ref = np.fft.rfft(c[:24571].astype(np.float64) * np.hanning(24571))
        g = h[2 * o : 2 * (o + 24571 // 2 + 1)].reshape(-1, 2)
        assert np.max(np.abs(g[:, 0] + 1j * g[:, 1] - ref)) / np.max(np.abs(ref)) < 5e-14


def test_chirp_filter_pools_grow_evict_and_fall_back():
    """Engine._filter_pool: pools per (M, stream) under one budget.  With a budget of a few filters the same spectra come
    out whether the filters are cached, rebuilt after an eviction, built into a pool that had to grow, or built into a
    private array because the call does not fit the budget at all."""
    from audio_analysis_amd.engine import get_engine
    eng = get_engine()
    rng = np.random.default_rng(31)
    x = (rng.standard_normal(40000) * np.exp(-np.arange(40000) / 6000.0)).astype(np.float32)
    b = eng.upload([x])
    sets = [[20011, 20013, 20017, 20021], [30011, 30013, 30029], [20011, 20023, 20029, 20047, 20051, 20063],
            [9001, 9007, 9011, 9013, 9029, 9041, 9043, 9049, 9059, 9067, 9091, 9103]]
    saved_budget, saved_pools = eng.filter_cache_bytes, dict(eng._filter_pools)
    try:
        ref = {}
        for lens in sets:
            for L, g in zip(lens, _spec(eng, x, lens, True)):
                ref[L] = g.copy()
                want = np.fft.rfft(x[:L].astype(np.float64) * np.hanning(L))
                assert np.max(np.abs(g[:, 0] + 1j * g[:, 1] - want)) / np.max(np.abs(want)) < 5e-14
        for budget in (6 << 20, 3 << 20, 1 << 20, 0):        # a handful of filters ... nothing: pool, eviction, private array
            eng._filter_pools.clear()
            eng.filter_cache_bytes = budget
            for rep in range(2):
                for lens in sets + sets[::-1]:
                    for L, g in zip(lens, _spec(eng, x, lens, True)):
                        assert np.array_equal(g, ref[L]), (budget, L)
            total = sum(q["cap"] * 16 * k[0] for k, q in eng._filter_pools.items())
            assert total <= max(budget, 0), (budget, total)
    finally:
        eng.filter_cache_bytes = saved_budget
        eng._filter_pools.clear()
        eng._filter_pools.update(saved_pools)


def test_smooth_lengths_take_the_direct_transform_and_match_numpy():
    """n = 2^a 3^b 5^c: two-pass mixed-radix four-step (ira_rfft_smooth) against numpy and against Bluestein."""
    from audio_analysis_amd.engine import get_engine
    eng = get_engine()
    rng = np.random.default_rng(21)
    nmax = 480000
    chans = [(rng.standard_normal(nmax) * np.exp(-np.arange(nmax) / 60000.0) * s).astype(np.float32) for s in (1.0, 0.25, 2.0)]
    b = eng.upload(chans)
    lengths = [480000, 96000, 6000, 1 << 15, 15552, 15625, 1000, 64, 3 * 1024 * 128, 2 * 3 * 5 * 7 * 64]
    for L in lengths:
        smooth = eng.smooth_split(L)
        assert (smooth is None) == (L == 2 * 3 * 5 * 7 * 64), (L, smooth)
        if smooth is not None:
            assert smooth[0] * smooth[1] == L and max(smooth) <= 1024
        for hann in (False, True):
            lens = np.array([L, L, L], np.int32)
            spec, off = eng.rfft_any(b.x, b.off, lens, hann)          # channels 0,1 paired, channel 2 single
            h = spec.cpu().numpy()
            refs = []
            for c in chans:
                seg = c[:L].astype(np.float64)
                refs.append(np.fft.rfft(seg * np.hanning(L) if hann else seg))
            peak = max(float(np.max(np.abs(r))) for r in refs)
            for o, ref in zip(off, refs):
                g = h[2 * o : 2 * (o + L // 2 + 1)].reshape(-1, 2)
                err = np.max(np.abs(g[:, 0] + 1j * g[:, 1] - ref)) / peak
                assert err < 5e-14, (L, hann, err)
                assert g[0, 1] == 0.0 and (L % 2 or g[-1, 1] == 0.0)
    # A/B against Bluestein on the bench length
    lens = np.array([480000] * 3, np.int32)
    s1, _ = eng.rfft_any(b.x, b.off, lens, True)
    try:
        eng.smooth_ffts = False
        s2, _ = eng.rfft_any(b.x, b.off, lens, True)
    finally:
        eng.smooth_ffts = True
    a1, a2 = s1.cpu().numpy(), s2.cpu().numpy()
    assert np.max(np.abs(a1 - a2)) / np.max(np.abs(a2)) < 1e-13


def test_rt60_bands_smooth_vs_bluestein_paths():
    """The band filter bank on a smooth file length (96000) gives the same RT60s through both transform paths."""
    from audio_analysis_amd.analyse import rt60bands as rb
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    chans = [synth_ir(60 + i, 0, 96000, rt60_seconds=0.5 + 0.1 * i) for i in range(3)]
    st = rb.Rt60BandsAnalysisSettings(band_mode="octave", include_t20=True, include_edt=True)
    assert eng.smooth_split(96000) is not None
    direct = rb.analyse_rt60_bands_batch(chans, SR, list("abc"), st)
    try:
        eng.smooth_ffts = False
        blue = rb.analyse_rt60_bands_batch(chans, SR, list("abc"), st)
    finally:
        eng.smooth_ffts = True
    o = O.analyse_rt60_bands(chans[0], SR, band_mode="octave", include_t20=True, include_edt=True)
    for d, bl in zip(direct, blue):
        for name, md in d.band_metrics_by_name.items():
            mb = bl.band_metrics_by_name[name]
            for f in ("rt60_t30_seconds", "rt60_t20_seconds", "edt_seconds"):
                vd, vb = getattr(md, f), getattr(mb, f)
                assert (vd is None) == (vb is None)
                if vd is not None:
                    assert _rel(vd, vb) < 1e-6, (name, f, vd, vb)
    for name, md in direct[0].band_metrics_by_name.items():
        w = o["metrics"][name]["t30"]
        assert (md.rt60_t30_seconds is None) == (w is None)
        if w is not None:
            assert _rel(md.rt60_t30_seconds, w) < 1e-4


def test_band_pairs_across_channels_match_unpaired():
    """rt60bands, three bands x two equal-length channels: the odd bands share one inverse transform."""
    from audio_analysis_amd.analyse import rt60bands as rb
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    chans = [synth_ir(40 + i, 0, 60000, rt60_seconds=0.4 + 0.2 * i) for i in range(3)]
    names = ["a", "b", "c"]
    st = rb.Rt60BandsAnalysisSettings(band_mode="three", include_t20=True, include_edt=True)
    paired = rb.analyse_rt60_bands_batch(chans, SR, names, st)
    try:
        eng.pair_real_ffts = False
        single = rb.analyse_rt60_bands_batch(chans, SR, names, st)
    finally:
        eng.pair_real_ffts = True
    for x, p, s in zip(chans, paired, single):
        o = O.analyse_rt60_bands(x, SR, band_mode="three", include_t20=True, include_edt=True)
        assert list(p.band_metrics_by_name) == list(s.band_metrics_by_name) == [b["name"] for b in o["bands"]]
        for name, mp in p.band_metrics_by_name.items():
            ms, mo = s.band_metrics_by_name[name], o["metrics"][name]
            for field, key in (("rt60_t30_seconds", "t30"), ("rt60_t20_seconds", "t20"), ("edt_seconds", "edt")):
                vp, vs, vo = getattr(mp, field), getattr(ms, field), mo[key]
                assert (vp is None) == (vo is None) == (vs is None), (name, field, vp, vs, vo)
                if vo is not None:
                    assert _rel(vp, vo) < 1e-4 and _rel(vs, vo) < 1e-4, (name, field, vp, vs, vo)
                    assert _rel(vp, vs) < 1e-6


@pytest.mark.parametrize("tag,inp", [("xa", "xa"), ("xc", "xc"), ("xd", "xd"), ("xa_sel", "xa"), ("xa_rect", "xa")])
def test_fr_and_filter_vs_golden(golden, tag, inp):
    from audio_analysis_amd.analyse import filterplot, frequency_response as fr
    g, c, _ = golden
    x = g[f"in/{inp}"]
    cs = c[f"{tag}/fr"]
    r = fr.analyse_frequency_response_for_channel(x, SR, "mono", fr.FrequencyResponseAnalysisSettings(**cs["kw"]))
    assert (r.analysis_start_sample_index, r.analysis_length_samples) == (cs["start"], cs["length"])
    ref = g[f"{tag}/fr/mag_db"]
    assert r.magnitude_db.shape == ref.shape and r.magnitude_db.dtype == np.float32
    np.testing.assert_allclose(r.magnitude_db, ref, rtol=0, atol=2e-5)      # float64 FFT: float32 rounding only
    assert r.peak_frequency_hz == cs["peak"]                                  # argmax bin: exact
    assert _rel(r.spectral_centroid_hz, cs["centroid"]) < 1e-9
    assert fr.summarise_frequency_response_results_text([r]) == cs["summary"]
    np.testing.assert_array_equal(r.frequency_hz, np.fft.rfftfreq(cs["length"], 1 / 48000.0).astype(np.float32))

    cs = c[f"{tag}/filter"]
    r = filterplot.analyse_filter_response_for_channel(x, SR, "mono", filterplot.FilterAnalysisSettings(**cs["kw"]))
    assert (r.analysis_start_sample_index, r.analysis_length_samples) == (cs["start"], cs["length"])
    np.testing.assert_allclose(r.magnitude_db, g[f"{tag}/filter/mag_db"], rtol=0, atol=2e-5)
    assert r.peak_frequency_hz == cs["peak"]
    assert abs(r.magnitude_at_1khz_db - cs["mag1k"]) < 2e-5
    ph_ref = g[f"{tag}/filter/phase"]
    # unwrapped phase in degrees (float32, values up to ~1e6 deg): any flipped 2*pi decision would show as 360
    assert np.max(np.abs(r.phase_response - ph_ref) / np.maximum(1.0, np.abs(ph_ref))) < 1e-6
    assert filterplot.summarise_filter_response_results_text([r]) == cs["summary"]


def test_fr_smoothing_and_radians(golden, monkeypatch):
    from audio_analysis_amd.analyse import filterplot, frequency_response as fr
    g, c, _ = golden
    # the smoothing must be the device kernel (ira_log_smooth_db), not the host restatement kept for curves too long for it
    monkeypatch.setattr(fr, "smooth_log_frequency", lambda *a, **k: pytest.fail("host smoothing ran"))
    r = fr.analyse_frequency_response_for_channel(g["in/xa"], SR, "m",
                                                  fr.FrequencyResponseAnalysisSettings(smoothing_log_bins=9))
    np.testing.assert_allclose(r.magnitude_db, g["xa_smooth/fr/mag_db"], rtol=0, atol=2e-5)
    assert r.peak_frequency_hz == c["xa_smooth/fr"]["peak"]
    assert _rel(r.spectral_centroid_hz, c["xa_smooth/fr"]["centroid"]) < 1e-7
    r = filterplot.analyse_filter_response_for_channel(
        g["in/xa"], SR, "m", filterplot.FilterAnalysisSettings(phase_mode="radians", unwrap_phase=False))
    ref = g["xa_rad/filter/phase"]
    d = np.abs(r.phase_response - ref)
    d = np.minimum(d, np.abs(d - 2 * np.pi))          # +pi / -pi are the same angle
    assert np.max(d) < 1e-5
    with pytest.raises(ValueError):
        fr.analyse_frequency_response_for_channel(g["in/xa"][:260], SR, "m",
                                                  fr.FrequencyResponseAnalysisSettings(trim_to_peak=True))


def test_band_tables_exact(golden):
    from audio_analysis_amd.analyse import rt60bands as rb
    _, c, _ = golden
    for mode in ("three", "octave", "third"):
        got = rb._build_band_definitions(rb.Rt60BandsAnalysisSettings(band_mode=mode), SR)
        want = c[f"xb/rt60bands/{mode}"]["bands"]
        assert [[b.name, b.centre_hz, b.kind, b.low_edge_hz, b.high_edge_hz] for b in got] == want
    with pytest.raises(ValueError):
        rb._build_band_definitions(rb.Rt60BandsAnalysisSettings(band_mode="nope"), SR)


@pytest.mark.parametrize("key", ["xb/rt60bands/three", "xb/rt60bands/octave", "xb/rt60bands/third",
                                 "xd/rt60bands/three", "xd/rt60bands/octave", "xd/rt60bands/third",
                                 "xb16/rt60bands/third", "xc/rt60bands/octave", "xb_ign/rt60bands/three"])
def test_rt60_bands_vs_golden(golden, key):
    from audio_analysis_amd.analyse import decay, rt60bands as rb
    g, c, _ = golden
    case = c[key]
    tag, _, mode = key.split("/")
    kw = dict(case["kw"])
    ign = kw.pop("ignore_leading_seconds", 0.0)
    s = rb.Rt60BandsAnalysisSettings(band_mode=mode, decay_settings=decay.DecayAnalysisSettings(ignore_leading_seconds=ign), **kw)
    r = rb.analyse_rt60_bands_for_channel(g[f"in/{tag.split('_')[0]}"], SR, "mono", s)
    worst = 0.0
    for name, m in r.band_metrics_by_name.items():
        want = case["metrics"][name]
        for got_v, want_v in zip((m.rt60_t30_seconds, m.rt60_t20_seconds, m.edt_seconds), want):
            assert (got_v is None) == (want_v is None), (name, got_v, want_v)
            if want_v is not None:
                worst = max(worst, _rel(got_v, want_v))
    assert worst < 1e-4, worst            # north_star tolerance for RT60 values
    assert rb.summarise_rt60_bands_results_text([r], s.include_t20, s.include_edt) == case["summary"]


def test_rt60_bands_batch_vs_oracle():
    from audio_analysis_amd.analyse import rt60bands as rb
    from audio_analysis_amd.synth import synth_ir
    chans = [synth_ir(i, 0, 60000 + 7 * i) for i in range(3)]
    res = rb.analyse_rt60_bands_batch(chans, SR, ["a", "b", "c"], rb.Rt60BandsAnalysisSettings(band_mode="octave"))
    for x, r in zip(chans, res):
        o = O.analyse_rt60_bands(x, SR, band_mode="octave")
        for name, m in r.band_metrics_by_name.items():
            w = o["metrics"][name]["t30"]
            assert (m.rt60_t30_seconds is None) == (w is None)
            if w is not None:
                assert _rel(m.rt60_t30_seconds, w) < 1e-4, (name, m.rt60_t30_seconds, w)


def test_band_masks_and_band_signals_vs_reference_goldens():
    """a8 and a9 on the DEVICE against the reference's own outputs (tests/golden/band_signals.npz): the mask values the
    inverse transforms multiply with are bit-identical to _make_*_mask on the float32 axis (the kernel's float64 cosine
    rounded once IS the float32 cosine numpy computes, for every bin of a low-pass, a high-pass and two band-pass
    masks on two axes), and irfft(rfft(x) * mask, n) agrees to the float32 rounding of a float64 result -- through
    Bluestein (n = 23257) and through the direct smooth transform (n = 48000)."""
    from pathlib import Path
    import ctypes as C
    from audio_analysis_amd import _lib
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.analyse import rt60bands as rb
    from audio_analysis_amd.analyse.frequency_response import rfft_bin_step
    gdir = Path(__file__).resolve().parent / "golden"
    gb, gi = np.load(gdir / "band_signals.npz"), np.load(gdir / "goldens.npz")
    eng = get_engine()
    t = eng.torch
    e = gb["third1k_edges"]
    defs = {"lp250": rb.BandDefinition("Low", 250.0, "lowpass", None, 250.0),
            "bp500_2000": rb.BandDefinition("Mid", 1000.0, "bandpass", 500.0, 2000.0),
            "hp4000": rb.BandDefinition("High", 4000.0, "highpass", 4000.0, None),
            "third1k": rb.BandDefinition("1000Hz", 1000.0, "bandpass", float(e[0]), float(e[1]))}
    for tag in ("xd", "xb"):
        x = gi[f"in/{tag}"]
        n = int(x.size)
        assert (eng.smooth_split(n) is None) == (tag == "xd")
        fv = rfft_bin_step(n, 48000)
        recs = {k: rb.band_mask_record(b, 1.0 / 6.0, 24000.0) for k, b in defs.items()}
        # ---- a8: masks ------------------------------------------------------------------------------------------------
        for name, rec in recs.items():
            out = eng.empty(n // 2 + 1, t.float32)
            rc = eng.lib.ira_band_mask_values(_lib.dbl_array(list(rec)), float(fv), n // 2 + 1, int(out.data_ptr()), eng.stream)
            assert rc == 0
            got = out.cpu().numpy()
            want = gb[f"mask/{tag}/{name}"]
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (tag, name, np.flatnonzero(got != want)[:5])
            assert 0 < np.count_nonzero((want > 0) & (want < 1)) < want.size // 4        # the transition band is exercised
        # ---- a9: band signals -----------------------------------------------------------------------------------------
        b = eng.upload([x])
        spec, spec_off = eng.rfft_any(b.x, b.off, b.length, use_hann=False)
        names = list(recs)
        y = eng.empty(n * len(names), t.float32)
        eng.band_irfft(spec, np.full(len(names), spec_off[0], np.int64), np.full(len(names), n, np.int32),
                       np.stack([recs[k] for k in names]), np.full(len(names), fv), y, np.arange(len(names), dtype=np.int64) * n)
        got = y.cpu().numpy().reshape(len(names), n)
        for i, name in enumerate(names):
            want = gb[f"y/{tag}/{name}"]
            peak = float(np.abs(want).max())
            err = np.abs(got[i].astype(np.float64) - want.astype(np.float64))
            assert np.all(err <= 2e-7 * peak + 2e-7 * np.abs(want)), (tag, name, float(err.max()), peak)
            assert np.mean(got[i] == want) > 0.9, (tag, name, float(np.mean(got[i] == want)))   # mostly the same float32


def test_narrow_bands_skip_the_first_pass():
    """Round 4: a band whose support spans few bins (third-octave bands below ~800 Hz of a 10 s file; a 250 Hz low-pass) does
    not run the first pass of the direct inverse: ira_band_irfft_smooth compacts its non-zero bins and sums the few terms
    of the pruned column transforms in the second pass.  Same band signals as with both passes (float32 outputs of float64
    sums in a different order) and as numpy's irfft of the masked spectrum; pairs, single bands (half-length inverse),
    a low-pass (bin 0 has no mirror) and wide bands in the same call."""
    import ctypes as C
    from audio_analysis_amd import _lib
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.analyse import rt60bands as rb
    from audio_analysis_amd.analyse.frequency_response import rfft_bin_step
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    t = eng.torch
    st = rb.Rt60BandsAnalysisSettings(band_mode="third")
    third = rb._build_band_definitions(st, SR)
    low = rb.BandDefinition("Low", 70.0, "lowpass", None, 250.0)
    # (4800 = 60 x 80 and 28800 = 160 x 180: plans with six and eight rows per workgroup in the second pass)
    for n, bands in ((96000, third), (480000, third[:9]), (96000, [low]), (96000, [low] + third[4:8]), (4800, third[8:]),
                     (28800, third[3:])):
        assert eng.smooth_split(n) is not None
        x = synth_ir(7, 0, n, rt60_seconds=0.8)
        fv = rfft_bin_step(n, SR)
        recs = np.stack([rb.band_mask_record(b, st.transition_width_octaves, 0.5 * SR) for b in bands])
        b = eng.upload([x])
        spec, spec_off = eng.rfft_any(b.x, b.off, b.length, use_hann=False)
        nb = len(bands)
        args = (spec, np.full(nb, spec_off[0], np.int64), np.full(nb, n, np.int32), recs, np.full(nb, fv))
        yoff = np.arange(nb, dtype=np.int64) * n
        outs = {}
        for sparse in (True, False):
            try:
                eng.sparse_bands = sparse
                y = eng.empty(n * nb, t.float32)
                y.fill_(123.0)
                eng.band_irfft(*args, y, yoff)
                outs[sparse] = y.cpu().numpy().reshape(nb, n)
                if sparse:
                    info = np.concatenate([i.cpu().numpy().reshape(-1, 4) for i in eng.last_band_info])
            finally:
                eng.sparse_bands = True
        assert info[:, 0].any(), (n, info)                              # the path is exercised ...
        if nb > 20:
            assert not info[:, 0].all(), (n, info)                      # ... beside jobs that take both passes
        X = np.fft.rfft(x.astype(np.float64))
        for i in range(nb):
            m = eng.empty(n // 2 + 1, t.float32)
            assert eng.lib.ira_band_mask_values(_lib.dbl_array(list(recs[i])), float(fv), n // 2 + 1, int(m.data_ptr()),
                                                eng.stream) == 0
            want = np.fft.irfft(X * m.cpu().numpy().astype(np.float64), n)
            peak = float(np.abs(want).max())
            for sparse in (True, False):
                err = np.abs(outs[sparse][i].astype(np.float64) - want)
                assert np.all(err <= 2e-7 * peak + 2e-7 * np.abs(want)), (n, bands[i].name, sparse, float(err.max()), peak)
            assert np.mean(outs[True][i] == outs[False][i]) > 0.95, (n, bands[i].name)


def test_band_tile_energies_come_from_the_inverse_and_the_fits_agree():
    """Round 5: the second pass of the smooth band inverses leaves the energies of the 4096-sample EDC tiles of every band
    signal it writes (partials per workgroup, ira.h: ira_band_irfft_smooth tile_part_dev), and ira_edc_fits takes its tile
    totals from them instead of reading every band signal again.  (1) The partials, added up, ARE the tile energies of the
    float32 signals (float64 sums of exact squares: 1e-12 relative) -- pairs, half-length singles, narrow jobs, lengths whose
    last tile is partial; (2) the band RT60s with and without them agree to 1e-9 relative (the carries differ by float64
    rounding of another summation order), None patterns identical; (3) runs are bit-reproducible."""
    from audio_analysis_amd.engine import get_engine
    from audio_analysis_amd.analyse import rt60bands as rb
    from audio_analysis_amd.analyse.frequency_response import rfft_bin_step
    from audio_analysis_amd.synth import synth_ir
    eng = get_engine()
    t = eng.torch
    for n, mode in ((96000, "third"), (48000, "three"), (28800, "octave")):
        st = rb.Rt60BandsAnalysisSettings(band_mode=mode)
        bands = rb._build_band_definitions(st, SR)
        nb = len(bands)
        x = [synth_ir(40 + i, 0, n, rt60_seconds=0.5 + 0.2 * i) for i in range(3)]
        fv = rfft_bin_step(n, SR)
        recs = np.stack([rb.band_mask_record(b, st.transition_width_octaves, 0.5 * SR) for b in bands])
        b = eng.upload(x)
        spec, spec_off = eng.rfft_any(b.x, b.off, b.length, use_hann=False)
        y = eng.empty(3 * nb * n, t.float32)
        yoff = np.arange(3 * nb, dtype=np.int64) * n
        try:
            eng.band_tile_energies = True                     # (off by default: see Engine.band_tile_energies)
            tiles = eng.band_irfft(spec, np.repeat(spec_off, nb), np.full(3 * nb, n, np.int32), np.tile(recs, (3, 1)),
                                   np.full(3 * nb, fv), y, yoff, want_tiles=True)
        finally:
            eng.band_tile_energies = False
        assert tiles is not None
        part, poff, pwgs, ptiles = tiles
        assert np.all(poff >= 0) and np.all(pwgs > 0) and np.all(ptiles == (n + 4095) // 4096)
        ph, yh = part.cpu().numpy(), y.cpu().numpy().astype(np.float64).reshape(3 * nb, n)
        ntile = (n + 4095) // 4096
        for e in range(3 * nb):
            got = ph[poff[e] : poff[e] + ntile * pwgs[e]].reshape(pwgs[e], ntile).sum(axis=0)
            want = np.array([np.sum(yh[e, max(0, n - (j + 1) * 4096) : n - j * 4096] ** 2) for j in range(ntile)])
            assert np.allclose(got, want, rtol=1e-12, atol=1e-300), (n, mode, e, float(np.abs(got - want).max()))
        # the whole block, both ways
        res = {}
        for on in (True, False, True):
            try:
                eng.band_tile_energies = on
                _, vals, have = rb.rt60_bands_device(eng, b, SR, st)
                res.setdefault(on, []).append((vals.copy(), have.copy()))
            finally:
                eng.band_tile_energies = False
        (v1, h1), (v1b, h1b) = res[True]
        (v0, h0), = res[False]
        assert np.array_equal(h1, h0) and np.array_equal(np.isnan(v1), np.isnan(v0))
        assert v1.tobytes() == v1b.tobytes()                                            # run to run: the same bits
        ok = ~np.isnan(v0)
        assert ok.any() and np.all(np.abs(v1[ok] - v0[ok]) <= 1e-9 * np.abs(v0[ok])), (n, mode, float(np.nanmax(np.abs(v1 - v0))))


def test_band_bank_with_edc_smoothing_vs_oracle():
    """The default-off dB smoothing of the EDC (decay.py:161-164) inside the band filter bank (rt60bands.py:356-360 hands the
    decay settings through): smoothed on the device (ira_edc_box_smooth), fitted on the smoothed curve."""
    from audio_analysis_amd.analyse import decay, rt60bands as rb
    from audio_analysis_amd.synth import synth_ir
    x = synth_ir(91, 0, 60000, rt60_seconds=0.3)
    dec = decay.DecayAnalysisSettings(edc_smoothing_window_samples=65)
    r = rb.analyse_rt60_bands_for_channel(x, 48000, "m", rb.Rt60BandsAnalysisSettings(band_mode="octave", include_t20=True,
                                                                                     decay_settings=dec))
    o = O.analyse_rt60_bands(x, 48000, band_mode="octave", include_t20=True, decay=dict(edc_smoothing_window_samples=65))
    assert list(r.band_metrics_by_name) == [b["name"] for b in o["bands"]]
    for name, mm in r.band_metrics_by_name.items():
        for got, key in ((mm.rt60_t30_seconds, "t30"), (mm.rt60_t20_seconds, "t20")):
            want = o["metrics"][name][key]
            assert (got is None) == (want is None), (name, key)
            if want is not None:
                assert abs(got - want) <= 1e-4 * abs(want), (name, key, got, want)
