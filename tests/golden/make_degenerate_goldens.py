#!/usr/bin/env python3
"""
Degenerate-input goldens from the REFERENCE implementation (run in the build container):

    PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 tests/golden/make_degenerate_goldens.py

Inputs are recipes (rebuilt identically by the tests): digital silence, a DC offset, all the energy in the LAST sample,
a NaN sample and an infinite sample inside an otherwise ordinary 1 s IR.  For every per-channel analysis of the hot path the
file stores what the reference returns -- or the exception type and message it raises.  tests/golden/degenerate.json.
"""
from __future__ import annotations

import json
import os
import sys
import warnings
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path[:] = [p for p in sys.path if p.rstrip("/") != "/root/reference"]
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

from audio_analysis_amd.synth import synth_ir  # noqa: E402

import analyse.decay as rdecay  # noqa: E402
import analyse.rt60bands as rbands  # noqa: E402
import analyse.spectrogram as rspec  # noqa: E402
import analyse.modalcloud as rmodal  # noqa: E402
import analyse.frequency_response as rfr  # noqa: E402
import analyse.filterplot as rfilt  # noqa: E402
import analyse.waterfall as rwf  # noqa: E402
import analyse.zplane as rz  # noqa: E402

SR = 48000


def inputs():
    n = 48000
    base = synth_ir(11, 0, n, rt60_seconds=0.2)
    nan = base.copy(); nan[1234] = np.nan
    inf = base.copy(); inf[20000] = np.inf
    last = np.zeros(n, np.float32); last[-1] = 1.0
    return dict(zeros=np.zeros(n, np.float32), dc=np.full(n, 0.25, np.float32), last=last, nan=nan, inf=inf)


def num(v):
    if v is None:
        return None
    v = float(v)
    return v if np.isfinite(v) else repr(v)


def guarded(fn):
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return fn()
    except Exception as exc:  # noqa: BLE001 -- the exception IS the golden
        return dict(raises=type(exc).__name__, message=str(exc))


def main():
    out = {"numpy": np.__version__}
    for tag, x in inputs().items():
        case = {}

        def decay():
            r = rdecay.analyse_decay_for_channel(x, SR, "m", rdecay.DecayAnalysisSettings(compute_edt=True))
            e = r.edc_db
            return dict(start=r.analysis_start_sample_index, early=num(r.early_decay_10db_time_seconds),
                        fits={k: [num(f.rt60_seconds), num(f.slope_db_per_second), num(f.r_squared)] for k, f in r.fits.items()},
                        edc_nan=int(np.isnan(e).sum()), edc_first=num(e[0]), edc_last=num(e[-1]), edc_min=num(np.nanmin(e)) if np.any(~np.isnan(e)) else None,
                        edc_len=int(e.size))
        case["decay"] = guarded(decay)

        def bands():
            r = rbands.analyse_rt60_bands_for_channel(x, SR, "m", rbands.Rt60BandsAnalysisSettings())
            return {k: [num(m.rt60_t30_seconds), num(m.rt60_t20_seconds), num(m.edt_seconds)] for k, m in r.band_metrics_by_name.items()}
        case["bands"] = guarded(bands)

        def fr():
            r = rfr.analyse_frequency_response_for_channel(x, SR, "m", rfr.FrequencyResponseAnalysisSettings())
            return dict(peak=num(r.peak_frequency_hz), centroid=num(r.spectral_centroid_hz), mag_nan=int(np.isnan(r.magnitude_db).sum()),
                        mag_max=num(np.nanmax(r.magnitude_db)) if np.any(~np.isnan(r.magnitude_db)) else None)
        case["fr"] = guarded(fr)

        def filt():
            r = rfilt.analyse_filter_response_for_channel(x, SR, "m", rfilt.FilterAnalysisSettings())
            return dict(mag_1k=num(r.magnitude_at_1khz_db), peak=num(r.peak_frequency_hz), phase_nan=int(np.isnan(r.phase_response).sum()))
        case["filter"] = guarded(filt)

        def spec():
            r = rspec.analyse_spectrogram_for_channel(x, SR, "m", rspec.SpectrogramAnalysisSettings())
            mm = r.magnitude_db
            return dict(shape=list(mm.shape), nan=int(np.isnan(mm).sum()), max=num(np.nanmax(mm)) if np.any(~np.isnan(mm)) else None,
                        min=num(np.nanmin(mm)) if np.any(~np.isnan(mm)) else None)
        case["spectrogram"] = guarded(spec)

        def wf():
            r = rwf.analyse_waterfall_for_channel(x, SR, "m", rwf.WaterfallAnalysisSettings())
            return dict(shape=list(r.slice_magnitude_rel_db.shape), nan=int(np.isnan(r.slice_magnitude_rel_db).sum()))
        case["waterfall"] = guarded(wf)

        def modal():
            r = rmodal.analyse_modal_cloud_for_channel(x, SR, "m", rmodal.ModalCloudAnalysisSettings())
            return dict(points=len(r.points))
        case["modal"] = guarded(modal)

        def zplane():
            seg = x.astype(np.float64)
            pk = int(np.argmax(np.abs(seg)))
            seg = seg[pk:]
            m = float(np.max(np.abs(seg))) if seg.size else 0.0
            if m > 0:
                seg = seg / m
            a = rz._fit_ar_least_squares(seg, 64, 0.0)
            roots = rz._roots_from_poly_descending(a)
            rad = np.abs(roots)
            return dict(npoles=int(roots.size), max_radius=num(rad.max()) if rad.size else None,
                        unstable=int(np.sum(rad >= 1.0)), coeff_nan=int(np.isnan(a).sum()))
        case["zplane"] = guarded(zplane)
        out[tag] = case
    (HERE / "degenerate.json").write_text(json.dumps(out, indent=1, sort_keys=True))
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
