#!/usr/bin/env python3
"""
bench.py -- IRs/sec of the metrics-only full report (BASELINE.json metric) on MI355X, and the other BASELINE configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config report|2|3|4|5] [--batch B] [--seconds S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

What a step is: one batch of B synthetic 48 kHz IRs per GPU goes from PINNED HOST MEMORY through
the host-to-device copy, the peak pick, every kernel of the configured blocks, the pack of the per-channel metrics
records, their copy back to the host and the gather to rank 0 (RCCL when N > 1).  K_host >= 4 DISTINCT batches rotate,
so no step re-analyses samples that are still in the 256 MB Infinity Cache, and the upload of batch k+1 runs on a copy
stream under the kernels of batch k (audio_analysis_amd.feed).  Weak scaling: B per GPU is fixed.

  --config report (default)  the BASELINE metric: full report (decay + rt60bands[three] + fr + filter + spectrogram +
                             waterfall + modalcloud + zplane AR(64)), B = 256 mono IRs of 10 s per step
  --config 2   256 x 2 s: STFT spectrogram + Schroeder decay
  --config 3   third-octave rt60bands + waterfall on 10 s IRs, B = 256 per step (16 steps = the config's 4096 IRs)
  --config 4   zplane AR(64) + modal cloud on 10 s IRs, B = 256 per step (8 steps = 2048 IRs = one GPU's shard of 16384)
  --config 5   bundle: stereo 5 s PCM16 taps from files through bundle.run_bundle_metrics (native ingest, int16 upload,
               full pipeline), 128 taps per step

Rank 0 prints ONE JSON line:
  "value"           H2D-inclusive throughput, float32 upload (what section 8d defines)
  "value_int16"     the same with 2-byte PCM16 upload + device conversion (what a tap bundle delivers)
  "value_resident"  inputs already in HBM (still rotating over the distinct batches): the compute-only rate
  "value_pull_kernel" the same steps with the batch moved by ira_host_pull instead of hipMemcpyAsync (A/B)
  "roofline"        for the dominant call (largest share of device time) measured live with HIP events recorded on the
                    launch stream -- in a short serialised pass of the same steps (one stream) right after the timed
                    region, because a kernel's own duration is not observable while other streams share the GPU
  "roofline_stft"   for the float32 spectrogram STFT kernel (the north-star HBM gate) + its achieved error distribution
  "cpu_baseline"    the oracle (NumPy restatement of the reference) on this box's host cores, same blocks, bounded sample
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# transfers and format conversion of the ingest: reported in device_ms_per_step_by_call, never the "dominant kernel"
INGEST_CALLS = ("ira_host_pull", "ira_pcm16_to_channels", "ira_pcm16_to_channels_jobs")
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak (FMA counted as two flops)


# ---------------------------------------------------------------------------------------------------------
# CPU baseline (oracle) -- runs in spawned worker processes that never touch the GPU
# ---------------------------------------------------------------------------------------------------------
def _cpu_init():
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[k] = "1"
    import numpy  # noqa: F401
    from oracle import ira_oracle  # noqa: F401
    from audio_analysis_amd import synth  # noqa: F401


def _cpu_one(args):
    """One file through the oracle: (seconds of CPU work, per-channel value dicts, the oracle's spectrogram matrices of the
    probe channels).  The values are what rank 0 compares the GPU's records of the SAME IR indices with (RT60
    and pole radii as max and as fraction-within-tolerance over the batch): the timing sample doubles as the parity sample."""
    index, seconds, blocks, band_mode, stereo, pcm16, want_spec = args
    import numpy as np
    from audio_analysis_amd.synth import synth_ir
    from oracle import ira_oracle as O
    chans = [synth_ir(index, c, int(seconds * 48000)) for c in range(2 if stereo else 1)]
    if pcm16:       # what a tap file holds (recorder.hpp:49-53) and what the reference's loader makes of it (io.py:58-59)
        chans = [O.pcm_to_float32((x * np.float32(32767.0)).astype(np.int16)) for x in chans]
    t0 = time.perf_counter()
    out, specs = [], []
    for x in chans:
        v = {}
        if "decay" in blocks:
            d = O.analyse_decay(x)
            v["start"] = d["start"]
            v["early10"] = d["early_10db"]
            for name in ("T20", "T30"):
                f = d["fits"].get(name)
                v[name.lower() + "_rt60"] = None if f is None else f["rt60"]
        if "rt60bands" in blocks:
            b = O.analyse_rt60_bands(x, band_mode=band_mode)
            v["bands_t30"] = [b["metrics"][bd["name"]]["t30"] for bd in b["bands"]]
        if "fr" in blocks:
            f = O.analyse_frequency_response(x)
            v["fr_peak_hz"], v["fr_centroid_hz"] = f["peak_hz"], f["centroid_hz"]
        if "filter" in blocks:
            f = O.analyse_filter_response(x)
            v["filter_peak_hz"], v["filter_1k_db"] = f["peak_hz"], f["mag_1k_db"]
        if "spectrogram" in blocks:
            sp = O.analyse_spectrogram(x)
            v["spec_frames"] = int(sp["magnitude_db"].shape[1])
            if want_spec:
                specs.append(sp["magnitude_db"])
        if "waterfall" in blocks:
            w = O.analyse_waterfall(x)
            v["wf_slices"], v["wf_bins"] = int(w["slice_rel_db"].shape[0]), int(w["slice_rel_db"].shape[1])
        if "modalcloud" in blocks:
            mc = O.analyse_modal_cloud(x)
            rt = np.array([p[1] for p in mc["points"]], dtype=np.float64)
            v["modal_points"] = int(rt.size)
            if rt.size:
                v["modal_median"], v["modal_p90"], v["modal_max"] = (float(np.median(rt)), float(np.percentile(rt, 90)),
                                                                     float(np.max(rt)))
        if "zplane" in blocks:
            z = O.analyse_zplane(x, ar_order=next((int(b.split("=")[1]) for b in blocks if b.startswith("ar_order=")), 64))
            v["ar_max_radius"], v["ar_median_radius"], v["ar_unstable"] = z["max_radius"], z["median_radius"], z["unstable"]
        if "groupdelay" in blocks:
            g = O.analyse_group_delay(x)["gd"]
            if g.size:
                v["gd_median"], v["gd_p10"], v["gd_p90"] = (float(np.median(g)), float(np.percentile(g, 10)),
                                                            float(np.percentile(g, 90)))
        if "diffusion" in blocks:
            import warnings
            d = O.analyse_diffusion(x, hop_seconds=0.05, max_lag_milliseconds=5.0)      # pipeline.FullReportSettings.diffusion
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", category=RuntimeWarning)
                v["diff_ac_median"], v["diff_ed_median"] = float(np.nanmedian(d["ac"])), float(np.nanmedian(d["ed"]))
        out.append(v)
    return time.perf_counter() - t0, out, specs


def _cpu_noop(_):
    return os.getpid()


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def host_cpu_share(share_cap=None):
    """CPUs this process may actually use: the affinity mask, cut by the cgroup CPU quota when one is set, and by the
    GPU box's per-GPU CPU share (16; IRA_BENCH_CPU_WORKERS overrides) -- a one-GPU lease of a 8-GPU host shows every
    core of the host in its affinity mask but is entitled to its share of them."""
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = os.cpu_count() or 1
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = max(1, int(float(q) / float(per)))
    except Exception:
        pass
    share = int(os.environ.get("IRA_BENCH_CPU_WORKERS", str(share_cap or 16)))
    workers = max(1, min(affinity, quota or affinity, share))
    return {"affinity": affinity, "cgroup_quota": quota, "share_cap": share, "workers": workers}


def cpu_baseline(seconds: float, blocks, band_mode: str, stereo: bool, per_ir_guess_s: float, unit: str,
                 first_index: int = 0, pcm16: bool = False, spec_probe: int = 0, allowed_cpus=None, share_cap=None):
    """One worker process per core this process may run on (the GPU box's CPU share), single-threaded NumPy in each,
    >= 64 units of work (4 per worker); pool start-up and imports are outside the timed wall.  The files are the FIRST ones
    the GPU analysed in the timed region (indices first_index ..), so the oracle's values come back with the timing:
    returns (baseline dict, per-file value dicts, oracle spectrograms of the first `spec_probe` files)."""
    import multiprocessing as mp
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[k] = "1"
    if allowed_cpus is not None:                     # multi-rank runs narrowed rank 0 to its GPU's cores: the other ranks
        try:                                         # are done by now, the baseline may use the host share again
            os.sched_setaffinity(0, allowed_cpus)
        except OSError:
            pass
    avail = host_cpu_share(share_cap)
    workers = avail["workers"]
    # bounded sample: >= 64 files, but no more than ~25 s of wall at the guessed per-file cost
    count = max(64, 4 * workers)
    cap = int(max(workers, 25.0 * workers / max(per_ir_guess_s, 1e-3)))
    count = max(workers, min(count, cap))
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers, initializer=_cpu_init) as pool:
        pool.map(_cpu_noop, range(4 * workers))                       # every worker has started and imported
        t0 = time.perf_counter()
        res = pool.map(_cpu_one, [(first_index + i, seconds, tuple(blocks), band_mode, stereo, pcm16, i < spec_probe)
                                  for i in range(count)], chunksize=1)
        wall = time.perf_counter() - t0
    per = [r[0] for r in res]
    base = {
        "value": count / wall, "unit": unit, "cores": workers, "kind": "port", "cpu_model": cpu_model(),
        "logical_cpus_on_box": os.cpu_count(), "cpus_in_affinity_mask": avail["affinity"],
        "cgroup_cpu_quota": avail["cgroup_quota"], "worker_cap": avail["share_cap"],
        "sample": f"{count} synthetic {seconds:g} s {'stereo files' if stereo else 'mono IRs'} (generator indices "
                  f"{first_index}..{first_index + count - 1}: the first files of the GPU's timed region) over {workers} worker "
                  f"processes (one per core this process may use; single-threaded NumPy oracle, same blocks as the GPU "
                  f"step: {','.join(blocks)}); mean {sum(per)/len(per):.2f} s per file per core, wall {wall:.1f} s "
                  f"(pool start-up excluded)",
    }
    return base, [r[1] for r in res], [m for r in res[:spec_probe] for m in r[2]]


# ---- parity of the gathered records against the oracle's values for the same IRs -----------------------
def parity_report(records, oracle_values, nbands: int):
    """records: (channels, METRICS_WIDTH) rows of the GPU's gathered records, oracle_values: one dict per channel (same
    order).  Relative quantities: n compared, max relative difference, fraction within 1e-4 (north star: RT60 values and
    pole radii within 1e-4 relative); integer / index quantities: exact-match counts; None patterns must agree."""
    import numpy as np
    from audio_analysis_amd import pipeline as P
    rel = {"early_10db_s": ("early10", P.M_EARLY10), "t20_rt60_s": ("t20_rt60", P.M_FIT_T20 + 6),
           "t30_rt60_s": ("t30_rt60", P.M_FIT_T30 + 6), "fr_centroid_hz": ("fr_centroid_hz", P.M_FR_CENTROID),
           "modal_median_rt60_s": ("modal_median", P.M_MODAL_MEDIAN), "modal_p90_rt60_s": ("modal_p90", P.M_MODAL_P90),
           "modal_max_rt60_s": ("modal_max", P.M_MODAL_MAX), "ar_max_radius": ("ar_max_radius", P.M_AR_MAX_R),
           "ar_median_radius": ("ar_median_radius", P.M_AR_MEDIAN_R), "group_delay_median": ("gd_median", P.M_GD_MEDIAN),
           "group_delay_p10": ("gd_p10", P.M_GD_P10), "group_delay_p90": ("gd_p90", P.M_GD_P90),
           "diffusion_autocorr_median": ("diff_ac_median", P.M_DIFF_AC_MEDIAN),
           "diffusion_echo_density_median": ("diff_ed_median", P.M_DIFF_ED_MEDIAN)}
    exact = {"start_index": ("start", P.M_START), "fr_peak_hz": ("fr_peak_hz", P.M_FR_PEAK),
             "filter_peak_hz": ("filter_peak_hz", P.M_FILT_PEAK), "spectrogram_frames": ("spec_frames", P.M_SPEC_FRAMES),
             "waterfall_slices": ("wf_slices", P.M_WF_SLICES), "waterfall_bins": ("wf_bins", P.M_WF_BINS),
             "modal_points": ("modal_points", P.M_MODAL_POINTS), "ar_unstable_poles": ("ar_unstable", P.M_AR_UNSTABLE)}
    out = {}

    def rel_entry(pairs):
        got = np.array([g for g, _ in pairs], dtype=np.float64)
        ref = np.array([np.nan if r is None else r for _, r in pairs], dtype=np.float64)
        none_ok = int(np.sum(np.isnan(got) == np.isnan(ref)))
        both = ~np.isnan(got) & ~np.isnan(ref)
        e = {"n": int(both.sum()), "none_pattern_matches": none_ok, "of": int(got.size)}
        if both.any():
            d = np.abs(got[both] - ref[both]) / np.maximum(np.abs(ref[both]), 1e-300)
            e.update(max_rel=float(d.max()), frac_within_1e_4=float(np.mean(d <= 1e-4)), frac_within_1e_6=float(np.mean(d <= 1e-6)))
        return e

    for name, (key, col) in rel.items():
        pairs = [(records[i, col], v.get(key)) for i, v in enumerate(oracle_values) if key in v]
        if pairs:
            out[name] = rel_entry(pairs)
    bands = [(records[i, P.M_BANDS + 3 * k], v["bands_t30"][k]) for i, v in enumerate(oracle_values) if "bands_t30" in v
             for k in range(min(nbands, len(v["bands_t30"])))]
    if bands:
        out["band_t30_rt60_s"] = rel_entry(bands)
    f1k = [(records[i, P.M_FILT_1K], v["filter_1k_db"]) for i, v in enumerate(oracle_values) if "filter_1k_db" in v]
    if f1k:
        d = np.abs(np.array([g - r for g, r in f1k], dtype=np.float64))
        out["filter_1k_db"] = {"n": len(f1k), "max_abs_db": float(d.max()), "frac_within_1e_3_db": float(np.mean(d <= 1e-3))}
    for name, (key, col) in exact.items():
        pairs = [(records[i, col], v[key]) for i, v in enumerate(oracle_values) if key in v]
        if pairs:
            out[name] = {"n": len(pairs), "exact_matches": int(sum(1 for g, r in pairs if float(g) == float(r)))}
    status = [records[i, P.M_STATUS] for i in range(len(oracle_values))]
    out["status_ok"] = {"n": len(status), "exact_matches": int(sum(1 for v in status if v == 0.0))}
    return out


def spectrogram_parity(gpu_mats, oracle_mats, floor_db: float):
    """float32 STFT dB against the ORACLE's float64 spectrogram of the same channels, on the bins SURVEY.md 8d names
    (reference value > floor + 20 dB): max |delta dB| and the fractions within 1e-3 / 4e-3 dB."""
    import numpy as np
    worst, in1, in4, total = 0.0, 0, 0, 0
    for g, r in zip(gpu_mats, oracle_mats):
        if g.shape != r.shape:
            return {"error": f"shape mismatch {g.shape} vs {r.shape}"}
        mask = r > (floor_db + 20.0)
        err = np.abs(g.astype(np.float64) - r.astype(np.float64))[mask]
        if err.size:
            worst = max(worst, float(err.max()))
            in1 += int((err <= 1e-3).sum()); in4 += int((err <= 4e-3).sum()); total += int(err.size)
    return {"against": "the oracle's spectrogram (NumPy float64 STFT, the reference's algorithm) of the same synthetic IRs, "
                       "computed by the cpu_baseline workers", "channels": len(oracle_mats), "bins": total,
            "bins_rule": "reference value > floor_db + 20 dB (SURVEY.md 8d)", "max_abs_err_db": worst,
            "fraction_within_1e-3_db": (in1 / total) if total else None,
            "fraction_within_4e-3_db": (in4 / total) if total else None,
            "tolerance_met": "max <= 4e-3 dB and >= 99.9 % within 1e-3 dB (tests/test_gpu_decay_stft.py); SURVEY.md 8d asks "
                             "1e-3 dB on every such bin: met on all but the weakest bins of a frame (a float32 transform's "
                             "error is ~2e-7 of the frame's rms)"}


# ---------------------------------------------------------------------------------------------------------
# configs
# ---------------------------------------------------------------------------------------------------------
def config_table():
    from dataclasses import replace
    from audio_analysis_amd.pipeline import FullReportSettings
    full = FullReportSettings()
    none = replace(full, run_decay=False, run_rt60_bands=False, run_frequency_response=False, run_filter=False,
                   run_spectrogram=False, run_waterfall=False, run_modal_cloud=False, run_zplane=False)
    third = replace(full.rt60_bands, band_mode="third")
    return {
        # B = 256 per step: the host enqueues ~150 launches and table uploads per step whatever the batch size; measured on
        # MI355X (H2D included): B = 64 8.4-9.3 k, 128 10.1 k, 256 10.5 k IRs/s
        "report": dict(settings=full, batch=256, seconds=10.0, steps=20, cpu_s=1.3,
                       metric="IRs/sec full report (STFT+RT60bands+zplane), 48 kHz 10 s IR",
                       what="metrics-only full report", excluded=["png rendering", "group delay", "diffusion", "ir plots"]),
        # the reference's literal default `report` (group delay + diffusion on top of the metric's blocks): a profiling target
        # (tools/profile_config.sh literal); the default run reports it as `literal_full_report`
        "literal": dict(settings=replace(full, run_group_delay=True, run_diffusion=True), batch=256, seconds=10.0, steps=10,
                        cpu_s=1.3, metric="IRs/sec literal default report (metric blocks + group delay + diffusion), 48 kHz 10 s IR",
                        what="metrics-only literal default report", excluded=["png rendering", "ir plots"]),
        "2": dict(settings=replace(none, run_decay=True, run_spectrogram=True), batch=256, seconds=2.0, steps=20, cpu_s=0.03,
                  metric="IRs/sec STFT spectrogram + Schroeder decay, 48 kHz 2 s IR (BASELINE config 2)",
                  what="spectrogram STFT 4096/512 + Schroeder decay (EDT/T20/T30)", excluded=["png rendering"]),
        "3": dict(settings=replace(none, run_rt60_bands=True, run_waterfall=True, rt60_bands=third), batch=256,
                  seconds=10.0, steps=16, cpu_s=1.0,
                  metric="IRs/sec third-octave rt60bands + waterfall CSD, 48 kHz 10 s IR (BASELINE config 3)",
                  what="third-octave (26 band) RT60 filter bank + waterfall slices; 16 steps of 256 = the config's 4096 IRs",
                  excluded=["png rendering"]),
        "4": dict(settings=replace(none, run_zplane=True, run_modal_cloud=True), batch=256, seconds=10.0, steps=8, cpu_s=4.5,
                  metric="IRs/sec zplane AR(64) + modalcloud, 48 kHz 10 s IR (BASELINE config 4)",
                  what="zplane AR(order 64) pole fit + modal cloud 8192/512; 8 steps of 256 = one GPU's 2048-IR shard of 16384",
                  excluded=["png rendering"]),
        # 128 taps (256 channels) per step: the host side (probe + read + ~150 launches per step) is what binds this
        # configuration; measured 32 / 64 / 128 taps per step: 3.5 / 4.0 / 4.6 k taps/s
        # (24 steps: one GPU's share of the configuration is 8192 taps = 64 steps; with 8 the first group's unoverlapped read
        # and the last groups' drain were a tenth of the timed region)
        "5": dict(settings=full, batch=128, seconds=5.0, steps=24, cpu_s=1.4,
                  metric="stereo taps/sec bundle report (full pipeline), 48 kHz 5 s stereo PCM16 taps (BASELINE config 5)",
                  what="bundle.run_bundle_metrics over stereo PCM16 tap files (native ingest, int16 upload, device "
                       "conversion, full metrics-only report of both channels)",
                  excluded=["png rendering", "group delay", "diffusion", "ir plots", "markdown"]),
    }


def block_names(settings):
    out = []
    for flag, name in (("run_decay", "decay"), ("run_rt60_bands", "rt60bands"), ("run_frequency_response", "fr"),
                       ("run_filter", "filter"), ("run_spectrogram", "spectrogram"), ("run_waterfall", "waterfall"),
                       ("run_modal_cloud", "modalcloud"), ("run_zplane", "zplane"), ("run_group_delay", "groupdelay"),
                       ("run_diffusion", "diffusion")):
        if getattr(settings, flag):
            out.append(name)
    if settings.run_zplane and settings.zplane.ar_order != 64:
        out.append(f"ar_order={settings.zplane.ar_order}")        # read by the cpu_baseline workers (_cpu_one)
    return out


# ---------------------------------------------------------------------------------------------------------
# roofline models: algorithmic bytes per step of one ABI call (DESIGN.md section 4 states the per-channel figures)
# ---------------------------------------------------------------------------------------------------------
def make_roof(ev, roof_steps, settings, L, n, nchan, traffic_tab):
    import numpy as np
    tot = {k: sum(v) for k, v in ev.items()}
    nb = 0
    if settings.run_rt60_bands:
        from audio_analysis_amd.analyse.rt60bands import _build_band_definitions
        nb = len(_build_band_definitions(settings.rt60_bands, settings.sample_rate_hz))
    nfits_decay = 1 if settings.run_decay else 0

    # group delay (section 8f): the zero-padded transform length of every channel
    gd_bins, gd_nfft = 0.0, None
    if getattr(settings, "run_group_delay", False):
        from audio_analysis_amd.analyse.group_delay import fft_size_for
        gd_nfft = np.array([fft_size_for(int(v), settings.group_delay) for v in L], dtype=np.float64)
        gd_bins = float(np.sum(gd_nfft // 2 + 1))

    def stft_bytes(nfft, hop):
        frames = 1 + (L - nfft) // hop
        return float(np.sum(4.0 * L + 4.0 * (nfft // 2 + 1) * frames))

    def traffic_of(name):
        t = traffic_tab.get(name)
        return None if t is None else t["hbm_bytes_per_channel"] * nchan

    def roof(name):
        """Bytes (or flops) and time are both PER STEP: a call that is launched twice in a step (ira_rfft_any,
        ira_edc_db) is charged the sum of its launches."""
        step_ms = tot[name] / roof_steps
        launches = len(ev[name]) / roof_steps
        b = None
        if name.startswith("ira_stft_mag_db") and "[f32" in name:
            b = stft_bytes(settings.spectrogram.n_fft, settings.spectrogram.hop_length)
            what = "4L in + 4*F*T out bytes per channel"
        elif name.startswith("ira_stft_mag_db") and ",sel]" in name:
            nf, S = settings.waterfall.n_fft, max(2, settings.waterfall.num_slices)
            b = float(nchan) * S * (8.0 * nf + 4.0 * (nf // 2 + 1))
            what = f"per channel: {S} selected frames x (n_fft samples in (f64 window product) + F floats out)"
        elif name.startswith("ira_stft_mag_db") and ("[f64,n%d]" % settings.modal_cloud.n_fft) in name:
            b = stft_bytes(settings.modal_cloud.n_fft, settings.modal_cloud.hop_length)
            what = "4L in + 4*F*T out bytes per channel"
        elif name.startswith("ira_stft_logbin"):
            nf = settings.modal_cloud.n_fft
            frames = 1 + (L - nf) // settings.modal_cloud.hop_length
            b = float(np.sum(4.0 * L + 4.0 * 240 * frames))
            what = "4L in + 4*nbins*T out bytes per channel (fused STFT + log-bin aggregation; the dB matrix is never written)"
        elif name.startswith("ira_rfft_any"):
            b = float(np.sum(4.0 * L + 16.0 * (L // 2 + 1)))
            what = "per channel: 4L + 16(L/2+1) bytes (fr/filter spectrum of arbitrary length)"
        elif name.startswith("ira_rfft_smooth") and "[gd]" in name:
            b = float(np.sum(4.0 * np.minimum(L, gd_nfft))) + 16.0 * gd_bins
            what = "per channel: 4 min(L, n_fft) sample bytes + 16(n_fft/2+1) (zero-padded group-delay transform, n_fft = 2^19)"
        elif name.startswith("ira_rfft_smooth"):
            b = float(nchan) * (4.0 * n + 16.0 * (n // 2 + 1))
            what = "per channel: 4n + 16(n/2+1) bytes (RT60 full-file forward transform, direct mixed radix, half-length complex)"
        elif name.startswith("ira_band_irfft") and nb:
            b = float(nchan) * (16.0 * (n // 2 + 1) + nb * 4.0 * n)
            what = f"per channel: 16(n/2+1) spectrum in + {nb} band signals x 4n out bytes"
        elif name.startswith("ira_edc_db"):
            b = float(np.sum(8.0 * L)) * (nfits_decay + nb)
            what = f"4L in + 4L out bytes per segment; {nfits_decay} decay + {nb} band segments per channel"
        elif name.startswith("ira_curve_fits"):
            b = float(np.sum(4.0 * L)) * (nfits_decay + nb)
            what = f"4L bytes per EDC curve read once; {nfits_decay} decay + {nb} band curves per channel (+ modal curves, small)"
        elif name.startswith("ira_ar_gram"):
            b = float(np.sum(4.0 * L))
            what = "4L bytes per channel (samples read once; p+1 lag sums)"
        elif name.startswith("ira_peak_index"):
            b = float(nchan) * 4.0 * n
            what = "4N bytes per channel"
        elif name.startswith("ira_edc_fits"):
            # the decay segment is the trimmed channel (L samples), a band segment the band signal from the start index on
            b = float(np.sum(4.0 * L)) * (nfits_decay + nb)
            what = (f"4L bytes per segment (samples read once; the EDC is never written): {nfits_decay} decay + {nb} band "
                    f"segments per channel")
        elif name.startswith("ira_spectrum_mag_phase"):
            bins = gd_bins if "[gd]" in name else np.sum(L // 2 + 1)
            b = float(bins) * (16.0 + 4.0 + 8.0)
            what = "per bin: 16 B spectrum in + 4 B dB + 8 B phase out"
        elif name.startswith("ira_phase_unwrap"):
            bins = gd_bins if "[gd]" in name else np.sum(L // 2 + 1)
            b = float(bins) * (8.0 + (8.0 if "[gd]" in name else 4.0))
            what = "per bin: 8 B wrapped phase in + unwrapped phase out (float32 degrees; float64 radians for the group delay)"
        elif name.startswith("ira_spectrum_stats"):
            b = float(np.sum(L // 2 + 1)) * 4.0
            what = "4 B dB per bin read once"
        elif name.startswith("ira_group_delay"):
            b = float(gd_bins) * 16.0
            what = "per bin: 8 B unwrapped phase in + 8 B group delay out"
        elif name.startswith("ira_diffusion"):
            b = float(nchan) * 4.0 * n
            what = "4N bytes per channel (every window's samples read once; windows overlap through LDS)"
        if b is None:
            return {"kernel": name, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None, "traffic": None, "avg_launch_ms": step_ms / launches, "ms_per_step": step_ms,
                    "algorithmic": "not modelled (latency-bound small-grid call)"}
        ach = b / (step_ms * 1e-3) / 1e9
        tr = traffic_of(name)
        out = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": ach / HBM_PEAK_GBS, "traffic": tr, "algorithmic_bytes": b, "launches_per_step": launches,
               "avg_launch_ms": step_ms / launches, "ms_per_step": step_ms, "algorithmic": what,
               "traffic_GBps": None if tr is None else tr / (step_ms * 1e-3) / 1e9}
        # Whole-file float64 transforms do not fit on a CU: SURVEY.md 8(d) asks for the two-pass STREAMING model beside the
        # compulsory bytes (every pass reads and writes its n complex values once), and they are vector-float64 work.
        stream_b = flops = None
        if name.startswith("ira_band_irfft_smooth") and nb:
            # a channel's transforms are its own (round 3): two of its bands ride one full-length complex inverse, a band
            # left over takes a half-length one
            full, half, h = nchan * (nb // 2), nchan * (nb % 2), n // 2
            stream_b = full * (16.0 * (n // 2 + 1) + 16.0 * n + 16.0 * n + 2 * 4.0 * n) + \
                       half * (16.0 * (n // 2 + 1) + 16.0 * h + 16.0 * h + 4.0 * n)
            flops = full * 5.0 * n * np.log2(n) + half * 5.0 * h * np.log2(h)
        elif name.startswith("ira_rfft_smooth") and "[gd]" in name:
            h = gd_nfft / 2.0
            stream_b = float(np.sum(4.0 * np.minimum(L, gd_nfft) + 16.0 * h * 3 + 16.0 * (h + 1)))
            flops = float(np.sum(5.0 * h * np.log2(h)))
        elif name.startswith("ira_rfft_smooth"):
            # one real signal of even length = ONE half-length complex transform (x[2m] + i x[2m+1]); the untangling is
            # part of the second pass since round 4 (no separate split pass)
            h = n // 2
            stream_b = nchan * (4.0 * n + 16.0 * h + 16.0 * h + 16.0 * (h + 1))
            flops = nchan * 5.0 * h * np.log2(h)
        elif name.startswith("ira_rfft_any"):
            # Bluestein: an even length rides a half-length transform (2 l - 1 lags), an odd one a full-length transform of a
            # single real signal (l + l/2 lags); M = the smallest of 2^k, 3 * 2^k (engine.conv_size)
            from audio_analysis_amd.engine import conv_size
            tlen = np.where(L % 2 == 0, L // 2, L)
            need = np.where(L % 2 == 0, 2 * tlen - 1, tlen + tlen // 2)
            M = np.array([conv_size(int(v)) for v in need], dtype=np.float64)
            stream_b = float(np.sum(4.0 * L + 16.0 * M * 5 + 16.0 * (L // 2 + 1)))
            flops = float(np.sum(2 * 5.0 * M * np.log2(M) + 6.0 * M))
        if name.startswith("ira_stft_logbin") or (name.startswith("ira_stft_mag_db") and "[f64" in name and ",sel]" not in name):
            # 2.5 n_fft log2(n_fft) flop per frame (a real transform of n_fft points), float64 vector work
            nf = settings.modal_cloud.n_fft
            frames = 1 + (L - nf) // settings.modal_cloud.hop_length
            flops = float(np.sum(2.5 * nf * np.log2(nf) * frames))
            out["flops_model"] = {"flop": flops, "achieved_TFLOPs": flops / (step_ms * 1e-3) / 1e12,
                                  "f64_vector_peak_TFLOPs": F64_VECTOR_PEAK_TFLOPS,
                                  "frac_of_f64_vector_peak": flops / (step_ms * 1e-3) / 1e12 / F64_VECTOR_PEAK_TFLOPS,
                                  "model": "2.5 n_fft log2(n_fft) flop per frame (SURVEY.md 8d); the kernel computes in float64"}
        if name.startswith("ira_diffusion"):
            from audio_analysis_amd.analyse.diffusion import window_geometry, _frame_count, trim_start
            win, hop, max_lag = window_geometry(settings.sample_rate_hz, settings.diffusion)
            frames = np.array([_frame_count(int(v), win, hop) for v in L], dtype=np.float64)
            mac = float(np.sum(frames)) * win * max_lag
            out["flops_model"] = {"flop": 2.0 * mac, "achieved_TFLOPs": 2.0 * mac / (step_ms * 1e-3) / 1e12,
                                  "f64_vector_peak_TFLOPs": F64_VECTOR_PEAK_TFLOPS,
                                  "frac_of_f64_vector_peak": 2.0 * mac / (step_ms * 1e-3) / 1e12 / F64_VECTOR_PEAK_TFLOPS,
                                  "model": f"windows x {win} samples x {max_lag} lags multiply-adds in float64 (the reference's "
                                           f"windowed autocorrelation over all lags, diffusion.py:139-226)"}
        if stream_b is not None:
            out["streaming_model"] = {"bytes": stream_b, "achieved_GBps": stream_b / (step_ms * 1e-3) / 1e9,
                                      "frac_of_hbm_peak": stream_b / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      "model": "every pass over the job's complex float64 work array reads and writes it once "
                                               "(two passes for the direct smooth transform, three plus the chirp-filter "
                                               "spectrum for Bluestein), plus inputs and outputs"}
            out["flops_model"] = {"flop": flops, "achieved_TFLOPs": flops / (step_ms * 1e-3) / 1e12,
                                  "f64_vector_peak_TFLOPs": F64_VECTOR_PEAK_TFLOPS,
                                  "frac_of_f64_vector_peak": flops / (step_ms * 1e-3) / 1e12 / F64_VECTOR_PEAK_TFLOPS,
                                  "model": "5 n log2 n per complex transform of n points (Bluestein: two of M points + 6 M)"}
        return out

    return tot, roof


def load_traffic(cfg: str):
    """Per-call HBM traffic from the PMC passes committed under profiles/ (tools/profile_config.sh + traffic_profile.py;
    MI355X_MICROARCH.md's 2 x FETCH_SIZE + WRITE_SIZE rule): the newest round's table for this configuration."""
    for rnd in ("r05", "r04", "r03", "r02"):
        try:
            return json.load(open(os.path.join(REPO, "profiles", f"{rnd}_traffic_{cfg}.json")))["calls"]
        except Exception:
            continue
    return {}


# ABI calls that are ONE kernel launch: their live per-call time is that kernel's own duration
SINGLE_KERNEL_CALLS = {"ira_stft_logbin": "stft5_kernel", "ira_stft_mag_db_tf": "stft6_kernel", "ira_ar_gram": "ar_lag_kernel",
                       "ira_spectrum_mag_phase": "mag_phase_kernel", "ira_phase_unwrap": "unwrap_kernel",
                       "ira_spectrum_stats": "stats_kernel", "ira_poly_roots": "poly_roots_kernel"}


def dominant_kernel_of_profile(cfg: str):
    """The kernel with the largest total duration in the committed rocprofv3 --kernel-trace --stats summary of this
    configuration (profiles/rNN_kernel_stats_cfg<cfg>.csv): (short name, share of device time, file, template arguments,
    average duration in ms, calls) or None."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r0[3-9]*_kernel_stats_cfg{cfg}.csv")))
    if not files:
        files = sorted(glob.glob(os.path.join(REPO, "profiles", f"r02_v2_kernel_stats_cfg{cfg}.csv")))
    if not files:
        return None
    best, total = None, 0.0
    for r in csv.DictReader(open(files[-1])):
        t = float(r["TotalDurationNs"])
        total += t
        if best is None or t > best[1]:
            best = (r["Name"], t, float(r["AverageNs"]) / 1e6, int(r["Calls"]))
    if best is None or total <= 0:
        return None
    full = best[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    name = full.split("<")[0]
    targs = full[len(name):]
    return name, best[1] / total, os.path.basename(files[-1]), targs, best[2], best[3]


# kernels of multi-launch ABI calls: (the call they belong to, bytes one launch streams as a function of the job geometry)
MULTI_KERNEL_CALLS = {"smooth_cols_kernel<1": "ira_band_irfft_smooth", "smooth_rows_kernel<1": "ira_band_irfft_smooth",
                      "smooth_rows_sparse_kernel": "ira_band_irfft_smooth", "band_compact_kernel": "ira_band_irfft_smooth",
                      "smooth_cols_kernel<0": "ira_rfft_smooth", "smooth_rows_kernel<0": "ira_rfft_smooth",
                      "cols_fwd_kernel<0": "ira_rfft_any", "rows_kernel<1": "ira_rfft_any", "cols_inv_kernel<0": "ira_rfft_any",
                      "edc_moments_kernel": "ira_edc_fits", "edc_fit_kernel": "ira_edc_fits", "edc_sums_kernel": "ira_edc_fits"}


def stft_error_distribution(eng, batch, settings, channels: int = 4):
    """Achieved float32-STFT error against the float64 STFT of the same kernels family on the same samples (which the
    parity suite pins to the oracle at <= 2e-5 dB): max |delta dB| and the fraction within 1e-3 dB over the bins the
    SURVEY.md section 8d tolerance names (value > floor + 20 dB)."""
    import numpy as np
    from audio_analysis_amd.analyse.spectrogram import select_stft_segments
    sp = settings.spectrogram
    t = eng.torch
    k = min(channels, batch.count)
    starts, lens, nframes = select_stft_segments(eng, batch, settings.sample_rate_hz, sp, "spectrogram")
    starts, nframes = starts[:k], nframes[:k]
    off = batch.off[:k] + starts
    f = sp.n_fft // 2 + 1
    a, a_off, cols = eng.stft_mag_db(batch.x, off, nframes, sp.n_fft, sp.hop_length, sp.use_hann_window, sp.floor_db,
                                     32, frame_major=eng.stft_frame_major_ok(sp.n_fft, 32))
    r, r_off, _ = eng.stft_mag_db(batch.x, off, nframes, sp.n_fft, sp.hop_length, sp.use_hann_window, sp.floor_db, 64)
    eng.sync()
    worst, inside, total = 0.0, 0, 0
    host = []                                                 # the float32 (F, T) matrices, for the comparison with the oracle's
    for i in range(k):
        T = int(cols[i])
        ai = a[int(a_off[i]) : int(a_off[i]) + f * T]
        ai = ai.view(T, f).t() if eng.stft_frame_major_ok(sp.n_fft, 32) else ai.view(f, T)
        host.append(ai.cpu().numpy().copy())
        ri = r[int(r_off[i]) : int(r_off[i]) + f * T].view(f, T)
        mask = ri > (sp.floor_db + 20.0)
        err = (ai.double() - ri.double()).abs()[mask]
        if err.numel():
            worst = max(worst, float(err.max().item()))
            inside += int((err <= 1e-3).sum().item())
            total += int(err.numel())
    return ({"against": "float64 STFT of the same samples on the device (pinned to the oracle at <= 2e-5 dB by the parity suite)",
             "bins": total, "bins_rule": "reference value > floor_db + 20 dB (SURVEY.md 8d)", "channels": k,
             "max_abs_err_db": worst, "fraction_within_1e-3_db": (inside / total) if total else None}, host)


def narrow_jobs_of(eng):
    """Band-inverse jobs of the last step that skipped the first pass (the device's job_info records, ira.h):
    (two-band jobs, single-band half-length jobs), or None with the narrow-band path off."""
    try:
        cnt = [0, 0]
        for info, half in zip(eng.last_band_info, eng.last_band_info_half):
            if info is not None:
                cnt[1 if half else 0] += int(info.view(-1, 4)[:, 0].sum().item())
        return tuple(cnt)
    except Exception:
        return None


def roofline_kernel(cfg, tot, roof, geom=None):
    """`roofline` is per ABI CALL (a call may be several kernels: ira_rfft_any is three).  This entry is per KERNEL: the
    kernel rocprofv3 ranks first in the committed profile of this configuration.  When its call is a single launch the
    kernel is timed live (the call's HIP-event time is the kernel's own duration).  When the call is several launches no
    live per-kernel time exists: the entry then carries the kernel's AVERAGE DURATION FROM THE COMMITTED PROFILE with the
    bytes one launch of it streams (geom: channels, n, bands), and the live roofline of its call beside it."""
    dom = dominant_kernel_of_profile(cfg)
    if dom is None:
        return None
    kname, share, src, targs, avg_ms, calls = dom
    # launches and milliseconds per step of that kernel from the per-kernel table of the same profile (tools/traffic_profile.py)
    per_step = None
    try:
        import csv
        tab = os.path.join(REPO, "profiles", src.replace("_kernel_stats_", "_per_kernel_"))
        for r in csv.DictReader(open(tab)):
            if r["kernel"].strip('"').replace(" ", "") == (kname + targs).replace(" ", ""):
                per_step = {"launches_per_step": float(r["launches_per_step"]), "ms_per_step": float(r["ms_per_step"])}
                break
    except Exception:
        per_step = None
    call = next((c for c, k in SINGLE_KERNEL_CALLS.items() if k == kname), None)
    live = next((n for n in tot if call and n.startswith(call)), None)
    out = {"kernel": kname + targs, "share_of_device_time_in_profile": share, "profile": f"profiles/{src}"}
    if live is None:
        mcall = next((c for k, c in MULTI_KERNEL_CALLS.items() if (kname + targs).startswith(k)), None)
        mlive = next((n for n in tot if mcall and n.startswith(mcall)), None)
        out.update(duration_source=f"average of {calls} launches in profiles/{src} (rocprofv3 --kernel-trace --stats): its call "
                                   f"{mcall} is several launches, so HIP events around the call do not time one kernel",
                   avg_launch_ms=avg_ms, call=mcall,
                   ms_per_step=None if per_step is None else per_step["ms_per_step"],
                   launches_per_step=None if per_step is None else per_step["launches_per_step"],
                   ms_per_step_source=f"profiles/{src.replace('_kernel_stats_', '_per_kernel_')} (same profiled run)")
        if geom is not None and mcall is not None:
            nchan, n, nb = geom["channels"], geom["n"], geom["bands"]
            b, what = None, None
            # jobs that skip the first pass (narrow bands, counted on the device) leave the regular kernels at once
            npair, nhalf = geom.get("narrow_jobs") or (0, 0)
            if (kname + targs).startswith("smooth_cols_kernel<1"):
                jobs = nchan * (nb // 2) - npair
                b, what = jobs * (16.0 * (n // 2 + 1) + 16.0 * n), (f"{jobs} two-band jobs: 16(n/2+1) B of spectrum in + 16n B of "
                                                                    f"column-transformed work array out")
            elif (kname + targs).startswith("smooth_rows_kernel<1"):
                jobs = nchan * (nb // 2) - npair
                b, what = jobs * (16.0 * n + 8.0 * n), f"{jobs} two-band jobs: 16n B of work array in + two float32 band signals out"
            elif kname == "smooth_rows_sparse_kernel" and (npair or nhalf):
                b, what = (npair * 8.0 + nhalf * 4.0) * n, (f"{npair} narrow two-band jobs + {nhalf} narrow single-band jobs "
                                                            f"(counted on the device, job_info): float32 band signals out; "
                                                            f"their input is 2(w + n2) compacted bins per job, read from L2")
            elif kname == "edc_moments_kernel" or kname == "edc_sums_kernel":
                segs = nchan * (nb + (1 if geom.get("decay") else 0))
                b, what = segs * 4.0 * float(geom["mean_len"]), f"{segs} segments: 4L B of samples read once"
            if b is not None:
                ach = b / (avg_ms * 1e-3) / 1e9
                out.update(bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                           algorithmic_bytes=b, algorithmic=what + " (streaming model of this pass; the call's compulsory "
                                                                  "bytes are in call_roofline)")
        if mlive is not None:
            out["call_roofline"] = roof(mlive)
        return out
    r = roof(live)
    r["call"] = r.pop("kernel")
    out.update(r)
    if "flops_model" in r:
        fm = r["flops_model"]
        out.update(bound="vector-f64", achieved=fm["achieved_TFLOPs"], peak=fm["f64_vector_peak_TFLOPs"], unit="TFLOP/s",
                   frac=fm["frac_of_f64_vector_peak"], hbm_frac=r.get("frac"))
    return out


# ---------------------------------------------------------------------------------------------------------
def make_bundle(root, taps: int, distinct: int, frames: int, first_index: int):
    """A bundle in the reference recorder's on-disk format (recorder.hpp:55-126): stereo PCM16 taps + meta.json.
    `distinct` different synthetic taps are written; the rest are hard links to them (same page-cache pages)."""
    import numpy as np
    from audio_analysis_amd.synth import synth_ir
    import struct
    os.makedirs(os.path.join(root, "taps"), exist_ok=True)
    names = [f"tap{i:05d}" for i in range(taps)]
    for i in range(taps):
        path = os.path.join(root, "taps", names[i] + ".wav")
        if i >= distinct:
            os.link(os.path.join(root, "taps", names[i % distinct] + ".wav"), path)
            continue
        lr = np.stack([synth_ir(first_index + i, c, frames) for c in range(2)], axis=1)
        pcm = (lr * np.float32(32767.0)).astype(np.int16)                    # recorder.hpp:49-53
        data = pcm.tobytes()
        hdr = b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, 48000,
                                                                                       48000 * 4, 4, 16)
        hdr += b"data" + struct.pack("<I", len(data))
        with open(path, "wb") as fh:
            fh.write(hdr + data)
    with open(os.path.join(root, "meta.json"), "w") as fh:
        json.dump({"sample_rate_hz": 48000, "length_samples": frames, "taps": names}, fh)


def note(msg):
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside a torchrun environment: start the N rank processes ourselves (one per GPU, fresh
    interpreters started BEFORE this process has imported torch or touched a GPU; never an exec), hand them the
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment torch.distributed.run would, and wait.  Rank 0 inherits stdout,
    so its one JSON line is this command's output.  Returns the largest exit code; when a rank dies the others (which
    would wait for it in the next barrier forever) are terminated."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            rc = max(rc, abs(code))
            if code != 0:
                for q in live:
                    q.terminate()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="report", choices=["report", "literal", "2", "3", "4", "5"])
    ap.add_argument("--batch", type=int, default=None, help="IRs (config 5: stereo taps) per GPU per step")
    ap.add_argument("--seconds", type=float, default=None, help="IR length")
    ap.add_argument("--host-batches", type=int, default=4, help="distinct batches rotating in pinned host memory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=4,
                    help="steps of the serialised (one stream) pass that measures per-kernel durations for the rooflines")
    ap.add_argument("--literal-steps", type=int, default=10,
                    help="config report only: extra steps with the reference's default-on group-delay and diffusion "
                         "blocks added (reported as literal_full_report; 0 = skip)")
    ap.add_argument("--upload", default="auto", choices=["auto", "pull", "copy"],
                    help="pull: the batch crosses PCIe under a pull kernel (ira_host_pull); copy: hipMemcpyAsync on the copy "
                         "engines, --upload-streams pieces; auto: copy, unless the warm-up finds the copies starved under the "
                         "kernels on this box (DeviceFeed.autotune: then one piece or the pull kernel, whichever steps fastest)")
    ap.add_argument("--pull-workgroups", type=int, default=8)
    ap.add_argument("--upload-streams", type=int, default=2,
                    help="float32 / int16 upload as this many pieces on as many copy streams (A/B)")
    ap.add_argument("--upload-priority", action="store_true", help="copy streams created with high priority (A/B)")
    ap.add_argument("--variants", default="all", choices=["all", "value", "resident"],
                    help="'value' skips the int16 / resident variants (profiling runs); 'resident' runs the headline and the "
                         "resident-input variant only (upload A/B runs)")
    ap.add_argument("--gather", default="final", choices=["final", "step"],
                    help="final: the records of every step stay on their rank and ONE gather to rank 0 closes the timed "
                         "region (north star: a single RCCL gather for the final metrics); step: one gather per step (A/B)")
    ap.add_argument("--ar-order", type=int, default=None,
                    help="AR order of the z-plane block (the configs use 64; the reference's CLI default is 256, cli.py:245)")
    ap.add_argument("--spawn-probe", action="store_true",
                    help="launch check without a GPU: every rank joins a gloo group, rank 0 prints the ranks it gathered "
                         "(tests/test_host_cpu.py runs `bench.py --gpus 2 --spawn-probe`)")
    a = ap.parse_args()
    a.roofline_steps = max(1, a.roofline_steps)      # the bench line needs the serialised pass (roofline, per-call times)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    if a.spawn_probe:
        from audio_analysis_amd import dist as D
        _, lr0, _ = D.env_world()
        my_cpus, how, before = D.apply_rank_cpu_affinity(lr0, int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
        import numpy as np
        rank, local_rank, world = D.init_process_group("gloo")
        got = D.gather_metrics(np.full((3, 4), float(rank)), equal_rows=True)
        row = np.full((1, 64), -1.0)
        now = sorted(os.sched_getaffinity(0))                        # what the kernel says, not what we asked for
        row[0, : min(64, len(now))] = now[:64]
        cpus = D.gather_metrics(row, equal_rows=True)
        D.barrier()
        if rank == 0:
            sets = [sorted(int(c) for c in r if c >= 0) for r in cpus]
            print(json.dumps({"n_gpus": world, "ranks": sorted(set(got[:, 0].tolist())), "rows": int(got.shape[0]),
                              "affinity": sets, "how": how, "cpus_before": len(before),
                              "disjoint": all(set(sets[i]).isdisjoint(sets[j]) for i in range(len(sets)) for j in range(i))}))
        return

    # ---- one rank = one GPU = the cores of that GPU's NUMA node: BEFORE torch is imported and before any pinned host batch
    # is allocated (first touch puts its pages on the node of the allocating thread), for ranks started by spawn_ranks and
    # by torch.distributed.run alike
    from audio_analysis_amd import dist as D
    _, lr0, _ = D.env_world()
    lw0 = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    my_cpus, affinity_how, cpus_before = D.apply_rank_cpu_affinity(lr0, lw0)
    os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, len(my_cpus)))))

    import numpy as np
    import torch

    from audio_analysis_amd.engine import Engine
    from audio_analysis_amd.feed import DeviceFeed, HostBatch, run_pipelined
    from audio_analysis_amd.pipeline import METRICS_WIDTH, FullReport
    from audio_analysis_amd.synth import synth_ir

    cfg = config_table()[a.config]
    settings = cfg["settings"]
    if a.ar_order is not None and a.ar_order != settings.zplane.ar_order:
        from dataclasses import replace as _rep
        settings = _rep(settings, zplane=_rep(settings.zplane, ar_order=int(a.ar_order)))
        cfg = dict(cfg, settings=settings, metric=cfg["metric"].replace("AR(64)", f"AR({a.ar_order})") + f" [AR order {a.ar_order}]",
                   what=cfg["what"].replace("order 64", f"order {a.ar_order}"),
                   # the oracle's SVD least squares costs ~ rows x order^2: 4 s per 10 s channel at order 64
                   cpu_s=cfg["cpu_s"] + 4.0 * ((a.ar_order / 64.0) ** 2 - 1.0) * (settings.run_zplane))
    B = a.batch or cfg["batch"]
    seconds = a.seconds or cfg["seconds"]
    steps = a.steps or cfg["steps"]
    n = int(seconds * 48000)

    rank, local_rank, world = D.init_process_group()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    if os.environ.get("IRA_DIST_BACKEND") == "gloo":          # rehearsal only (dist.init_process_group): ranks may share a device
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    eng = Engine(f"cuda:{local_rank}")
    import audio_analysis_amd.engine as _engine_mod
    _engine_mod._ENGINE = eng                      # the process-wide engine (bundle.run_bundle_metrics asks get_engine())
    report = FullReport(eng, settings)
    blocks = block_names(settings)

    # every rank's CPU set travels to rank 0 with one small gather (first cpu, last cpu, count, 64-bit masks of cpus 0..255)
    aff_row = np.zeros((1, 8), dtype=np.float64)
    aff_row[0, :4] = [rank, my_cpus[0] if my_cpus else -1, my_cpus[-1] if my_cpus else -1, len(my_cpus)]
    for c in my_cpus:
        if c < 208:
            aff_row[0, 4 + c // 52] += float(1 << (c % 52))           # 52-bit pieces: exact in a float64
    aff_all = D.gather_metrics(aff_row, eng.device, equal_rows=True)
    affinity = None
    if rank == 0:
        ranks = []
        for r in aff_all:
            cpus = [52 * j + b for j in range(4) for b in range(52) if (int(r[4 + j]) >> b) & 1]
            ranks.append({"rank": int(r[0]), "cpus": int(r[3]), "first": int(r[1]), "last": int(r[2]), "_set": cpus})
        sets = [set(x.pop("_set")) for x in ranks]
        affinity = {"how_rank0": affinity_how, "cpus_allowed_before": len(cpus_before), "ranks": ranks,
                    "disjoint": all(sets[i].isdisjoint(sets[j]) for i in range(len(sets)) for j in range(i))
                                if world > 1 else True}
    if a.config == "5":
        return bench_bundle(a, cfg, eng, rank, world, B, n, steps, blocks, affinity, cpus_before)

    # ---- K distinct synthetic batches per rank in pinned host memory (float32 and the PCM16 the recorder would store) -----
    K = max(1, a.host_batches)
    from concurrent.futures import ThreadPoolExecutor
    first = rank * K * B
    with ThreadPoolExecutor(max_workers=min(16, len(os.sched_getaffinity(0)))) as ex:
        chans = list(ex.map(lambda i: synth_ir(first + i, 0, n), range(K * B)))
    host_f32 = [HostBatch(eng, np.stack(chans[k * B : (k + 1) * B])) for k in range(K)]
    host_i16 = None
    if a.variants == "all":
        host_i16 = [HostBatch(eng, (np.stack(chans[k * B : (k + 1) * B]) * np.float32(32767.0)).astype(np.int16), pcm16=True)
                    for k in range(K)]
    del chans
    note(f"{K} host batches of {B} x {seconds:g} s synthesised and pinned")
    feed = DeviceFeed(eng, B * n, depth=4, pull=(a.upload == "pull"), pull_workgroups=a.pull_workgroups,
                      copy_streams=a.upload_streams, high_priority=a.upload_priority)
    last = {}
    kept = []                                          # this rank's records of the steps since the last flush
    counters = {"steps": 0}

    def gather(rec):
        counters["steps"] += 1
        if a.gather == "step":
            last["g"] = D.gather_metrics(rec, eng.device, eng.side_stream(), equal_rows=True)
        else:
            kept.append(rec)

    def flush():
        """--gather final: ONE gather of all the records this rank produced since the last flush (every rank holds the
        same number of rows: no count exchange, no host synchronisation before the collective)."""
        if a.gather == "final" and kept:
            last["g"] = D.gather_metrics(np.concatenate(kept, axis=0), eng.device, equal_rows=True)
        kept.clear()

    def run_fed(count, host, rep=None):
        run_pipelined(rep or report, feed, (host[i % K] for i in range(count)), gather)

    def timed(fn, count):
        flush()
        D.barrier(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(count)
        flush()                                        # inside the timed region: the gather is part of the job
        D.barrier(); torch.cuda.synchronize()
        return D.max_over_ranks(time.perf_counter() - t0, eng.device)

    # ---- headline: H2D-inclusive, float32 upload ---------------------------------------------------------------------------
    # Plan pass (untimed, like the warm-up): every distinct batch once, so that plan data keyed by the data-dependent
    # segment lengths (chirp-filter spectra of the arbitrary-length transforms: an LRU pool, Engine._filters) and the
    # caching allocator's block sizes exist before the W warm-up steps -- a long-running job is in that state.
    run_fed(K, host_f32)
    run_fed(a.warmup, host_f32)
    D.barrier(); torch.cuda.synchronize()
    # the link alone: three uploads of a batch with nothing else on the GPU, HIP events on the copy streams (DeviceFeed.timing)
    feed.timing = []
    for k in range(3):
        feed.push(host_f32[k % K])
        torch.cuda.synchronize()
    upload_alone = feed.upload_times()
    upload_tuning = None
    if a.upload == "auto":
        def tune_steps(c):
            # (records dropped, no collective: ranks may try different numbers of arms, and the final gather expects every
            # rank to hold the same number of rows)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            run_pipelined(report, feed, (host_f32[i % K] for i in range(c)), None)
            torch.cuda.synchronize()
            return time.perf_counter() - t0
        upload_tuning = feed.autotune(tune_steps, upload_alone["GBps"] if upload_alone else None,
                                      first_pieces=a.upload_streams)
        note("upload: " + "; ".join(f"{r['mode']} {r['ms_per_step']:.2f} ms/step" + (" (chosen)" if r["chosen"] else "") for r in upload_tuning))
        flush()
        D.barrier(); torch.cuda.synchronize()
    eng.events = []
    feed.timing = []                                   # ... and under the kernels of the timed region
    elapsed = timed(lambda c: run_fed(c, host_f32), steps)
    upload_timed = feed.upload_times()
    feed.timing = None
    ev_timed = eng.collect_events()
    eng.events = None
    gathered = last.get("g")
    note(f"timed region: {steps} steps in {elapsed:.3f} s = {B * world * steps / elapsed:.0f} IRs/s")

    el_pull = None
    if a.variants == "all" and a.upload != "pull" and not feed.pull:
        feed.pull = True                               # A/B: the same steps with the pull-kernel upload (ira_host_pull)
        run_fed(2, host_f32)
        el_pull = timed(lambda c: run_fed(c, host_f32), steps)
        feed.pull = False
    # ---- variants: int16 upload; inputs resident in HBM (rotating over the K device-resident batches) ----------------------
    el_i16 = el_res = None
    resident = None
    if a.variants in ("all", "resident"):
        if a.variants == "all":
            run_fed(2, host_i16)
            el_i16 = timed(lambda c: run_fed(c, host_i16), steps)
        resident = []
        for k in range(K):
            resident.append(eng.wrap(eng.to_dev(host_f32[k].pinned.numpy()[: B * n].copy()), host_f32[k].off, host_f32[k].length))
        torch.cuda.synchronize()

        def run_resident(count, rep=None):
            rep = rep or report
            pending = None
            for i in range(count):
                b = resident[i % K]
                b.peak = None                                  # the peak pick is part of every step
                h = rep.submit(b)
                if pending is not None:
                    gather(rep.finish(pending))
                pending = h
            if pending is not None:
                gather(rep.finish(pending))

        run_resident(2)
        el_res = timed(run_resident, steps)

    # ---- cold plan data: the same steps with the chirp-filter pools forgotten before every step, i.e. every fr / filter
    # segment length is NEW to the engine (they are data dependent: N - argmax|x|; a bundle of real taps sees new lengths
    # all the time) and ira_bluestein_filter runs inside the timed region
    el_cold = ev_cold = None
    if a.variants == "all":
        def cold_batches(count):
            for i in range(count):
                eng.forget_filters()
                yield host_f32[i % K]
        run_pipelined(report, feed, cold_batches(2), gather)
        eng.events = []
        el_cold = timed(lambda c: run_pipelined(report, feed, cold_batches(c), gather), steps)
        ev_cold = eng.collect_events()
        eng.events = None
    note("variants done" + ("" if el_cold is None else f": int16 {B * world * steps / el_i16:.0f}, resident {B * world * steps / el_res:.0f}, "
                                                      f"new lengths every step {B * world * steps / el_cold:.0f} IRs/s"))
    # ---- per-kernel durations: a short SERIALISED pass (one stream, kernels one at a time) in the same run ---------------
    lanes_used = eng.num_lanes
    eng.num_lanes = 1
    run_fed(K, host_f32)                               # every distinct batch once: plan data (chirp filters) of this stream
    D.barrier(); torch.cuda.synchronize()
    eng.events = []
    run_fed(a.roofline_steps, host_f32)
    D.barrier(); torch.cuda.synchronize()
    ev = eng.collect_events()
    ev_cold_serial = None
    if a.variants == "all":
        eng.events = []
        run_pipelined(report, feed, cold_batches(a.roofline_steps), gather)
        D.barrier(); torch.cuda.synchronize()
        ev_cold_serial = eng.collect_events()
    eng.events = None
    eng.num_lanes = lanes_used
    roof_steps = a.roofline_steps

    # ---- second, shorter measurement: the LITERAL default report (group delay + diffusion blocks added) ------------------
    literal = None
    if a.config == "report" and a.literal_steps > 0 and a.variants == "all":
        from dataclasses import replace as _replace
        rep2 = FullReport(eng, _replace(settings, run_group_delay=True, run_diffusion=True))
        run_fed(max(2, K), host_f32, rep2)            # plan pass: every rotating batch once (its lengths' plan data), untimed
        el2 = timed(lambda c: run_fed(c, host_f32, rep2), a.literal_steps)
        literal = {"value": B * world * a.literal_steps / el2, "unit": "IRs/s", "steps": a.literal_steps,
                   "ms_per_step": 1e3 * el2 / a.literal_steps, "blocks": rep2.s.blocks(),
                   "note": "same step (H2D included) plus the reference's default-on group-delay and diffusion blocks "
                           "(SURVEY.md 8f); only the IR waveform plots and PNG rendering remain excluded"}
        # per-call device times of the literal step: the same serialised (one stream) pass as the headline's
        eng.num_lanes = 1
        run_fed(K, host_f32, rep2)
        D.barrier(); torch.cuda.synchronize()
        eng.events = []
        run_fed(a.roofline_steps, host_f32, rep2)
        D.barrier(); torch.cuda.synchronize()
        ev_lit = eng.collect_events()
        eng.events = None
        eng.num_lanes = lanes_used
        literal["_events"] = ev_lit
        literal["_settings"] = rep2.s

    flush()
    stft_err, stft_probe = None, []
    if settings.run_spectrogram and a.variants == "all":      # (profiling runs skip the probe: its extra peak pick and STFT
        probe = feed.push(host_f32[0])                        # launches would be counted as a step by the traffic tools)
        eng.peaks_begin(probe); eng.peaks(probe)
        stft_err, stft_probe = stft_error_distribution(eng, probe, settings)

    if rank != 0:
        return
    rows = B * world * (steps if a.gather == "final" else 1)
    assert gathered is not None and gathered.shape == (rows, METRICS_WIDTH), (None if gathered is None else gathered.shape, rows)
    assert np.all(gathered[:, 0] == 0.0), "a channel of the timed region did not report status ok"
    total_irs = B * world * steps
    # trimmed lengths of the batch the serialised pass saw last (synthetic pre-delays are 240 + i mod 512)
    pre = np.array([240 + ((first + i) % 512) for i in range(B)], dtype=np.float64)
    L = n - pre
    traffic_tab = load_traffic(a.config)
    tot, roof = make_roof(ev, roof_steps, settings, L, n, B, traffic_tab)
    dev_ms = sum(tot.values())
    analysis = {k: v for k, v in tot.items() if k not in INGEST_CALLS}
    dominant = max(analysis, key=analysis.get)
    stft_name = next((k for k in tot if k.startswith("ira_stft_mag_db") and "[f32" in k), None)
    rs = roof(stft_name) if stft_name else None
    if rs is not None and stft_err is not None:
        rs["f32_error"] = stft_err
    out = {
        "metric": cfg["metric"],
        "value": total_irs / elapsed,
        "unit": "IRs/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{cfg['what']} on {B} mono synthetic IRs per GPU per step, {seconds:g} s @ 48 kHz (SURVEY.md 8d "
                        f"generator), H2D included: every step uploads its batch from pinned host memory (float32, "
                        f"{B * n * 4 / 1e6:.0f} MB) on a copy stream under the previous step's kernels; {K} distinct "
                        f"batches rotate",
            "baseline_config": a.config,
            "blocks": settings.blocks(),
            "excluded": cfg["excluded"],
            "batch_per_gpu": B, "ir_seconds": seconds, "parallelism": f"file-sharded dp{world}",
            "arithmetic": "f64 (EDC scan, long FFTs, modal/waterfall STFT, AR Gram/solve/roots); "
                          "f32 butterflies for the spectrogram STFT",
            "timed_region": "H2D + peak pick + all kernels + metric pack + D2H of records + "
                            + ("ONE gather of every step's records to rank 0 after the last step" if a.gather == "final"
                               else "a gather to rank 0 per step"),
            "gather": a.gather,
        },
        "steps_in_process": counters["steps"],
        "value_int16": None if el_i16 is None else total_irs / el_i16,
        "value_resident": None if el_res is None else total_irs / el_res,
        "value_pull_kernel": None if el_pull is None else total_irs / el_pull,
        "upload": "pull kernel (ira_host_pull reads pinned host memory over PCIe)" if feed.pull else
                  f"hipMemcpyAsync, {len(feed.side_streams) + 1} piece(s) on as many copy streams",
        "upload_autotune": upload_tuning,
        "variants": {"value": "float32 upload inside the timed region (SURVEY.md 8d)",
                     "value_int16": "PCM16 upload (2 B/sample) + device conversion inside the timed region",
                     "value_resident": "no upload: the same distinct batches already in HBM (compute-only rate)",
                     "value_pull_kernel": "float32 upload by the pull kernel (ira_host_pull) instead of hipMemcpyAsync (A/B)"},
        "h2d_GBps": B * n * 4.0 * steps / elapsed / 1e9,
        "h2d_GBps_is": "upload bytes / wall time of the timed region (what the link SUSTAINED, not what a copy ran at: see upload)",
        "roofline": roof(dominant),
        "roofline_kernel": roofline_kernel(a.config, tot, roof, dict(channels=B, n=n, bands=nb_bands(settings),
                                                                    decay=settings.run_decay, mean_len=float(np.mean(L)),
                                                                    narrow_jobs=narrow_jobs_of(eng))),
        "roofline_stft": rs,
        "roofline_measured": f"serialised pass of {roof_steps} steps in this run (one stream, kernels one at a time, H2D "
                             f"included); the timed region deals the report blocks onto {lanes_used} streams",
        "lanes": lanes_used,
        "device_ms_per_step_by_call": {k: v / roof_steps for k, v in sorted(tot.items(), key=lambda kv: -kv[1])},
        "device_ms_per_step": dev_ms / roof_steps,
        "timed_region_ms_per_step_by_call": {k: sum(v) / steps for k, v in
                                             sorted(ev_timed.items(), key=lambda kv: -sum(kv[1]))},
        "literal_full_report": literal,
    }
    out["affinity"] = affinity
    # ---- what the upload cost: events around every piece on the copy streams ----------------------------
    step_ms = 1e3 * elapsed / steps
    res_ms = None if el_res is None else 1e3 * el_res / steps
    up = {"pieces": upload_timed["pieces_per_upload"] if upload_timed else None, "copy_streams": len(feed.side_streams) + 1,
          "high_priority_streams": bool(a.upload_priority),
          "bytes_per_step": B * n * 4.0,
          "alone": upload_alone, "under_compute": upload_timed,
          "measured": "HIP events recorded on the copy streams around every piece (DeviceFeed.timing): ms_per_upload = first "
                      "piece's start to last piece's end; alone = three uploads with an otherwise idle GPU before the timed "
                      "region; under_compute = every upload of the timed region"}
    out["upload_ms_per_step"] = upload_timed["ms_per_upload"] if upload_timed else None
    out["h2d_alone_GBps"] = upload_alone["GBps"] if upload_alone else None
    out["h2d_under_compute_GBps"] = upload_timed["GBps"] if upload_timed else None
    if upload_timed:
        busy = upload_timed["ms_per_upload"] / step_ms
        up["copy_engines_busy_fraction_of_step"] = busy
        if res_ms is not None:
            up["GBps_needed_for_resident_rate"] = B * n * 4.0 / (res_ms * 1e-3) / 1e9
            up["resident_ms_per_step"] = res_ms
        # a step waits for its upload when the copy of one batch takes (nearly) as long as the step itself -- or longer than
        # the same step takes with its inputs already resident
        out["bound"] = "pcie" if (busy >= 0.85 or (res_ms is not None and upload_timed["ms_per_upload"] >= 0.95 * res_ms)) else "compute"
        out["bound_reason"] = (f"one batch's upload takes {upload_timed['ms_per_upload']:.2f} ms under the kernels "
                               f"({upload_timed['GBps']:.1f} GB/s; {upload_alone['GBps']:.1f} GB/s alone) of a "
                               f"{step_ms:.2f} ms step" + ("" if res_ms is None else f"; the same steps with resident inputs take {res_ms:.2f} ms"))
    out["upload_detail"] = up
    if el_cold is not None:
        out["value_new_lengths"] = total_irs / el_cold
        out["variants"]["value_new_lengths"] = ("the same steps with the chirp-filter pools forgotten before every step: every "
                                                "fr / filter segment length is new to the engine and ira_bluestein_filter runs "
                                                "inside the timed region (a job whose lengths never repeat)")
        tot_c = {k: sum(v) / roof_steps for k, v in ev_cold_serial.items()}
        out["new_lengths"] = {"ms_per_step": 1e3 * el_cold / steps,
                              "device_ms_per_step_by_call": dict(sorted(tot_c.items(), key=lambda kv: -kv[1])),
                              "device_ms_per_step": sum(tot_c.values()),
                              "ira_bluestein_filter_ms_per_step": tot_c.get("ira_bluestein_filter"),
                              "timed_region_ira_bluestein_filter_ms_per_step":
                                  sum(ev_cold.get("ira_bluestein_filter", [])) / steps}
    if literal is not None:
        ev_lit, set_lit = literal.pop("_events"), literal.pop("_settings")
        tot_l, roof_l = make_roof(ev_lit, roof_steps, set_lit, L, n, B, traffic_tab)
        literal["device_ms_per_step_by_call"] = {k: v / roof_steps for k, v in sorted(tot_l.items(), key=lambda kv: -kv[1])}
        literal["device_ms_per_step"] = sum(tot_l.values()) / roof_steps
        literal["added_blocks_device_ms"] = {
            "groupdelay": sum(v for k, v in tot_l.items() if "[gd]" in k or k.startswith("ira_group_delay")
                              or k.startswith("ira_order_stats")) / roof_steps,
            "diffusion": sum(v for k, v in tot_l.items() if k.startswith("ira_diffusion")) / roof_steps}
        literal["rooflines"] = [roof_l(k) for k in tot_l if "[gd]" in k or k.startswith("ira_group_delay")
                                or k.startswith("ira_diffusion")]
    note("device side done; cpu baseline next")
    if not a.no_cpu_baseline:
        bm = settings.rt60_bands.band_mode
        row0 = first + ((0 if a.gather == "final" else steps - 1) % K) * B          # generator index of gathered row 0
        base, oracle_values, oracle_specs = cpu_baseline(seconds, blocks, bm, False, cfg["cpu_s"], "IRs/s", first_index=row0,
                                                         spec_probe=len(stft_probe) if row0 == first else 0, allowed_cpus=cpus_before if world > 1 else None,
                                                         share_cap=16 * world)
        if world > 1:
            base["sample"] += f"; run on rank 0 after the device side of all {world} ranks, over the host share of {world} GPUs"
        out["cpu_baseline"] = base
        # the timed region's step s analysed host batch s % K = generator indices first + (s % K) B ...: rank 0's rows of
        # step 0 are the files the workers just analysed
        vals = [v[0] for v in oracle_values][:B]
        out["parity"] = parity_report(gathered[: len(vals)], vals, nb_bands(settings))
        out["parity"]["sample"] = (f"the GPU's gathered records of generator indices {row0}..{row0 + len(vals) - 1} (rank 0, "
                                   f"{'first' if a.gather == 'final' else 'last'} step of the timed region) against the oracle values the cpu_baseline workers returned "
                                   f"for the same indices")
        if oracle_specs and rs is not None:
            rs["f32_error_vs_device_f64"] = rs.get("f32_error")
            rs["f32_error"] = spectrogram_parity(stft_probe[: len(oracle_specs)], oracle_specs, settings.spectrogram.floor_db)
    print(json.dumps(out))


def nb_bands(settings) -> int:
    if not settings.run_rt60_bands:
        return 0
    from audio_analysis_amd.analyse.rt60bands import _build_band_definitions
    return len(_build_band_definitions(settings.rt60_bands, settings.sample_rate_hz))


def bench_bundle(a, cfg, eng, rank, world, B, n, steps, blocks, affinity=None, cpus_before=None):
    """BASELINE config 5: stereo PCM16 tap files -> bundle.run_bundle_metrics.  A step = one group of B tap files."""
    import shutil
    import tempfile

    import numpy as np
    import torch

    from audio_analysis_amd import dist as D
    from audio_analysis_amd.analyse.bundle import run_bundle_metrics
    from audio_analysis_amd.pipeline import METRICS_WIDTH

    settings = cfg["settings"]
    base = tempfile.mkdtemp(prefix=f"ira_bundle_r{rank}_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        distinct = min(4 * B, 128)
        # every rank analyses its own bundle directory of W*B resp. K*B taps (= its contiguous shard of a W*B*world bundle)
        make_bundle(os.path.join(base, "warm"), max(1, a.warmup) * B, distinct, n, 100000 + rank * 4096)
        make_bundle(os.path.join(base, "timed"), steps * B, distinct, n, 200000 + rank * 4096)
        # every rank owns a whole private bundle here (its shard of the job), so run_bundle_metrics is told it is alone
        # and the gather of the records is done once below
        def run(path):
            return run_bundle_metrics(path, settings, taps_per_step=B, rank_world=(0, 1), gather=False)

        run(os.path.join(base, "warm"))
        D.barrier(); torch.cuda.synchronize()
        eng.events = []
        os.environ["IRA_BUNDLE_TIMING"] = "1"          # host milliseconds per phase of the loop (a dozen clock reads per group)
        t0 = time.perf_counter()
        labels, local = run(os.path.join(base, "timed"))
        gathered = D.gather_metrics(local, eng.device)
        D.barrier(); torch.cuda.synchronize()
        elapsed = D.max_over_ranks(time.perf_counter() - t0, eng.device)
        os.environ.pop("IRA_BUNDLE_TIMING", None)
        from audio_analysis_amd.analyse import bundle as _bundle
        host_phases = _bundle.LAST_HOST_MS_PER_GROUP
        ev_timed = eng.collect_events()
        eng.events = None
        # serialised pass for the per-call durations
        lanes_used = eng.num_lanes
        eng.num_lanes = 1
        eng.events = []
        run(os.path.join(base, "warm"))
        ev = eng.collect_events()
        eng.events = None
        eng.num_lanes = lanes_used
        roof_steps = max(1, a.warmup)
    finally:
        shutil.rmtree(base, ignore_errors=True)
    if rank != 0:
        return
    assert gathered.shape == (2 * B * steps * world, METRICS_WIDTH)
    pre = np.repeat(np.array([240 + ((100000 + i) % 512) for i in range(B)], dtype=np.float64), 2)
    L = n - pre
    traffic_tab = load_traffic("5")
    tot, roof = make_roof(ev, roof_steps, settings, L, n, 2 * B, traffic_tab)
    analysis = {k: v for k, v in tot.items() if k not in INGEST_CALLS}
    dominant = max(analysis, key=analysis.get)
    files = B * steps * world
    out = {
        "metric": cfg["metric"], "value": files / elapsed, "unit": "stereo taps/s", "n_gpus": world, "steps": steps,
        "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{cfg['what']}; {B} stereo taps of {n / 48000:g} s per step per GPU read from files on "
                               f"tmpfs (page-cache resident), {2 * B} channels per step; H2D (int16) included",
                   "baseline_config": "5", "blocks": settings.blocks(), "excluded": cfg["excluded"],
                   "batch_per_gpu": B, "ir_seconds": n / 48000, "parallelism": f"file-sharded dp{world}",
                   "timed_region": "file reads + H2D (int16) + device conversion + peak pick + all kernels + metric pack + "
                                   "D2H + gather"},
        "channels_per_s": 2 * files / elapsed,
        "roofline": roof(dominant),
        "roofline_kernel": roofline_kernel("5", tot, roof, dict(channels=2 * B, n=n, bands=nb_bands(settings),
                                                                decay=settings.run_decay, mean_len=float(np.mean(L)),
                                                                narrow_jobs=narrow_jobs_of(eng))),
        "roofline_measured": f"serialised pass ({roof_steps} steps, one stream) in this run",
        "affinity": affinity,
        "lanes": lanes_used,
        "device_ms_per_step_by_call": {k: v / roof_steps for k, v in sorted(tot.items(), key=lambda kv: -kv[1])},
        "device_ms_per_step": sum(tot.values()) / roof_steps,
        "host_ms_per_step_by_phase": host_phases,
    }
    # what binds the step: the serialised kernels' time against the step (two lanes overlap them further), beside the host's
    # own phases -- the host thread enqueues `view` + `prepare` + `submit` and blocks in `finish` / for the reader thread
    dev = out["device_ms_per_step"]
    out["bound"] = "compute" if dev >= 0.97 * out["ms_per_step"] else "host"
    out["bound_reason"] = (f"{out['ms_per_step']:.2f} ms per step against {dev:.2f} ms of kernels when they run one at a time "
                           f"(less on two lanes); host phases per step: {host_phases}")
    if not a.no_cpu_baseline:
        base, oracle_values, _ = cpu_baseline(n / 48000, blocks, settings.rt60_bands.band_mode, True, cfg["cpu_s"],
                                              "stereo taps/s", first_index=200000, pcm16=True,
                                              allowed_cpus=cpus_before if world > 1 else None, share_cap=16 * world)
        out["cpu_baseline"] = base
        # tap i of rank 0's timed bundle is generator index 200000 + i (the first `distinct` taps are distinct files), its two
        # channels are rows 2i and 2i + 1 of the gathered records
        vals = [v for per_file in oracle_values[: min(len(oracle_values), distinct)] for v in per_file]
        out["parity"] = parity_report(gathered[: len(vals)], vals, nb_bands(settings))
        out["parity"]["sample"] = (f"both channels of the first {len(vals) // 2} taps of rank 0's timed bundle (generator indices "
                                   f"200000.., PCM16 as the recorder stores it) against the oracle on the same PCM16 samples")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
