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
    index, seconds, blocks, band_mode, stereo = args
    from audio_analysis_amd.synth import synth_ir
    from oracle import ira_oracle as O
    chans = [synth_ir(index, c, int(seconds * 48000)) for c in range(2 if stereo else 1)]
    t0 = time.perf_counter()
    for x in chans:
        if "decay" in blocks:
            O.analyse_decay(x)
        if "rt60bands" in blocks:
            O.analyse_rt60_bands(x, band_mode=band_mode)
        if "fr" in blocks:
            O.analyse_frequency_response(x)
        if "filter" in blocks:
            O.analyse_filter_response(x)
        if "spectrogram" in blocks:
            O.analyse_spectrogram(x)
        if "waterfall" in blocks:
            O.analyse_waterfall(x)
        if "modalcloud" in blocks:
            O.analyse_modal_cloud(x)
        if "zplane" in blocks:
            O.analyse_zplane(x, ar_order=64)
    return time.perf_counter() - t0


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


def host_cpu_share():
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
    share = int(os.environ.get("IRA_BENCH_CPU_WORKERS", "16"))
    workers = max(1, min(affinity, quota or affinity, share))
    return {"affinity": affinity, "cgroup_quota": quota, "share_cap": share, "workers": workers}


def cpu_baseline(seconds: float, blocks, band_mode: str, stereo: bool, per_ir_guess_s: float, unit: str):
    """One worker process per core this process may run on (the GPU box's CPU share), single-threaded NumPy in each,
    >= 64 units of work (4 per worker); pool start-up and imports are outside the timed wall."""
    import multiprocessing as mp
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[k] = "1"
    avail = host_cpu_share()
    workers = avail["workers"]
    # bounded sample: >= 64 files, but no more than ~25 s of wall at the guessed per-file cost
    count = max(64, 4 * workers)
    cap = int(max(workers, 25.0 * workers / max(per_ir_guess_s, 1e-3)))
    count = max(workers, min(count, cap))
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers, initializer=_cpu_init) as pool:
        pool.map(_cpu_noop, range(4 * workers))                       # every worker has started and imported
        t0 = time.perf_counter()
        per = pool.map(_cpu_one, [(1000 + i, seconds, tuple(blocks), band_mode, stereo) for i in range(count)],
                       chunksize=1)
        wall = time.perf_counter() - t0
    return {
        "value": count / wall, "unit": unit, "cores": workers, "kind": "port", "cpu_model": cpu_model(),
        "logical_cpus_on_box": os.cpu_count(), "cpus_in_affinity_mask": avail["affinity"],
        "cgroup_cpu_quota": avail["cgroup_quota"], "worker_cap": avail["share_cap"],
        "sample": f"{count} synthetic {seconds:g} s {'stereo files' if stereo else 'mono IRs'} over {workers} worker "
                  f"processes (one per core this process may use; single-threaded NumPy oracle, same blocks as the GPU "
                  f"step: {','.join(blocks)}); mean {sum(per)/len(per):.2f} s per file per core, wall {wall:.1f} s "
                  f"(pool start-up excluded)",
    }


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
        "5": dict(settings=full, batch=128, seconds=5.0, steps=8, cpu_s=1.4,
                  metric="stereo taps/sec bundle report (full pipeline), 48 kHz 5 s stereo PCM16 taps (BASELINE config 5)",
                  what="bundle.run_bundle_metrics over stereo PCM16 tap files (native ingest, int16 upload, device "
                       "conversion, full metrics-only report of both channels)",
                  excluded=["png rendering", "group delay", "diffusion", "ir plots", "markdown"]),
    }


def block_names(settings):
    out = []
    for flag, name in (("run_decay", "decay"), ("run_rt60_bands", "rt60bands"), ("run_frequency_response", "fr"),
                       ("run_filter", "filter"), ("run_spectrogram", "spectrogram"), ("run_waterfall", "waterfall"),
                       ("run_modal_cloud", "modalcloud"), ("run_zplane", "zplane")):
        if getattr(settings, flag):
            out.append(name)
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
        elif name.startswith("ira_rfft_smooth"):
            # one real signal of even length = ONE half-length complex transform (x[2m] + i x[2m+1]) + the untangling pass
            h = n // 2
            stream_b = nchan * (4.0 * n + 16.0 * h + 16.0 * h + 16.0 * h + 16.0 * h + 16.0 * (h + 1))
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
    for rnd in ("r03", "r02"):
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
    configuration (profiles/rNN_kernel_stats_cfg<cfg>.csv): (short name, share of device time) or None."""
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
            best = (r["Name"], t)
    if best is None or total <= 0:
        return None
    name = best[0].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    return name, best[1] / total, os.path.basename(files[-1])


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
    for i in range(k):
        T = int(cols[i])
        ai = a[int(a_off[i]) : int(a_off[i]) + f * T]
        ai = ai.view(T, f).t() if eng.stft_frame_major_ok(sp.n_fft, 32) else ai.view(f, T)
        ri = r[int(r_off[i]) : int(r_off[i]) + f * T].view(f, T)
        mask = ri > (sp.floor_db + 20.0)
        err = (ai.double() - ri.double()).abs()[mask]
        if err.numel():
            worst = max(worst, float(err.max().item()))
            inside += int((err <= 1e-3).sum().item())
            total += int(err.numel())
    return {"against": "float64 STFT of the same samples on the device (pinned to the oracle at <= 2e-5 dB by the parity suite)",
            "bins": total, "bins_rule": "reference value > floor_db + 20 dB (SURVEY.md 8d)", "channels": k,
            "max_abs_err_db": worst, "fraction_within_1e-3_db": (inside / total) if total else None}


def roofline_kernel(cfg, tot, roof):
    """`roofline` is per ABI CALL (a call may be several kernels: ira_rfft_any is four).  This entry is per KERNEL: the
    kernel rocprofv3 ranks first in the committed profile of this configuration, timed live when its call is a single
    launch (then the call's HIP-event time is the kernel's own duration)."""
    dom = dominant_kernel_of_profile(cfg)
    if dom is None:
        return None
    kname, share, src = dom
    call = next((c for c, k in SINGLE_KERNEL_CALLS.items() if k == kname), None)
    live = next((n for n in tot if call and n.startswith(call)), None)
    out = {"kernel": kname, "share_of_device_time_in_profile": share, "profile": f"profiles/{src}"}
    if live is None:
        out["note"] = "not a single-launch call: no live per-kernel duration (see the per-kernel CSV of the profile)"
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
    ap.add_argument("--config", default="report", choices=["report", "2", "3", "4", "5"])
    ap.add_argument("--batch", type=int, default=None, help="IRs (config 5: stereo taps) per GPU per step")
    ap.add_argument("--seconds", type=float, default=None, help="IR length")
    ap.add_argument("--host-batches", type=int, default=4, help="distinct batches rotating in pinned host memory")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=4,
                    help="steps of the serialised (one stream) pass that measures per-kernel durations for the rooflines")
    ap.add_argument("--literal-steps", type=int, default=10,
                    help="config report only: extra steps with the reference's default-on group-delay and diffusion "
                         "blocks added (reported as literal_full_report; 0 = skip)")
    ap.add_argument("--upload", default="copy", choices=["pull", "copy"],
                    help="pull: the batch crosses PCIe under a pull kernel (ira_host_pull); copy: hipMemcpyAsync on the copy engine")
    ap.add_argument("--pull-workgroups", type=int, default=8)
    ap.add_argument("--upload-streams", type=int, default=2,
                    help="float32 / int16 upload as this many pieces on as many copy streams (A/B)")
    ap.add_argument("--variants", default="all", choices=["all", "value"],
                    help="'value' skips the int16 / resident variants (profiling runs)")
    ap.add_argument("--gather", default="final", choices=["final", "step"],
                    help="final: the records of every step stay on their rank and ONE gather to rank 0 closes the timed "
                         "region (north star: a single RCCL gather for the final metrics); step: one gather per step (A/B)")
    ap.add_argument("--spawn-probe", action="store_true",
                    help="launch check without a GPU: every rank joins a gloo group, rank 0 prints the ranks it gathered "
                         "(tests/test_host_cpu.py runs `bench.py --gpus 2 --spawn-probe`)")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus))
    if a.spawn_probe:
        import numpy as np
        from audio_analysis_amd import dist as D
        rank, local_rank, world = D.init_process_group("gloo")
        got = D.gather_metrics(np.full((3, 4), float(rank)), equal_rows=True)
        D.barrier()
        if rank == 0:
            print(json.dumps({"n_gpus": world, "ranks": sorted(set(got[:, 0].tolist())), "rows": int(got.shape[0])}))
        return

    import numpy as np
    import torch

    from audio_analysis_amd import dist as D
    from audio_analysis_amd.engine import Engine
    from audio_analysis_amd.feed import DeviceFeed, HostBatch, run_pipelined
    from audio_analysis_amd.pipeline import METRICS_WIDTH, FullReport
    from audio_analysis_amd.synth import synth_ir

    cfg = config_table()[a.config]
    settings = cfg["settings"]
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

    if a.config == "5":
        return bench_bundle(a, cfg, eng, rank, world, B, n, steps, blocks)

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
                      copy_streams=a.upload_streams)
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
    eng.events = []
    elapsed = timed(lambda c: run_fed(c, host_f32), steps)
    ev_timed = eng.collect_events()
    eng.events = None
    gathered = last.get("g")
    note(f"timed region: {steps} steps in {elapsed:.3f} s = {B * world * steps / elapsed:.0f} IRs/s")

    el_pull = None
    if a.variants == "all" and a.upload == "copy":
        feed.pull = True                               # A/B: the same steps with the pull-kernel upload (ira_host_pull)
        run_fed(2, host_f32)
        el_pull = timed(lambda c: run_fed(c, host_f32), steps)
        feed.pull = False
    # ---- variants: int16 upload; inputs resident in HBM (rotating over the K device-resident batches) ----------------------
    el_i16 = el_res = None
    resident = None
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

    note("variants done" + ("" if el_res is None else f": int16 {B * world * steps / el_i16:.0f}, resident {B * world * steps / el_res:.0f} IRs/s"))
    # ---- per-kernel durations: a short SERIALISED pass (one stream, kernels one at a time) in the same run ---------------
    lanes_used = eng.num_lanes
    eng.num_lanes = 1
    run_fed(K, host_f32)                               # every distinct batch once: plan data (chirp filters) of this stream
    D.barrier(); torch.cuda.synchronize()
    eng.events = []
    run_fed(a.roofline_steps, host_f32)
    D.barrier(); torch.cuda.synchronize()
    ev = eng.collect_events()
    eng.events = None
    eng.num_lanes = lanes_used
    roof_steps = a.roofline_steps

    # ---- second, shorter measurement: the LITERAL default report (group delay + diffusion blocks added) ------------------
    literal = None
    if a.config == "report" and a.literal_steps > 0:
        from dataclasses import replace as _replace
        rep2 = FullReport(eng, _replace(settings, run_group_delay=True, run_diffusion=True))
        run_fed(max(2, K), host_f32, rep2)            # plan pass: every rotating batch once (its lengths' plan data), untimed
        el2 = timed(lambda c: run_fed(c, host_f32, rep2), a.literal_steps)
        literal = {"value": B * world * a.literal_steps / el2, "unit": "IRs/s", "steps": a.literal_steps,
                   "ms_per_step": 1e3 * el2 / a.literal_steps, "blocks": rep2.s.blocks(),
                   "note": "same step (H2D included) plus the reference's default-on group-delay and diffusion blocks "
                           "(SURVEY.md 8f); only the IR waveform plots and PNG rendering remain excluded"}

    flush()
    stft_err = None
    if settings.run_spectrogram and a.variants == "all":      # (profiling runs skip the probe: its extra peak pick and STFT
        probe = feed.push(host_f32[0])                        # launches would be counted as a step by the traffic tools)
        eng.peaks_begin(probe); eng.peaks(probe)
        stft_err = stft_error_distribution(eng, probe, settings)

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
        "upload": "pull kernel (ira_host_pull reads pinned host memory over PCIe)" if feed.pull else "hipMemcpyAsync on a copy stream",
        "variants": {"value": "float32 upload inside the timed region (SURVEY.md 8d)",
                     "value_int16": "PCM16 upload (2 B/sample) + device conversion inside the timed region",
                     "value_resident": "no upload: the same distinct batches already in HBM (compute-only rate)",
                     "value_pull_kernel": "float32 upload by the pull kernel (ira_host_pull) instead of hipMemcpyAsync (A/B)"},
        "h2d_GBps": B * n * 4.0 * steps / elapsed / 1e9,
        "roofline": roof(dominant),
        "roofline_kernel": roofline_kernel(a.config, tot, roof),
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
    note("device side done; cpu baseline next")
    if world == 1 and not a.no_cpu_baseline:
        bm = settings.rt60_bands.band_mode
        out["cpu_baseline"] = cpu_baseline(seconds, blocks, bm, False, cfg["cpu_s"], "IRs/s")
    print(json.dumps(out))


def bench_bundle(a, cfg, eng, rank, world, B, n, steps, blocks):
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
        t0 = time.perf_counter()
        labels, local = run(os.path.join(base, "timed"))
        gathered = D.gather_metrics(local, eng.device)
        D.barrier(); torch.cuda.synchronize()
        elapsed = D.max_over_ranks(time.perf_counter() - t0, eng.device)
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
        "roofline_measured": f"serialised pass ({roof_steps} steps, one stream) in this run",
        "lanes": lanes_used,
        "device_ms_per_step_by_call": {k: v / roof_steps for k, v in sorted(tot.items(), key=lambda kv: -kv[1])},
        "device_ms_per_step": sum(tot.values()) / roof_steps,
    }
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(n / 48000, blocks, settings.rt60_bands.band_mode, True, cfg["cpu_s"],
                                           "stereo taps/s")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
