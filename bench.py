#!/usr/bin/env python3
"""
bench.py -- IRs/sec of the metrics-only full report (BASELINE.json metric) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--seconds S]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Steps are software-pipelined (submit k+1 before finishing k) and the independent report blocks of a step run on several
HIP streams (Engine.block_streams); all K steps complete inside the timed region.
One "step" = one pass of the full report (decay + rt60bands[three] + fr + filter + spectrogram + waterfall +
modalcloud + zplane AR(64); PNG rendering, group delay and diffusion excluded -- SURVEY.md section 8d) over
one batch of B synthetic 48 kHz, S-second mono IRs per GPU that is already resident in HBM, ending with the
gather of the per-channel metrics records to rank 0 (RCCL when N > 1).  Weak scaling: B per GPU is fixed.
Rank 0 prints ONE JSON line (contract in the task statement), including
  "roofline"      for the dominant kernel (largest share of device time), measured live with HIP events recorded on the
                  launch stream -- in a short serialised pass of the same steps (one stream) right after the timed region,
                  because a kernel's own duration is not observable while other streams share the GPU with it
                  ("timed_region_ms_per_step_by_call" holds the overlapped event times of the timed region itself),
  "roofline_stft" for the float32 spectrogram STFT kernel (the north-star HBM gate), same method,
  "cpu_baseline"  the oracle (NumPy restatement of the reference) timed on this box's host cores on a bounded
                  sample of the same workload.
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

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
F64_VECTOR_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak (FMA counted as two flops)


# ---------------------------------------------------------------------------------------------------------
# CPU baseline (oracle) -- runs in spawned worker processes that never touch the GPU
# ---------------------------------------------------------------------------------------------------------
def _cpu_one(args):
    index, seconds = args
    import numpy as np  # noqa: F401
    from audio_analysis_amd.synth import synth_ir
    from oracle import ira_oracle as O
    x = synth_ir(index, 0, int(seconds * 48000))
    t0 = time.perf_counter()
    O.analyse_decay(x)
    O.analyse_rt60_bands(x, band_mode="three")
    O.analyse_frequency_response(x)
    O.analyse_filter_response(x)
    O.analyse_spectrogram(x)
    O.analyse_waterfall(x)
    O.analyse_modal_cloud(x)
    O.analyse_zplane(x, ar_order=64)
    return time.perf_counter() - t0


def cpu_baseline(seconds: float, budget_s: float = 20.0):
    import multiprocessing as mp
    for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ[k] = "1"
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        cores = os.cpu_count() or 1
    workers = max(1, min(cores, 16))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(workers) as pool:
        per = pool.map(_cpu_one, [(1000 + i, seconds) for i in range(workers)])
    wall = time.perf_counter() - t0
    return {
        "value": workers / wall, "unit": "IRs/s", "cores": workers, "kind": "port",
        "sample": f"{workers} synthetic {seconds:g} s IRs, one per worker process (single-threaded NumPy oracle, "
                  f"same blocks as the GPU step); mean {sum(per)/len(per):.2f} s per IR per core, wall {wall:.1f} s",
    }


# ---------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="IRs per GPU per step")
    ap.add_argument("--seconds", type=float, default=10.0, help="IR length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--roofline-steps", type=int, default=5,
                    help="steps of the serialised (one stream) pass that measures per-kernel durations for the rooflines")
    ap.add_argument("--literal-steps", type=int, default=5,
                    help="extra steps with the reference's default-on group-delay and diffusion blocks added "
                         "(reported as literal_full_report; 0 = skip)")
    a = ap.parse_args()

    import numpy as np
    import torch

    from audio_analysis_amd import dist as D
    from audio_analysis_amd.engine import Engine
    from audio_analysis_amd.pipeline import METRICS_WIDTH, FullReport, FullReportSettings
    from audio_analysis_amd.synth import synth_ir

    rank, local_rank, world = D.init_process_group()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    eng = Engine(f"cuda:{local_rank}")
    settings = FullReportSettings()
    report = FullReport(eng, settings)

    n = int(a.seconds * 48000)
    first = rank * a.batch
    host = np.stack([synth_ir(first + i, 0, n) for i in range(a.batch)])
    batch = eng.wrap(eng.to_dev(host.reshape(-1)), np.arange(a.batch, dtype=np.int64) * n,
                     np.full(a.batch, n, dtype=np.int64))
    del host

    # Software pipeline over steps: step k+1 is ENQUEUED (FullReport.submit) before step k's results are read back
    # and gathered (FullReport.finish), so the GPU works on k+1 while the host post-processes k.  Every one of the
    # K timed steps is submitted, finished and gathered inside the timed region.
    def run_steps(count, rep=None):
        rep = rep or report
        out, pending = None, None
        for _ in range(count):
            batch.peak = None                  # the peak pick is part of every step
            h = rep.submit(batch)
            if pending is not None:
                out = D.gather_metrics(rep.finish(pending), eng.device, eng.side_stream())
            pending = h
        if pending is not None:
            out = D.gather_metrics(rep.finish(pending), eng.device, eng.side_stream())
        return out

    run_steps(a.warmup)
    D.barrier(); torch.cuda.synchronize()
    eng.events = []
    t0 = time.perf_counter()
    gathered = run_steps(a.steps)
    D.barrier(); torch.cuda.synchronize()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, eng.device)
    ev_timed = eng.collect_events()
    eng.events = None

    # ---- per-kernel durations: a short SERIALISED pass (one stream, kernels one at a time) in the same run ---------------
    # The timed region runs the independent report blocks on several streams (Engine.block_streams), so a kernel's
    # event-to-event time there includes the kernels it shares the GPU with.  A kernel's roofline needs the time it takes
    # when it owns the machine: the same steps are run once more with one lane and HIP events around every call.
    lanes_used = eng.num_lanes
    if lanes_used > 1:
        eng.num_lanes = 1
        run_steps(1)
        D.barrier(); torch.cuda.synchronize()
        eng.events = []
        run_steps(a.roofline_steps)
        D.barrier(); torch.cuda.synchronize()
        ev = eng.collect_events()
        eng.events = None
        eng.num_lanes = lanes_used
        roof_steps = a.roofline_steps
    else:
        ev, roof_steps = ev_timed, a.steps

    # ---- second, shorter measurement: the LITERAL default report (group delay + diffusion blocks added) ------------------
    literal = None
    if a.literal_steps > 0:
        from dataclasses import replace as _replace
        rep2 = FullReport(eng, _replace(settings, run_group_delay=True, run_diffusion=True))
        run_steps(2, rep2)
        D.barrier(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        run_steps(a.literal_steps, rep2)
        D.barrier(); torch.cuda.synchronize()
        el2 = D.max_over_ranks(time.perf_counter() - t1, eng.device)
        literal = {"value": a.batch * world * a.literal_steps / el2, "unit": "IRs/s", "steps": a.literal_steps,
                   "ms_per_step": 1e3 * el2 / a.literal_steps, "blocks": rep2.s.blocks(),
                   "note": "same step plus the reference's default-on group-delay and diffusion blocks (SURVEY.md 8f); "
                           "only the IR waveform plots and PNG rendering remain excluded"}

    if rank != 0:
        return
    assert gathered is not None and gathered.shape == (a.batch * world, METRICS_WIDTH)
    total_irs = a.batch * world * a.steps
    # ---- per-call device time over the timed region -> dominant kernel + rooflines ---------------------------
    tot = {k: sum(v) for k, v in ev.items()}
    dev_ms = sum(tot.values())
    dominant = max(tot, key=tot.get)
    peaks = np.asarray(batch.peak)
    L = (n - peaks).astype(np.float64)

    def stft_bytes(nfft, hop):
        frames = 1 + (L - nfft) // hop
        return float(np.sum(4.0 * L + 4.0 * (nfft // 2 + 1) * frames))

    # HBM traffic per launch from the committed PMC profile (profiles/r01_traffic.json: FETCH_SIZE / WRITE_SIZE
    # collected in separate rocprofv3 --pmc passes of this same command, gfx950 correction applied there)
    try:
        traffic_tab = json.load(open(os.path.join(REPO, "profiles", "r01_traffic.json")))["calls"]
    except Exception:
        traffic_tab = {}

    def traffic_of(name):
        t = traffic_tab.get(name)
        return None if t is None else t["hbm_bytes_per_channel"] * a.batch

    def roof(name):
        """Roofline of one ABI call over the timed region.  Bytes (or flops) and time are both PER STEP: a call that is
        launched twice in a step (ira_rfft_any, ira_edc_db) is charged the sum of its launches."""
        step_ms = tot[name] / roof_steps
        launches = len(ev[name]) / roof_steps
        nb3 = 3 if settings.rt60_bands.band_mode == "three" else None
        if name.startswith("ira_stft_mag_db") and "[f32" in name:
            b = stft_bytes(settings.spectrogram.n_fft, settings.spectrogram.hop_length)
            what = "4L in + 4*F*T out bytes per channel"
        elif name.startswith("ira_stft_mag_db") and ("[f64,n%d]" % settings.modal_cloud.n_fft) in name:
            b = stft_bytes(settings.modal_cloud.n_fft, settings.modal_cloud.hop_length)
            what = "4L in + 4*F*T out bytes per channel"
        elif name.startswith("ira_stft_logbin"):
            nf = settings.modal_cloud.n_fft
            frames = 1 + (L - nf) // settings.modal_cloud.hop_length
            b = float(np.sum(4.0 * L + 4.0 * 240 * frames))
            what = "4L in + 4*nbins*T out bytes per channel (fused STFT + log-bin aggregation; the dB matrix is never written)"
        elif name.startswith("ira_rfft_any"):
            # compulsory traffic: samples in + half spectra out of the windowed fr/filter transform (arbitrary length ->
            # Bluestein).  The three float64 passes over M = 2^20 move ~15x that through L2/MALL/HBM (see "traffic");
            # that working-set traffic is what bounds these kernels.
            b = float(np.sum(4.0 * L + 16.0 * (L // 2 + 1)))
            what = "per channel: 4L + 16(L/2+1) bytes (fr/filter spectrum, Bluestein)"
        elif name.startswith("ira_rfft_smooth"):
            b = float(a.batch) * (4.0 * n + 16.0 * (n // 2 + 1))
            what = "per channel: 4n + 16(n/2+1) bytes (RT60 full-file forward transform, direct mixed radix, paired)"
        elif name.startswith("ira_band_irfft") and nb3:
            b = float(a.batch) * (16.0 * (n // 2 + 1) + nb3 * 4.0 * n)
            what = f"per channel: 16(n/2+1) spectrum in + {nb3} band signals x 4n out bytes"
        elif name.startswith("ira_edc_db") and nb3:
            b = float(np.sum(8.0 * L)) * (1 + nb3)
            what = f"4L in + 4L out bytes per segment; 1 decay + {nb3} band segments per channel"
        elif name.startswith("ira_ar_gram"):
            b = float(np.sum(4.0 * L))
            what = "4L bytes per channel (samples read once; p+1 lag sums)"
        else:
            return {"kernel": name, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None, "traffic": None, "avg_launch_ms": step_ms / launches, "algorithmic": "not modelled"}
        ach = b / (step_ms * 1e-3) / 1e9
        tr = traffic_of(name)
        out = {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
               "frac": ach / HBM_PEAK_GBS, "traffic": tr, "algorithmic_bytes": b, "launches_per_step": launches,
               "avg_launch_ms": step_ms / launches, "ms_per_step": step_ms, "algorithmic": what,
               # measured L2<->fabric traffic of the call divided by its time (what the memory system actually moved)
               "traffic_GBps": None if tr is None else tr / (step_ms * 1e-3) / 1e9}
        # Whole-file float64 transforms do not fit on a CU: SURVEY.md 8(d) asks for the two-pass STREAMING model beside the
        # compulsory bytes (every pass reads and writes its n complex values once), and they are vector-float64 work.
        if name.startswith("ira_band_irfft_smooth") and nb3:
            jobs = a.batch * nb3 / 2.0                                   # two bands ride one complex inverse
            stream_b = jobs * (2 * 16.0 * n + 16.0 * n + 16.0 * n + 2 * 4.0 * n)   # spectra (both halves) in, work out/in, 2 bands out
            flops = jobs * 5.0 * n * np.log2(n)
        elif name.startswith("ira_rfft_smooth"):
            jobs = a.batch / 2.0                                         # two channels ride one complex transform
            stream_b = jobs * (2 * 4.0 * n + 16.0 * n + 16.0 * n + 16.0 * n + 16.0 * n + 2 * 16.0 * (n // 2 + 1))
            flops = jobs * 5.0 * n * np.log2(n)
        elif name.startswith("ira_rfft_any"):
            # Bluestein over M = 2^m >= 2*len - 1; even L runs as a complex transform of L/2 (half the convolution size).
            tlen = np.where(L % 2 == 0, L / 2, L)
            M = 2.0 ** np.ceil(np.log2(2 * tlen - 1))
            stream_b = float(np.sum(4.0 * L + 16.0 * M * 5 + 16.0 * (L // 2 + 1)))     # K1 w, K2 r + filter r + w, K3 r
            flops = float(np.sum(2 * 5.0 * M * np.log2(M) + 6.0 * M))
        else:
            stream_b = flops = None
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

    stft_name = next((k for k in tot if k.startswith("ira_stft_mag_db") and "[f32" in k), None)
    out = {
        "metric": "IRs/sec full report (STFT+RT60bands+zplane), 48 kHz 10 s IR",
        "value": total_irs / elapsed,
        "unit": "IRs/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": 1e3 * elapsed / a.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"metrics-only full report on {a.batch} mono synthetic IRs per GPU, {a.seconds:g} s @ 48 kHz "
                        f"(SURVEY.md 8d generator), inputs resident in HBM",
            "blocks": settings.blocks(),
            "excluded": ["png rendering", "group delay", "diffusion", "ir plots"],
            "batch_per_gpu": a.batch, "ir_seconds": a.seconds, "parallelism": f"file-sharded dp{world}",
            "arithmetic": "f64 (EDC scan, long FFTs, modal/waterfall STFT, AR Gram/solve/roots); "
                          "f32 butterflies for the spectrogram STFT",
        },
        "roofline": roof(dominant),
        "roofline_stft": roof(stft_name) if stft_name else None,
        "roofline_measured": (f"serialised pass of {roof_steps} steps in this run (one stream, kernels one at a time); the "
                              f"timed region deals the report blocks onto {lanes_used} streams" if lanes_used > 1 else
                              "timed region (one stream)"),
        "lanes": lanes_used,
        "device_ms_per_step_by_call": {k: v / roof_steps for k, v in sorted(tot.items(), key=lambda kv: -kv[1])},
        "device_ms_per_step": dev_ms / roof_steps,
        "timed_region_ms_per_step_by_call": {k: sum(v) / a.steps for k, v in
                                             sorted(ev_timed.items(), key=lambda kv: -sum(kv[1]))},
        "literal_full_report": literal,
    }
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.seconds)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
