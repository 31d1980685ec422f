#!/usr/bin/env python3
"""Runs ONE report block over a synthetic batch a few times -- the program rocprofv3 (--kernel-trace --stats / --pmc) wraps
when a single kernel family is studied.   python3 tools/block_probe.py --block modal|decay|bands|bands3rd|spectrum|stft|zplane|gd|diffusion"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from dataclasses import replace
from audio_analysis_amd.engine import Engine
from audio_analysis_amd.synth import synth_ir
from audio_analysis_amd.pipeline import FullReportSettings
from audio_analysis_amd.analyse import decay, rt60bands, modalcloud, frequency_response, spectrogram, zplane
from audio_analysis_amd.analyse import group_delay, diffusion

ap = argparse.ArgumentParser()
ap.add_argument("--block", default="modal")
ap.add_argument("--batch", type=int, default=64); ap.add_argument("--seconds", type=float, default=10.0)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--no-fused-split", action="store_true")
ap.add_argument("--no-band-pairs", action="store_true", help="every band a half-length inverse of its own (A/B)")
ap.add_argument("--no-sparse-bands", action="store_true")
a = ap.parse_args()
eng = Engine("cuda:0")
eng.num_lanes = 1
eng.fuse_half_split = not a.no_fused_split
if a.no_band_pairs:
    eng.pair_real_ffts = False
if a.no_sparse_bands:
    eng.sparse_bands = False
n = int(a.seconds * 48000)
host = np.stack([synth_ir(i, 0, n) for i in range(a.batch)])
b = eng.wrap(eng.to_dev(host.reshape(-1)), np.arange(a.batch, dtype=np.int64) * n, np.full(a.batch, n, np.int64))
eng.peaks(b)
s = FullReportSettings()
run = {
    "modal": lambda: modalcloud.modal_cloud_device(eng, b, 48000, s.modal_cloud),
    "decay": lambda: decay.decay_device(eng, b, 48000, s.decay),
    "bands": lambda: rt60bands.rt60_bands_device(eng, b, 48000, s.rt60_bands),
    "bands3rd": lambda: rt60bands.rt60_bands_device(eng, b, 48000, replace(s.rt60_bands, band_mode="third")),
    "spectrum": lambda: frequency_response.spectrum_device(eng, b, 48000, s.frequency_response, "spectrum", want_phase=True),
    "stft": lambda: spectrogram.spectrogram_device(eng, b, 48000, s.spectrogram, frame_major=True),
    "zplane": lambda: zplane.zplane_device(eng, b, 48000, s.zplane),
    "gd": lambda: group_delay.summary_statistics_device(eng, group_delay.group_delay_device(eng, b, 48000, s.group_delay), 48000, s.group_delay),
    "diffusion": lambda: diffusion.diffusion_device(eng, b, 48000, s.diffusion),
    "peak": lambda: (setattr(b, "peak", None), eng.peaks_begin(b), eng.peaks(b)),
}[a.block]
for _ in range(2):
    run()
torch.cuda.synchronize(); eng.events = []
t0 = time.perf_counter()
for _ in range(a.iters):
    run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
ev = eng.collect_events(); eng.events = None
print(f"block={a.block} B={a.batch} {a.seconds:g}s: {dt*1e3:.3f} ms/iter;", " ".join(f"{k} {sum(v)/a.iters:.3f}" for k, v in sorted(ev.items(), key=lambda kv: -sum(kv[1]))))
