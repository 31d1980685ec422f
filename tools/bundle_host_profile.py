#!/usr/bin/env python3
"""cProfile of the batched bundle path's host side (BASELINE config 5 shape: stereo 5 s PCM16 taps, 128 per step):
where does the wall time of run_bundle_metrics go when the GPU needs ~12 ms per step?
    python3 tools/bundle_host_profile.py [--taps 512] [--per-step 128] [--root /dev/shm/ira_prof]"""
import argparse, cProfile, json, os, pstats, shutil, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.bundle_rate import write_tap

ap = argparse.ArgumentParser()
ap.add_argument("--taps", type=int, default=512)
ap.add_argument("--per-step", type=int, default=128)
ap.add_argument("--seconds", type=float, default=5.0)
ap.add_argument("--root", default="/dev/shm/ira_prof")
a = ap.parse_args()
from audio_analysis_amd.synth import synth_ir
n = int(a.seconds * 48000)
os.makedirs(os.path.join(a.root, "taps"), exist_ok=True)
names = [f"tap{i:05d}" for i in range(a.taps)]
base = [np.stack([synth_ir(i, 0, n), synth_ir(i, 1, n)], axis=1) for i in range(32)]
for i, name in enumerate(names):
    write_tap(os.path.join(a.root, "taps", name + ".wav"), base[i % 32])
json.dump({"sample_rate_hz": 48000, "length_samples": n, "taps": names}, open(os.path.join(a.root, "meta.json"), "w"))
import torch
from audio_analysis_amd.analyse import bundle
try:
    bundle.run_bundle_metrics(a.root, taps_per_step=a.per_step)
    torch.cuda.synchronize()
    pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
    labels, rec = bundle.run_bundle_metrics(a.root, taps_per_step=a.per_step)
    torch.cuda.synchronize(); pr.disable()
    dt = time.perf_counter() - t0
    print(f"{a.taps} taps in {dt*1e3:.1f} ms = {a.taps/dt:.0f} taps/s, {1e3*dt/(a.taps/a.per_step):.2f} ms per step of {a.per_step}")
    pstats.Stats(pr).sort_stats("cumulative").print_stats(16); pstats.Stats(pr).sort_stats("tottime").print_stats(30)
finally:
    shutil.rmtree(a.root, ignore_errors=True)
