#!/usr/bin/env python3
"""Device time of the section-8f blocks (group delay, diffusion) on the bench batch: B mono 10 s IRs."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audio_analysis_amd.engine import Engine
from audio_analysis_amd.analyse import group_delay as gdm, diffusion as dm
from audio_analysis_amd.synth import synth_ir
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine("cuda:0"); n = 480000
host = np.stack([synth_ir(i, 0, n) for i in range(B)])
batch = eng.wrap(eng.to_dev(host.reshape(-1)), np.arange(B, dtype=np.int64) * n, np.full(B, n, np.int64))
eng.peaks(batch)
def gd_with_stats():
    st = gdm.GroupDelayAnalysisSettings()
    dev = gdm.group_delay_device(eng, batch, 48000, st)
    return gdm.summary_statistics_device(eng, dev, 48000, st)


for name, fn in (("group delay", lambda: gdm.group_delay_device(eng, batch, 48000, gdm.GroupDelayAnalysisSettings())),
                 ("group delay + device quantiles", gd_with_stats),
                 ("diffusion (report defaults: hop 50 ms, lag 5 ms)", lambda: dm.diffusion_device(eng, batch, 48000, dm.DiffusionAnalysisSettings(hop_seconds=0.05, max_lag_milliseconds=5.0))),
                 ("diffusion (module defaults: hop 10 ms, lag 10 ms)", lambda: dm.diffusion_device(eng, batch, 48000, dm.DiffusionAnalysisSettings()))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); eng.events = []
    for _ in range(3):
        fn()
    ev = eng.collect_events(); eng.events = None
    tot = {k: sum(v) / 3 for k, v in ev.items()}
    print(f"{name}: {sum(tot.values()):.3f} ms per {B} channels  " + ", ".join(f"{k} {v:.3f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])))
