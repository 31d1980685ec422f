#!/usr/bin/env python3
"""A/B of the batch upload inside the full-report step: pull kernel (ira_host_pull) vs copy engine (hipMemcpyAsync), alternated
in ONE process so that warm-up order cannot decide.   python3 tools/upload_ab.py [--batch 256] [--steps 20] [--rounds 3]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from concurrent.futures import ThreadPoolExecutor
from audio_analysis_amd.engine import Engine
from audio_analysis_amd.feed import DeviceFeed, HostBatch, run_pipelined
from audio_analysis_amd.pipeline import FullReport
from audio_analysis_amd.synth import synth_ir

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256); ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3); ap.add_argument("--workgroups", default="8")
a = ap.parse_args()
eng = Engine("cuda:0")
rep = FullReport(eng)
n, K, B = 480000, 4, a.batch
with ThreadPoolExecutor(16) as ex:
    chans = list(ex.map(lambda i: synth_ir(i, 0, n), range(K * B)))
host = [HostBatch(eng, np.stack(chans[k * B:(k + 1) * B])) for k in range(K)]
del chans
feed = DeviceFeed(eng, B * n, depth=4)

def run(count):
    run_pipelined(rep, feed, (host[i % K] for i in range(count)))

run(K + 3)
modes = [("copy", 0)] + [("pull", int(w)) for w in a.workgroups.split(",")]
for r in range(a.rounds):
    for name, wg in (modes if r % 2 == 0 else modes[::-1]):
        feed.pull = name == "pull"
        feed.pull_workgroups = wg or 8
        run(2)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        run(a.steps)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"round {r} {name:4s} wg={wg:2d}: {B * a.steps / dt:8.0f} IRs/s  {1e3 * dt / a.steps:6.2f} ms/step", flush=True)
