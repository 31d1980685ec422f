#!/usr/bin/env python3
"""cProfile of the host side of FullReport.run (where does the non-kernel time of a step go?)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from audio_analysis_amd.engine import Engine
from audio_analysis_amd.pipeline import FullReport
from audio_analysis_amd.synth import synth_ir
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine("cuda:0"); n = 480000
host = np.stack([synth_ir(i, 0, n) for i in range(B)])
batch = eng.wrap(eng.to_dev(host.reshape(-1)), np.arange(B, dtype=np.int64) * n, np.full(B, n, np.int64))
rep = FullReport(eng)
for _ in range(2):
    batch.peak = None; rep.run(batch)
torch.cuda.synchronize()
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
for _ in range(3):
    batch.peak = None; rep.run(batch)
torch.cuda.synchronize(); pr.disable()
print(f"wall per step {1e3*(time.perf_counter()-t0)/3:.2f} ms")
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

# ---- host cost of the two halves of a step with the GPU out of the way -------------------------------------
ts, tf = [], []
for _ in range(5):
    torch.cuda.synchronize(); batch.peak = None
    t0 = time.perf_counter(); h = rep.submit(batch); t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter(); rep.finish(h); t3 = time.perf_counter()
    ts.append(1e3 * (t1 - t0)); tf.append(1e3 * (t3 - t2))
print(f"host: submit {np.median(ts):.2f} ms (enqueue only, includes the peak round trip), finish {np.median(tf):.2f} ms (results already on the host)")
