"""How many small host->device copies does one pipeline step enqueue?  (GPU box)  python tools/count_copies.py"""
import collections, sys, time, traceback
import numpy as np
sys.path.insert(0, ".")
from audio_analysis_amd.engine import Engine
from audio_analysis_amd.pipeline import FullReport
from audio_analysis_amd.synth import synth_ir

eng = Engine("cuda:0")
n = 480000
host = np.stack([synth_ir(i, 0, n) for i in range(64)])
batch = eng.wrap(eng.to_dev(host.reshape(-1)), np.arange(64, dtype=np.int64) * n, np.full(64, n, dtype=np.int64))
rep = FullReport(eng)
rep.run(batch); rep.run(batch)
calls = collections.Counter(); sizes = collections.Counter()
orig = eng.to_dev
def counted(a):
    fr = traceback.extract_stack(limit=3)[0]
    calls[f"{fr.name}"] += 1; sizes[f"{fr.name}"] += np.asarray(a).nbytes
    return orig(a)
eng.to_dev = counted
batch.peak = None
t0 = time.perf_counter(); h = rep.submit(batch); t1 = time.perf_counter(); rep.finish(h); t2 = time.perf_counter()
print("submit %.2f ms finish %.2f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
print("to_dev calls per step:", sum(calls.values()), "bytes", sum(sizes.values()))
for k, v in calls.most_common(): print(f"  {k:28s} {v:4d} copies {sizes[k]:9d} B")
