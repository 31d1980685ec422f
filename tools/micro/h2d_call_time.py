#!/usr/bin/env python3
"""How long does the HOST spend inside an 'asynchronous' H2D copy of a pinned buffer?  (The bundle path uploads 123 MB of PCM16
per step with tensor.to(device, non_blocking=True).)"""
import time, torch
dev = torch.device("cuda:0")
n = 61_440_000                      # int16 samples of 128 stereo 5 s taps
src = torch.empty(n, dtype=torch.int16, pin_memory=True); src.zero_()
dst = torch.empty(n, dtype=torch.int16, device=dev)
copy_stream = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
def timed(label, fn, reps=5):
    best = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        best.append((t1 - t0, t2 - t0))
    print(f"{label:60s} host call {min(b[0] for b in best)*1e3:7.3f} ms   until done {min(b[1] for b in best)*1e3:7.3f} ms")
timed("src.to(dev, non_blocking=True)  (allocates the destination)", lambda: src.to(dev, non_blocking=True))
timed("dst.copy_(src, non_blocking=True), current stream", lambda: dst.copy_(src, non_blocking=True))
def on_side():
    with torch.cuda.stream(copy_stream):
        dst.copy_(src, non_blocking=True)
timed("dst.copy_(src, non_blocking=True), side stream", on_side)
src2 = torch.empty(n, dtype=torch.int16); src2.zero_()
timed("pageable source, dst.copy_(src2, non_blocking=True)", lambda: dst.copy_(src2, non_blocking=True))
import threading
box = {}
def alloc():
    box["p"] = torch.empty(n, dtype=torch.int16, pin_memory=True); box["p"].zero_()
th = threading.Thread(target=alloc); th.start(); th.join()
timed("pinned in ANOTHER thread, .to(dev, non_blocking=True)", lambda: box["p"].to(dev, non_blocking=True))
