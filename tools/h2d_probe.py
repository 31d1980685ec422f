#!/usr/bin/env python3
"""Host-to-device copy rate of pinned memory on this box: by NUMA node of the pinned allocation (first touch from a CPU of
that node), alone and split over several streams.   python3 tools/h2d_probe.py [MB]"""
import glob, os, sys, time
import torch

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 492
def read(p):
    try:
        return open(p).read().strip()
    except OSError as e:
        return f"<{e.__class__.__name__}>"
print("gpu numa nodes:", {p: read(p) for p in glob.glob("/sys/class/drm/card*/device/numa_node")})
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
print("nodes:", [(os.path.basename(n), read(n + "/cpulist")) for n in nodes])
print("affinity now:", len(os.sched_getaffinity(0)), "cpus")
def parse(cpulist):
    out = []
    for part in cpulist.split(","):
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out
dev = torch.device("cuda:0")
dst = torch.empty(mb << 20, dtype=torch.uint8, device=dev)
allowed = os.sched_getaffinity(0)
def rate(src, streams=1, reps=5):
    ss = [torch.cuda.Stream() for _ in range(streams)]
    n = src.numel(); step = (n + streams - 1) // streams
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        for i, s in enumerate(ss):
            with torch.cuda.stream(s):
                dst[i * step:(i + 1) * step].copy_(src[i * step:(i + 1) * step], non_blocking=True)
        torch.cuda.synchronize()
        best = max(best, n / (time.perf_counter() - t0) / 1e9)
    return best
for n in nodes:
    cpus = set(parse(read(n + "/cpulist"))) & allowed
    if not cpus:
        print(os.path.basename(n), "no cpus allowed"); continue
    os.sched_setaffinity(0, cpus)
    src = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    src.fill_(1)                                                   # first touch from this node
    os.sched_setaffinity(0, allowed)
    print(f"{os.path.basename(n)}: pinned {mb} MB -> H2D {rate(src):.1f} GB/s (1 stream), {rate(src, 2):.1f} (2), {rate(src, 4):.1f} (4)", flush=True)
    del src
