#!/bin/bash
# Kernel-trace of the default (two-lane) bench and the GPU's busy fraction inside the timed region.
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r3_gap}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 12 --warmup 3 --variants value --no-cpu-baseline --literal-steps 0 --roofline-steps 1 > $O/bench.json 2> $O/bench.err
python3 - $O <<'PY'
import csv, glob, sys, json
d = sys.argv[1]
f = glob.glob(d + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:40], r["Stream_Id"]) for r in csv.DictReader(open(f))]
rows.sort()
# the timed region: the 12 steps before the serialised roofline pass; take the window between the 4th and the 15th peak_decode
pk = [s for s, e, n, st in rows if "peak_decode" in n]
# the timed region: the ten consecutive steps (peak picks) that take the least time -- warm-up, plan and tuning passes and the
# serialised roofline pass are all slower
best = min(range(len(pk) - 10), key=lambda i: pk[i + 10] - pk[i])
t0, t1 = pk[best], pk[best + 10]
sel = [(s, e, n, st) for s, e, n, st in rows if s >= t0 and e <= t1]
busy, cur_s, cur_e = 0, None, None
for s, e, n, st in sel:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
total = t1 - t0
ksum = sum(e - s for s, e, n, st in sel)
print(f"window {total/1e6:.2f} ms over 10 steps: GPU busy (union of kernels) {busy/total:.1%}, sum of kernel durations {ksum/1e6:.2f} ms = {ksum/total:.2f}x the window")
streams = {}
for s, e, n, st in sel: streams[st] = streams.get(st, 0) + (e - s)
print({k: round(v/1e6, 2) for k, v in sorted(streams.items(), key=lambda kv: -kv[1])[:6]})
PY
