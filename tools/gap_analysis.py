"""Idle GPU time between consecutive operations (kernels + copies) of the bench's timed steps, from a rocprofv3
--kernel-trace --memory-copy-trace CSV dump.  Prints the busy/idle split and the operations that follow the largest
total idle time."""
import collections, csv, glob, sys

root = sys.argv[1]
ops = []
for f in glob.glob(f"{root}/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:70]))
for f in glob.glob(f"{root}/trace/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ops.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Kind", "?"))))
ops.sort()
if not ops:
    sys.exit("no trace rows found")
# steady state: the last 60 % of the timeline
t_lo = ops[0][0] + 0.4 * (ops[-1][1] - ops[0][0])
ops = [o for o in ops if o[0] >= t_lo]
span = ops[-1][1] - ops[0][0]
busy_end = ops[0][1]
busy = ops[0][1] - ops[0][0]
idle_after = collections.Counter(); idle_cnt = collections.Counter()
for (s, e, name), prev in zip(ops[1:], ops[:-1]):
    if s > busy_end:
        gap = s - busy_end
        idle_after[name] += gap; idle_cnt[name] += 1
        busy += e - s
    else:
        busy += max(0, e - busy_end)
    busy_end = max(busy_end, e)
print(f"window {span/1e6:.2f} ms: busy {busy/1e6:.2f} ms ({100*busy/span:.1f} %), idle {(span-busy)/1e6:.2f} ms; {len(ops)} operations")
kinds = collections.Counter(); ktime = collections.Counter()
for s, e, name in ops:
    kinds[name] += 1; ktime[name] += e - s
print("\nidle time charged to the operation that FOLLOWS the gap:")
for name, t in idle_after.most_common(25):
    print(f"  {t/1e6:8.3f} ms  {idle_cnt[name]:5d} gaps  avg {t/idle_cnt[name]/1e3:7.1f} us  {name}")
print("\ncopies:")
for name in kinds:
    if name.startswith("COPY"):
        print(f"  {name}: {kinds[name]} ops, {ktime[name]/1e6:.3f} ms total, avg {ktime[name]/kinds[name]/1e3:.1f} us")
