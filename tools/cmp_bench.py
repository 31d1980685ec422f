#!/usr/bin/env python3
"""Side by side: per-call device times (serialised pass) of two or more bench lines.  usage: cmp_bench.py a.json b.json ..."""
import json, sys
lines = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:]]
print("value            ", *[f"{d['value']:10.0f}" for d in lines])
print("resident         ", *[f"{(d.get('value_resident') or 0):10.0f}" for d in lines])
print("device_ms_per_step", *[f"{d['device_ms_per_step']:10.3f}" for d in lines])
keys = []
for d in lines:
    for k in d["device_ms_per_step_by_call"]:
        if k not in keys:
            keys.append(k)
for k in keys:
    print(f"{k:36s}", *[f"{d['device_ms_per_step_by_call'].get(k, float('nan')):10.3f}" for d in lines])
for d in lines:
    l = d.get("literal_full_report")
    if l:
        print("literal", round(l["value"]), l.get("added_blocks_device_ms"))
    if d.get("roofline_stft"):
        print("stft frac", d["roofline_stft"]["frac"])
