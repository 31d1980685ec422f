#!/usr/bin/env python3
"""Instruction counts of one kernel between consecutive s_barrier instructions (hipcc -S --cuda-device-only assembly):
   python3 tools/isa_segments.py file.s <kernel name substring>
A straight-line (fully unrolled) kernel executes every segment once per wave, so the table shows where a wave's VALU
instructions go; the compiler moves code across barriers, so a segment is a neighbourhood, not a source region."""
import re, sys
t = open(sys.argv[1]).read()
m = re.search(r"^(\S*%s\S*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(sys.argv[2]), t, flags=re.S | re.M)
seg = []
keys = ("valu", "f64", "trans", "cvt", "mov", "int_cmp", "lds", "vmem", "salu")
cur = dict.fromkeys(keys, 0)
for line in m.group(2).splitlines():
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        continue
    op = line.split()[0]
    if op.startswith("s_barrier"):
        seg.append(cur); cur = dict.fromkeys(keys, 0); continue
    if op.startswith("v_"):
        cur["valu"] += 1
        if op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64")): cur["f64"] += 1
        elif op.startswith(("v_rsq", "v_rcp", "v_log", "v_exp", "v_sqrt")): cur["trans"] += 1
        elif op.startswith("v_cvt"): cur["cvt"] += 1
        elif op.startswith(("v_mov", "v_readlane", "v_readfirstlane")): cur["mov"] += 1
        else: cur["int_cmp"] += 1
    elif op.startswith("ds_"): cur["lds"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): cur["vmem"] += 1
    elif op.startswith("s_"): cur["salu"] += 1
seg.append(cur)
print(f"# {m.group(1)}: {len(seg)} segments between barriers")
print("seg  " + "  ".join(f"{k:>7s}" for k in keys))
for i, c in enumerate(seg):
    print(f"{i:3d}  " + "  ".join(f"{c[k]:7d}" for k in keys))
print("all  " + "  ".join(f"{sum(c[k] for c in seg):7d}" for k in keys))
