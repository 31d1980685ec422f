#!/usr/bin/env python3
"""VALU instructions per loop iteration of every variant of tools/micro/dft16_rate.hip, from the device assembly:

    hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I audio_analysis_amd/csrc --cuda-device-only -S \
        tools/micro/dft16_rate.hip -o /tmp/dft16_rate.s
    python3 tools/micro/dft16_count.py /tmp/dft16_rate.s > counts.txt        (one number per variant, in order)

The timed loop of each kernel is its only loop (`#pragma unroll 1`): the body is the text between the label a backward
s_cbranch targets and that branch.  Prints the class mix to stderr."""
import collections, re, sys

text = open(sys.argv[1]).read()
out = []
NVAR = int(sys.argv[2]) if len(sys.argv) > 2 else 5
for var in range(NVAR):
    m = re.search(r"^(_Z1kILi%dE[^:\n]*):[^\n]*\n(.*?)^\.Lfunc_end" % var, text, flags=re.S | re.M)
    if not m:
        out.append(0); continue
    lines = m.group(2).splitlines()
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    spills = sum(1 for l in lines if "scratch_" in l)
    if spills:
        print(f"variant {var}: {spills} scratch instructions (spills) -- counts are not the kernel's", file=sys.stderr)
    best = None
    for i, l in enumerate(lines):
        mm = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            span = (labels[mm.group(1)], i)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    cls = collections.Counter()
    if best:
        for l in lines[best[0]:best[1] + 1]:
            l = l.split(";")[0].strip()
            if not l or l.endswith(":") or l.startswith("."):
                continue
            op = l.split()[0]
            if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_pk_")): c = "valu f32 arith"
            elif op.startswith(("v_log", "v_exp", "v_rcp", "v_rsq", "v_sqrt")): c = "valu transcendental"
            elif op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_swap")): c = "valu move"
            elif op.startswith(("v_cmp", "v_cndmask", "v_max", "v_min")): c = "valu cmp/select"
            elif op.startswith("v_"): c = "valu other"
            elif op.startswith("ds_"): c = "lds"
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "vmem"
            elif op.startswith("s_waitcnt"): c = "s_waitcnt"
            else: c = "scalar"
            cls[c] += 1
    valu = sum(v for k, v in cls.items() if k.startswith("valu"))
    print(f"variant {var}: VALU {valu}  " + "  ".join(f"{k} {v}" for k, v in sorted(cls.items())), file=sys.stderr)
    out.append(valu)
print(" ".join(str(v) for v in out))
