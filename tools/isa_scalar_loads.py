#!/usr/bin/env python3
"""Scalar loads (s_load_*, s_buffer_load_*) that sit inside a loop of a kernel's assembly (hipcc -S --cuda-device-only):
   python3 tools/isa_scalar_loads.py file.s [kernel substring]
A kernel-argument struct field indexed by a loop counter (P.radix[pass], J.off[e] ...) is re-loaded from the kernel-argument
segment every iteration, each load followed by s_waitcnt lgkmcnt(0): ~200 cycles of serial latency per load on the critical
path (and lgkmcnt also drains the LDS queue).  A loop = a label that a LATER branch jumps back to."""
import re, sys
text = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(\S+):[^\n]*\n(.*?)^\.Lfunc_end", text, flags=re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name or not name.startswith("_Z"):
        continue
    lines = [l.split(";")[0].strip() for l in body.splitlines()]
    lines = [l for l in lines if l and not l.startswith(".") or re.match(r"^\.LBB\S*:$", l or "")]
    label_at = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    loops = []
    for i, l in enumerate(lines):
        b = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", l)
        if b:
            tgt = b.group(1) or b.group(2)
            if tgt in label_at and label_at[tgt] < i:
                loops.append((label_at[tgt], i))
    sl = [i for i, l in enumerate(lines) if l.startswith(("s_load", "s_buffer_load"))]
    inside = [i for i in sl if any(a <= i <= b for a, b in loops)]
    total = len([l for l in lines if not l.endswith(":")])
    print(f"{name[:90]:90s} instr {total:6d}  scalar loads {len(sl):3d}  inside loops {len(inside):3d}  loops {len(loops):3d}")
