#!/usr/bin/env python3
"""Static instruction budget of one kernel from hipcc's assembly (hipcc --save-temps -> *.s): counts per class over the kernel's
straight-line body (all our register-FFT kernels are fully unrolled: no loops, so static = dynamic per wave for the
frame-major variants).   python3 tools/isa_budget.py file.s <kernel name substring>"""
import collections, re, sys

path, pat = sys.argv[1], sys.argv[2]
text = open(path).read()
# kernels start at "<mangled>:" after ".globl"; end at ".Lfunc_end"
kernels = re.findall(r"^(\S*%s\S*):[^\n]*\n(.*?)^\.Lfunc_end" % re.escape(pat), text, flags=re.S | re.M)
for name, body in kernels:
    cls = collections.Counter()
    ops = collections.Counter()
    for line in body.splitlines():
        line = line.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        op = line.split()[0]
        ops[op] += 1
        if op.startswith("v_pk_"): c = "VALU packed f32"
        elif op.startswith(("v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_mad")): c = "VALU f32 arithmetic"
        elif op.startswith(("v_fma_f64", "v_mul_f64", "v_add_f64", "v_fmac_f64")): c = "VALU f64 arithmetic"
        elif op.startswith(("v_log", "v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")): c = "VALU transcendental"
        elif op.startswith(("v_cmp", "v_cndmask", "v_max", "v_min")): c = "VALU compare/select/max"
        elif op.startswith(("v_mov", "v_accvgpr", "v_readlane", "v_readfirstlane", "v_writelane", "v_perm", "v_bfi", "v_swap")): c = "VALU moves"
        elif op.startswith("v_"): c = "VALU other (int/addr/cvt)"
        elif op.startswith("ds_"): c = "LDS"
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c = "VMEM"
        elif op.startswith("s_waitcnt"): c = "s_waitcnt"
        elif op.startswith("s_barrier"): c = "s_barrier"
        elif op.startswith("s_"): c = "SALU/other scalar"
        else: c = "other"
        cls[c] += 1
    valu = sum(v for k, v in cls.items() if k.startswith("VALU"))
    print(f"== {name}: {sum(cls.values())} instructions, {valu} VALU")
    for k, v in sorted(cls.items(), key=lambda kv: -kv[1]):
        print(f"   {k:30s} {v:6d}")
    print("   top opcodes:", ", ".join(f"{k} {v}" for k, v in ops.most_common(14)))
