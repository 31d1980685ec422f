#!/usr/bin/env python3
"""Static check of the gfx950 ISA of every kernel for SERIALISED global loads: a global_load whose result is waited for
(s_waitcnt vmcnt(0)) within a few instructions, before the next load is issued -- each such pair is a full memory round trip
on the critical path.  Typical causes (both found in round 2): per-job values written as J.off[e] inside a loop (vector load +
wait per mention), and loads the optimiser sank into the branch that guards their only use.
    python3 tools/isa_serial_loads.py [file.hip ...]      (default: every csrc/*.hip)"""
import re, subprocess, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "audio_analysis_amd" / "csrc"
files = [Path(a) for a in sys.argv[1:]] or sorted(CSRC.glob("*.hip"))
for f in files:
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT/'include'}",
                        f"-I{CSRC}", "-S", "--cuda-device-only", "-o", tmp.name, str(f)], check=True,
                       stderr=subprocess.DEVNULL)
        lines = Path(tmp.name).read_text().splitlines()
    kern, body = None, []
    def report():
        if kern is None:
            return
        ins = [l.strip() for l in body if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        loads = [i for i, l in enumerate(ins) if l.startswith(("global_load", "buffer_load", "flat_load"))]
        serial = 0
        for i in loads:
            for j in range(i + 1, min(i + 4, len(ins))):
                if ins[j].startswith(("global_load", "buffer_load", "flat_load")):
                    break
                if ins[j].startswith("s_waitcnt") and "vmcnt(0)" in ins[j]:
                    serial += 1
                    break
        name = subprocess.run(["c++filt", kern], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)[:70]
        print(f"{f.name:18s} {name:70s} loads {len(loads):4d}  load->wait(0) pairs {serial:3d}")
    for l in lines:
        m = re.match(r"^(_Z\w+):\s", l)
        if m and ("kernel" in m.group(1)):
            report()
            kern, body = m.group(1), []
        elif l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            report()
            kern, body = None, []
        elif kern:
            body.append(l)
