"""
Build libira.so (hand-written HIP kernels + C-ABI) in-tree for gfx950 with hipcc.

    python -m audio_analysis_amd.build            # incremental
    python -m audio_analysis_amd.build --force

hipcc cross-compiles without a GPU.  The shared object lands next to the sources
(audio_analysis_amd/csrc/libira.so) so it travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

CSRC = Path(__file__).resolve().parent / "csrc"
LIB = CSRC / "libira.so"
ARCH = "gfx950"

# per-file extra flags; -ffp-contract=off where f64 arithmetic must round like NumPy's (no FMA fusion)
SOURCES = {
    "ira_api.hip": [],
    "ira_edc.hip": ["-ffp-contract=off"],
    # -fno-slp-vectorize: hipcc otherwise packs the complex arithmetic into v_pk_{add,mul,fma}_f32.  Measured with
    # tools/pk_f32_rate.hip on MI355X: a packed op occupies the SIMD twice as long as a scalar one (0.87 vs 1.5
    # wave-instructions per CU-cycle at saturation) and the packing itself costs register moves.
    "ira_stft.hip": ["-fno-slp-vectorize"],
    "ira_stft2.hip": ["-fno-slp-vectorize"],
    "ira_stft3.hip": ["-fno-slp-vectorize"],
    "ira_stft4.hip": [],
    "ira_fftlong.hip": [],
    "ira_fftsmooth.hip": [],
    "ira_spectrum.hip": ["-ffp-contract=off"],
    "ira_modal.hip": ["-ffp-contract=off"],
    "ira_ar.hip": [],
    "ira_ingest.hip": ["-ffp-contract=off"],
    "ira_deconv.hip": ["-ffp-contract=off"],
    # float32 arithmetic of the reference is reproduced operation by operation: no FMA contraction
    "ira_diffusion.hip": ["-ffp-contract=off"],
}
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(exe).exists():
        raise RuntimeError("hipcc not found; libira cannot be built")
    return exe


def _stale(obj: Path, deps) -> bool:
    if not obj.exists():
        return True
    t = obj.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def build_tuning(verbose: bool = True) -> Path:
    """The TUNING build: the same sources with -DIRA_TUNING_BUILD, i.e. with the IRA_* environment knobs (ablations, tile
    and radix overrides, A/B kernel selection) compiled in -> csrc/libira_tuning.so.  Never loaded by the product; set
    IRA_TUNING=1 and point IRA_LIBRARY at it to profile (audio_analysis_amd._lib honours the pair for this purpose only and
    checks its ABI version like the product library's)."""
    hipcc = _hipcc()
    objs = []
    out = CSRC / "libira_tuning.so"
    tmp = CSRC / "_tuning"
    tmp.mkdir(exist_ok=True)
    for name, extra in SOURCES.items():
        obj = tmp / (Path(name).stem + ".o")
        cmd = [hipcc, *COMMON, "-DIRA_TUNING_BUILD", *extra, "-c", str(CSRC / name), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(out)], check=True)
    return out


def build(force: bool = False, verbose: bool = False) -> Path:
    hipcc = _hipcc()
    headers = list(CSRC.glob("*.h")) + [CSRC.parent.parent / "include" / "ira.h"]
    jobs = []
    objs = []
    for name, extra in SOURCES.items():
        src = CSRC / name
        obj = CSRC / (src.stem + ".o")
        objs.append(obj)
        if force or _stale(obj, [src, Path(__file__)] + headers):
            jobs.append([hipcc, *COMMON, *extra, "-c", str(src), "-o", str(obj)])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, flush=True)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _stale(LIB, objs):
        run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", *map(str, objs), "-o", str(LIB)])
    return LIB


if __name__ == "__main__":
    if "--tuning" in sys.argv:
        print(build_tuning())
        sys.exit(0)
    p = build(force="--force" in sys.argv, verbose=True)
    print(p)
