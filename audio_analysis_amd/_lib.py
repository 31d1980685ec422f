"""
ctypes binding of libira.so (the C-ABI declared in include/ira.h).

There is NO fallback: if the shared library is missing or fails to load, importing the product path
raises.  Build it with `python -m audio_analysis_amd.build` (hipcc, gfx950).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "csrc" / "libira.so"
_lib = None
# must equal IRA_ABI_VERSION of include/ira.h: a stale .so called with this file's prototypes would read shifted
# arguments or undersized scratch (memory corruption on the GPU instead of a clean error)
ABI_VERSION = 8

c_f32p = C.c_void_p
c_i64p = C.c_void_p
c_i32p = C.c_void_p
c_f64p = C.c_void_p
vp = C.c_void_p
i32 = C.c_int32
f64 = C.c_double
f32 = C.c_float

# name -> (restype, argtypes); must list every symbol include/ira.h declares (checked by tests).
PROTOTYPES = {
    "ira_abi_version": (i32, []),
    "ira_error_string": (C.c_char_p, [i32]),
    "ira_peak_index": (i32, [vp, vp, vp, i32, C.c_int64, vp, vp, vp]),
    "ira_edc_db": (i32, [vp, vp, vp, i32, C.c_int64, f64, f64, vp, vp, vp, vp, vp]),
    "ira_curve_fits": (i32, [vp, vp, vp, i32, i32, f32, f32, vp, C.POINTER(f64), i32, i32, C.POINTER(f64), i32, i32,
                             f64, f64, vp, vp, vp]),
    "ira_edc_box_smooth": (i32, [vp, vp, vp, i32, C.c_int64, i32, f64, vp, vp]),
    "ira_edc_fits": (i32, [vp, vp, vp, i32, C.c_int64, f64, f64, f32, f32, C.POINTER(f64), i32, i32, C.POINTER(f64), i32,
                           vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ira_stft_mag_db": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, f64, vp, vp, vp, vp, vp]),
    "ira_stft_logbin": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, f64, i32, vp, vp, i32, vp, vp, vp]),
    "ira_stft_mag_db_tf": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp, i32, f64, vp, vp, vp, vp, vp]),
    "ira_fft_split": (i32, [i32, vp, vp]),
    "ira_bluestein_filter": (i32, [vp, i32, i32, vp, vp, vp, vp, vp]),
    "ira_rfft_any": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp, vp,
                           vp, i32, vp]),
    "ira_band_irfft": (i32, [vp, vp, vp, i32, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "ira_spectrum_mag_phase": (i32, [vp, vp, vp, i32, i32, f64, vp, vp, vp, vp, vp, vp]),
    "ira_phase_unwrap": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, vp, vp]),
    "ira_log_smooth_db": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "ira_group_delay": (i32, [vp, vp, vp, i32, i32, vp, f64, vp, i32, vp, vp]),
    "ira_fft_smooth_split": (i32, [i32, vp, vp]),
    "ira_rfft_smooth": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp]),
    "ira_band_irfft_smooth": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp]),
    "ira_band_tile_layout": (i32, [i32, i32, C.POINTER(i32), C.POINTER(i32)]),
    "ira_wav_probe": (i32, [C.c_char_p, vp, vp, vp, vp]),
    "ira_wav_read_pcm16": (i32, [C.c_char_p, C.c_int64, C.c_int64, i32, vp]),
    "ira_wav_probe_batch": (i32, [vp, i32, i32, vp, vp, vp, vp, vp]),
    "ira_wav_read_pcm16_batch": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "ira_pcm16_to_channels": (i32, [vp, C.c_int64, i32, i32, vp, vp]),
    "ira_pcm16_to_channels_jobs": (i32, [vp, vp, vp, vp, vp, vp, i32, C.c_int64, vp, vp]),
    "ira_band_mask_values": (i32, [C.POINTER(f64), f64, C.c_int64, vp, vp]),
    "ira_host_pull": (i32, [vp, C.c_int64, i32, vp, i32, vp]),
    "ira_deconv_divide": (i32, [vp, vp, vp, vp, vp, i32, i32, f64, vp, vp]),
    "ira_deconv_finish": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, f64, vp, vp, vp]),
    "ira_order_stats": (i32, [vp, vp, vp, i32, vp, i32, vp, vp]),
    "ira_diffusion": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, f64, f64, vp, vp, vp, vp]),
    "ira_diffusion_stereo": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp]),
    "ira_spectrum_stats": (i32, [vp, vp, vp, i32, vp, f64, f64, f64, vp, vp]),
    "ira_waterfall_rel": (i32, [vp, vp, vp, i32, i32, i32, i32, f64, vp, vp, vp]),
    "ira_logbin_aggregate": (i32, [vp, vp, vp, i32, i32, i32, vp, vp, i32, vp, vp, i32, vp]),
    "ira_ar_partial_doubles": (C.c_int64, [i32, i32]),
    "ira_ar_gram": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, i32, vp]),
    "ira_ar_solve": (i32, [vp, vp, i32, i32, i32, f64, vp, vp, vp, i32, vp]),
    "ira_ar_minnorm": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, f64, vp]),
    "ira_ar_exact_doubles": (C.c_int64, [i32, i32, i32]),
    "ira_ar_exact": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f64, vp, vp, vp, vp, vp, f64, vp]),
    "ira_ar_refine": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, f64, i32, i32, vp]),
    "ira_ar_fit": (i32, [vp, vp, vp, vp, vp, i32, i32, i32, f64, vp, vp, vp, vp, i32, vp]),
    "ira_poly_roots": (i32, [vp, i32, i32, f64, vp, vp, vp]),
    "ira_fir_numerator": (i32, [vp, i32, vp, vp, vp, vp, i32, i32, vp, vp]),
}


class IraError(RuntimeError):
    pass


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load libira.so once; raise loudly if it is absent (no CPU fallback exists by design)."""
    global _lib
    if _lib is not None:
        return _lib
    import os
    # IRA_LIBRARY: the separately built tuning / ablation library (python -m audio_analysis_amd.build --tuning), honoured
    # only together with IRA_TUNING=1 (profiling tools set both); it passes the same ABI check as the product library
    tuning = os.environ.get("IRA_LIBRARY") if os.environ.get("IRA_TUNING") == "1" else None
    path = Path(tuning) if tuning else _LIB_PATH
    if not path.exists():
        raise IraError(
            f"{path} not found. The HIP library is the product path and has no fallback; "
            "build it with `python -m audio_analysis_amd.build`."
        )
    # torch bundles its own libamdhip64; load it FIRST so libira.so resolves to the same HIP runtime instance
    # (two runtimes in one process do not share devices, streams or allocations).
    import torch  # noqa: F401

    lib = C.CDLL(str(path))
    rebuild = "rebuild it with `python -m audio_analysis_amd.build`" + (" --tuning" if tuning else "")
    # The version FIRST: a stale library from an older ABI lacks the newer symbols, and binding them before this check would
    # end in a bare AttributeError instead of the message a stale library is the case for.
    try:
        version = lib.ira_abi_version
    except AttributeError:
        raise IraError(f"{path} exports no ira_abi_version (not a libira build, or one older than the versioned ABI): "
                       f"{rebuild}") from None
    version.restype, version.argtypes = PROTOTYPES["ira_abi_version"]
    have = int(version())
    if have != ABI_VERSION:
        raise IraError(f"{path} has ABI version {have}, this package binds version {ABI_VERSION}: {rebuild}")
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError:
            raise IraError(f"{path} (ABI version {have}) does not export {name}: {rebuild}") from None
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ira_error_string(rc)
        raise IraError(f"libira {what} failed with code {rc}: {msg.decode() if msg else '?'}")


def dbl_array(values):
    arr = (f64 * max(1, len(values)))(*[float(v) for v in values])
    return arr
