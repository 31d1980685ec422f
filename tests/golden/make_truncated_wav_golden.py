#!/usr/bin/env python3
"""
Generate tests/golden/truncated_wav.json: what the REFERENCE's loader (analyse.io.load_wav_file -> scipy.io.wavfile.read,
io.py:181-221) returns for PCM16 files whose data chunk is shorter than its header says -- run in the build container:

    python3 tests/golden/make_truncated_wav_golden.py

Committed output (data only): per case the file's bytes (hex), and either the shape + samples the reference returned
or the exception type it raised.
"""
from __future__ import annotations

import json
import os
import struct
import sys
import tempfile
import warnings
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path[:] = [p for p in sys.path if p.rstrip("/") != "/root/reference"]
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

import analyse.io as rio  # noqa: E402  (the reference's)


def wav(channels: int, claimed_bytes: int, payload: bytes) -> bytes:
    fmt = struct.pack("<HHIIHH", 1, channels, 48000, 48000 * 2 * channels, 2 * channels, 16)
    body = b"WAVE" + b"fmt " + struct.pack("<I", 16) + fmt + b"data" + struct.pack("<I", claimed_bytes) + payload
    return b"RIFF" + struct.pack("<I", len(body)) + body


def main() -> None:
    rng = np.random.default_rng(20261004)
    pcm = rng.integers(-32768, 32767, size=64, dtype=np.int16).astype("<i2").tobytes()
    cases = {
        "stereo_100_of_4000": wav(2, 4000, pcm[:100]),          # 25 whole frames
        "mono_24_of_30": wav(1, 30, pcm[:24]),                    # 12 samples
        "stereo_stray_byte": wav(2, 400, pcm[:101]),              # 50 samples + one stray byte
        "stereo_odd_samples": wav(2, 400, pcm[:102]),             # 51 samples: no whole number of frames
        "mono_nothing": wav(1, 64, b""),                          # header only
    }
    out = {}
    with tempfile.TemporaryDirectory() as td:
        for name, blob in cases.items():
            p = Path(td) / f"{name}.wav"
            p.write_bytes(blob)
            rec = {"file_hex": blob.hex()}
            try:
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    la = rio.load_wav_file(p, expected_channel_mode="mono_or_stereo", allow_mono_and_upmix_to_stereo=False)
                rec["warnings"] = sorted({type(x.message).__name__ for x in w})
                s = np.asarray(la.samples)
                rec["shape"] = list(s.shape)
                rec["samples_f32_hex"] = s.astype("<f4").tobytes().hex()
            except Exception as e:  # noqa: BLE001
                rec["raises"] = type(e).__name__
            out[name] = rec
    import scipy
    out["_meta"] = {"numpy": np.__version__, "scipy": scipy.__version__}
    (HERE / "truncated_wav.json").write_text(json.dumps(out, indent=1))
    print({k: (v.get("shape"), v.get("raises")) for k, v in out.items() if k != "_meta"})


if __name__ == "__main__":
    main()
