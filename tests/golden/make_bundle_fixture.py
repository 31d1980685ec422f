#!/usr/bin/env python3
"""
Generate tests/golden/bundle/ -- a small bundle WRITTEN BY THE REFERENCE's C++ recorder and READ BY THE REFERENCE's
Python loader, in the build container:

    make -C oracle/ref_bundle            # g++ on the reference's header where it lies -> oracle/_ref/ref_bundle_driver
    python3 tests/golden/make_bundle_fixture.py

Committed outputs (data only, no reference source):
    tests/golden/bundle/meta.json, taps/<name>.wav     bytes written by include/analysis/recorder.hpp:55-126
    tests/golden/bundle_expected.npz                   the float32 taps fed to the recorder, and what the reference's
                                                       analyse.io.load_wav_file + get_analysis_channels return for
                                                       each tap (left/right and the --mono downmix)
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
sys.path.insert(0, str(REPO))
sys.path[:] = [p for p in sys.path if p.rstrip("/") != "/root/reference"]
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

from audio_analysis_amd.synth import synth_ir  # noqa: E402

import analyse.io as rio  # noqa: E402  (the reference's)

SR = 48000
FRAMES = 12000          # 0.25 s = 25 blocks of 480
BLOCK = 480
TAPS = ["early", "late_hot"]


def main():
    drv = REPO / "oracle" / "_ref" / "ref_bundle_driver"
    assert drv.exists(), "run `make -C oracle/ref_bundle` first"
    taps = []
    for i, name in enumerate(TAPS):
        l = synth_ir(40 + i, 0, FRAMES, SR, rt60_seconds=0.08)
        r = synth_ir(40 + i, 1, FRAMES, SR, rt60_seconds=0.08)
        st = np.stack([l, r], axis=1).astype(np.float32)
        if name == "late_hot":
            st *= np.float32(1.3)            # drives samples past +-1: the recorder clamps (recorder.hpp:49-53)
            st[100, 0], st[101, 1] = np.float32(-1.0), np.float32(1.0)
        taps.append(st)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        raw = Path(tmp) / "in.f32"
        np.concatenate([t.reshape(-1) for t in taps]).astype("<f4").tofile(raw)
        subprocess.run([str(drv), str(raw), str(FRAMES), str(BLOCK), str(Path(tmp) / "runs"), *TAPS], check=True)
        (run,) = list((Path(tmp) / "runs").iterdir())          # <out>/<timestamp>/
        dst = HERE / "bundle"
        shutil.rmtree(dst, ignore_errors=True)
        (dst / "taps").mkdir(parents=True)
        shutil.copy(run / "meta.json", dst / "meta.json")
        for name, st in zip(TAPS, taps):
            shutil.copy(run / "taps" / f"{name}.wav", dst / "taps" / f"{name}.wav")
            out[f"{name}/input"] = st
            loaded = rio.load_wav_file(dst / "taps" / f"{name}.wav", expected_channel_mode="mono_or_stereo",
                                       allow_mono_and_upmix_to_stereo=False)
            for mono in (False, True):
                for ch_name, x in rio.get_analysis_channels(loaded, mono):
                    out[f"{name}/{'mix' if mono else 'split'}/{ch_name}"] = np.asarray(x)
    np.savez_compressed(HERE / "bundle_expected.npz", **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
