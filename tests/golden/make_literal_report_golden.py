#!/usr/bin/env python3
"""
Golden Markdown of the reference's LITERAL default report (`ReportSettings()`: every block on, impulse-response plots
included, PNGs rendered), made by importing the reference in the build container on the WAV inputs already stored in
goldens.npz:

    PYTHONDONTWRITEBYTECODE=1 python3 tests/golden/make_literal_report_golden.py

Committed output: tests/golden/report_literal.json ({case: markdown with the input path replaced by {WAV}}).  Data only.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
from pathlib import Path

import numpy as np
from scipy.io import wavfile

HERE = Path(__file__).resolve().parent
sys.path[:] = [p for p in sys.path if p.rstrip("/") != "/root/reference"]
sys.path.insert(0, "/root/reference")
os.environ.setdefault("MPLBACKEND", "Agg")

import analyse.report as rreport  # noqa: E402  (the reference's)

SR = 48000


def main():
    g = np.load(HERE / "goldens.npz")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        tmp = Path(tmp)
        for name in ("stereo16", "mono16"):
            pcm = g[f"report/{name}/pcm"]
            wav = tmp / f"{name}.wav"
            wavfile.write(str(wav), SR, pcm[:, 0] if (pcm.ndim == 2 and pcm.shape[1] == 1) else pcm)
            for tag, settings in (("literal", rreport.ReportSettings()),
                                  ("literal_mono", rreport.ReportSettings(common_use_mono_downmix_for_stereo=True))):
                res = rreport.run_report_from_wav_file(wav, tmp / f"out_{name}_{tag}" / "rep", settings)
                out[f"{name}/{tag}"] = res.summary_markdown.replace(str(wav), "{WAV}")
                pngs = sorted(p.name for p in (tmp / f"out_{name}_{tag}").glob("*.png"))
                out[f"{name}/{tag}/pngs"] = pngs
    (HERE / "report_literal.json").write_text(json.dumps(out, indent=1))
    for k, v in out.items():
        print(k, len(v))


if __name__ == "__main__":
    main()
