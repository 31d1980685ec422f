"""
Bundle runner: meta.json + taps/<name>.wav -> reports/<name>/<name>_report.md + reports/bundle_report.md.

Host-side mirror of the reference's analyse/bundle.py (:29-74).  Bundle layout is the one the reference's C++
recorder writes (include/analysis/recorder.hpp:102-126).  Taps are independent files, so with
torch.distributed initialised (one process per GPU) each rank analyses a contiguous block of taps
(audio_analysis_amd.dist.shard_files) and rank 0 writes the index; single process = the reference's loop.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional

from .. import dist as _dist
from .report import ReportSettings, run_report_from_wav_file


@dataclass(frozen=True)
class BundleRunSettings:
    reports_subdir: str = "reports"
    report_settings: Optional[ReportSettings] = None


def run_bundle_report(bundle_root: str | Path, settings: Optional[BundleRunSettings] = None) -> Path:
    settings = settings or BundleRunSettings()
    root = Path(bundle_root)
    meta = json.loads((root / "meta.json").read_text())
    taps: List[str] = list(meta.get("taps", []))
    reports = root / settings.reports_subdir
    reports.mkdir(parents=True, exist_ok=True)

    rank, _, world = _dist.env_world()
    lo, hi = _dist.shard_files(len(taps), rank, world)
    for tap in taps[lo:hi]:
        out_dir = reports / tap
        out_dir.mkdir(parents=True, exist_ok=True)
        run_report_from_wav_file(input_wav_file_path=root / "taps" / f"{tap}.wav", output_basename=out_dir / tap,
                                 settings=settings.report_settings)
    _dist.barrier()

    index = reports / "bundle_report.md"
    if rank == 0:
        lines = [
            "# IR Bundle Report\n",
            f"**Bundle:** `{root}`\n",
            f"**Sample rate:** {meta.get('sample_rate_hz')}\n",
            f"**Length (samples):** {meta.get('length_samples')}\n",
            "\n## Taps\n",
        ]
        lines += [f"- [{tap}]({settings.reports_subdir}/{tap}/{tap}_report.md)" for tap in taps]
        index.write_text("\n".join(lines) + "\n")
    return index
