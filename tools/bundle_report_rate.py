#!/usr/bin/env python3
"""
Bundle REPORT rate (Markdown + optional PNGs) on one GPU: the reference's serial per-tap loop (taps_per_batch = 1, inline
rendering) against the batched path (every block once over the channels of 16 taps) and PNG worker processes.

    python tools/bundle_report_rate.py [--taps 16] [--seconds 10] [--root /tmp/ira_bundle_rep]
"""
import argparse, json, os, shutil, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tools.bundle_rate import write_tap


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--taps", type=int, default=16)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--root", default="/tmp/ira_bundle_rep")
    ap.add_argument("--plots", type=int, default=1)
    a = ap.parse_args()
    from audio_analysis_amd.synth import synth_ir
    n = int(a.seconds * 48000)
    shutil.rmtree(a.root, ignore_errors=True)
    os.makedirs(os.path.join(a.root, "taps"))
    names = [f"tap{i:03d}" for i in range(a.taps)]
    for i, name in enumerate(names):
        write_tap(os.path.join(a.root, "taps", name + ".wav"), np.stack([synth_ir(i, 0, n), synth_ir(i, 1, n)], axis=1))
    with open(os.path.join(a.root, "meta.json"), "w") as f:
        json.dump({"sample_rate_hz": 48000, "length_samples": n, "taps": names}, f)
    import torch
    from audio_analysis_amd.analyse import bundle, report as rp
    out = {"taps": a.taps, "seconds": a.seconds}

    def run(tag, render, per_batch, workers):
        rs = rp.ReportSettings(run_impulse_response_plots=False, render_plots=render)
        s = bundle.BundleRunSettings(reports_subdir=f"reports_{tag}", report_settings=rs, taps_per_batch=per_batch,
                                     plot_workers=workers)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bundle.run_bundle_report(a.root, s)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[tag] = {"wall_s": dt, "files_per_s": a.taps / dt}
        print(tag, out[tag], flush=True)

    run("warmup", False, 16, 0)
    run("markdown_serial", False, 1, 0)
    run("markdown_batched16", False, 16, 0)
    if a.plots:
        cores = len(os.sched_getaffinity(0))
        run("png_inline_batched16", True, 16, 0)
        run(f"png_workers{min(15, cores - 1)}_batched16", True, 16, min(15, cores - 1))
    a_md = open(os.path.join(a.root, "reports_markdown_serial", names[0], names[0] + "_report.md")).read()
    b_md = open(os.path.join(a.root, "reports_markdown_batched16", names[0], names[0] + "_report.md")).read()
    out["markdown_identical"] = a_md == b_md
    print(json.dumps(out))


if __name__ == "__main__":
    main()
