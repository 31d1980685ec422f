#!/usr/bin/env python3
"""
End-to-end rate of the batched bundle path on one GPU, files on disk -> gathered metric records:
    native ingest (fread of PCM16 taps into pinned memory, int16 upload, conversion on the device)
  + metrics-only full report (pipeline.FullReport) + record fetch,
through audio_analysis_amd.analyse.bundle.run_bundle_metrics.  Writes a synthetic bundle in the recorder's on-disk
format (44-byte header, stereo PCM16; reference include/analysis/recorder.hpp:55-126) under --root first.

    python tools/bundle_rate.py [--taps 128] [--seconds 10] [--per-step 32] [--root /tmp/ira_bundle]
"""
import argparse, json, os, struct, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np


def write_tap(path, stereo_f32, sr=48000):
    pcm = np.trunc(np.clip(stereo_f32, -1.0, 1.0).astype(np.float32) * np.float32(32767.0)).astype("<i2")
    data = pcm.tobytes()
    hdr = (b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 2, sr, sr * 4, 4, 16)
           + b"data" + struct.pack("<I", len(data)))
    with open(path, "wb") as f:
        f.write(hdr + data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--taps", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--per-step", type=int, default=32)
    ap.add_argument("--root", default="/tmp/ira_bundle")
    a = ap.parse_args()
    from audio_analysis_amd.synth import synth_ir
    n = int(a.seconds * 48000)
    os.makedirs(os.path.join(a.root, "taps"), exist_ok=True)
    names = [f"tap{i:05d}" for i in range(a.taps)]
    t0 = time.perf_counter()
    for i, name in enumerate(names):
        st = np.stack([synth_ir(i, 0, n), synth_ir(i, 1, n)], axis=1)
        write_tap(os.path.join(a.root, "taps", name + ".wav"), st)
    with open(os.path.join(a.root, "meta.json"), "w") as f:
        json.dump({"sample_rate_hz": 48000, "length_samples": n, "taps": names}, f)
    t_write = time.perf_counter() - t0

    import torch
    from audio_analysis_amd.analyse import bundle
    bundle.run_bundle_metrics(a.root, taps_per_step=a.per_step)                 # warm-up: plans, tables, page cache
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    labels, rec = bundle.run_bundle_metrics(a.root, taps_per_step=a.per_step)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert rec.shape[0] == 2 * a.taps and len(labels) == 2 * a.taps
    print(json.dumps({"taps": a.taps, "channels": 2 * a.taps, "seconds_per_tap": a.seconds, "taps_per_step": a.per_step,
                      "wall_s": dt, "files_per_s": a.taps / dt, "channels_per_s": 2 * a.taps / dt,
                      "pcm_bytes": 4 * n * a.taps, "ingest_GBps_equivalent": 4 * n * a.taps / dt / 1e9,
                      "bundle_write_s": t_write}))


if __name__ == "__main__":
    main()
