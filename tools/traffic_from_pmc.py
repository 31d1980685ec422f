#!/usr/bin/env python3
"""Reduce rocprofv3 counter CSVs (FETCH_SIZE / WRITE_SIZE passes of bench.py) to HBM bytes per ABI call.

Kernel -> call attribution follows the launch order: a rows_kernel belongs to the call whose cols_fwd_kernel
preceded it.  gfx950 correction (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE under-counts wide
reads by 2x, both counters are in KiB:  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.
Writes <dir>/traffic.json and <dir>/<counter>_per_kernel.csv; copy them into profiles/ for the judged record.
"""
import collections, csv, glob, json, os, sys

d, batch = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 64


def call_of(kernel, last_fft):
    k = kernel
    if "stft3_kernel" in k: return "ira_stft_mag_db_tf[f32,n4096]" if ", true>" in k else "ira_stft_mag_db[f32,n4096]", last_fft
    if "stft4_kernel" in k: return "ira_stft_logbin[f64,n8192]", last_fft
    if "stft2_kernel<double, 2" in k: return "ira_stft_mag_db[f64,n8192]", last_fft
    if "stft2_kernel<double, 1" in k: return "ira_stft_mag_db[f64,n4096,sel]", last_fft
    if "smooth_cols_kernel<0>" in k: return "ira_rfft_smooth", "ira_rfft_smooth"
    if "smooth_cols_kernel<1>" in k: return "ira_band_irfft_smooth", "ira_band_irfft_smooth"
    if "smooth_rows_kernel" in k or "smooth_pair_split" in k: return last_fft, last_fft
    if "cols_fwd_kernel<0>" in k: return "ira_rfft_any", "ira_rfft_any"
    if "cols_fwd_kernel<2>" in k: return "ira_band_irfft", "ira_band_irfft"
    if "cols_fwd_kernel<1>" in k: return "ira_bluestein_filter", "ira_bluestein_filter"
    if "rows_kernel" in k or "cols_inv_kernel" in k or "pair_split" in k or "half_split" in k: return last_fft, last_fft
    if "ar_lag_kernel" in k or "ar_gram_kernel" in k: return "ira_ar_gram", last_fft
    if "ar_solve" in k: return "ira_ar_solve", last_fft
    if "edc_" in k: return "ira_edc_db", last_fft
    if "curve_fit" in k: return "ira_curve_fits", last_fft
    if "logbin" in k: return "ira_logbin_aggregate", last_fft
    if "diffusion" in k: return "ira_diffusion", last_fft
    if "order_stats" in k: return "ira_order_stats", last_fft
    return None, last_fft


per = {}
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    by_kernel = collections.defaultdict(lambda: [0.0, 0])
    by_call = collections.defaultdict(float)
    last = None
    seen = set()
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        v = float(r["Counter_Value"])
        name = r["Kernel_Name"]
        by_kernel[name][0] += v; by_kernel[name][1] += 1
        call, last = call_of(name, last)
        if call:
            by_call[call] += v
    with open(os.path.join(d, f"{counter}_per_kernel.csv"), "w") as f:
        f.write("kernel,dispatches,total_KiB,avg_KiB\n")
        for k, (tot, cnt) in sorted(by_kernel.items(), key=lambda kv: -kv[1][0]):
            f.write(f"\"{k[:100]}\",{cnt},{tot:.1f},{tot / cnt:.2f}\n")
    per[counter] = dict(by_call)

steps = 3   # bench.py --steps 2 --warmup 1: every call runs in three steps
calls = {}
for call in sorted(set(per.get("FETCH_SIZE", {})) | set(per.get("WRITE_SIZE", {}))):
    f = per.get("FETCH_SIZE", {}).get(call, 0.0) / steps
    w = per.get("WRITE_SIZE", {}).get(call, 0.0) / steps
    calls[call] = {"fetch_kb_per_step": f, "write_kb_per_step": w,
                   "hbm_bytes_per_channel": (2.0 * f + w) * 1024.0 / batch}
json.dump({
    "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 bench.py --steps 2 "
              f"--warmup 1 --batch {batch} --no-cpu-baseline   (tools/profile_config.sh)",
    "correction": "hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE under-counts wide reads by 2x)",
    "batch": batch, "calls": calls}, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print(json.dumps(calls, indent=1))
