#!/usr/bin/env python3
"""Reduce the passes of tools/profile_config.sh: per-kernel durations and HBM-side counters, per-ABI-call traffic.

Byte estimates (MI355X_MICROARCH.md, HBM / rocprofv3 section):
  hbm_bytes_guide   = (2 * FETCH_SIZE + WRITE_SIZE) KiB   -- the guide's rule for wide coalesced reads (FETCH_SIZE tallies
                      128-byte requests at 64 bytes)
  fetch_bytes_req   = 32 * RDREQ_32B + 128 * (RDREQ - RDREQ_32B)   -- by request size, where the 32-byte request counter exists
                      (a blanket 2x over-counts patterns made of 32-byte pieces)
Per step = divided by the number of steps the run made: `steps_in_process` of the bench line in stats.log (bench.py counts
every step it submits; profiling runs skip the STFT-error probe, whose extra peak pick round 2's version of this tool
counted as a ninth step: every per-step figure of profiles/r02_traffic_{report,2}.json is 8/9 of the truth).  The
dispatch count of the once-per-step peak_decode_kernel must agree with it, or the tool stops.
"""
import collections, csv, glob, json, os, sys

d, cfg = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "report")

CALLS = [   # (substring of the kernel name, ABI call)
    ("stft3_kernel", "ira_stft_mag_db_tf[f32,n4096]"), ("stft4_kernel", "ira_stft_logbin[f64,n8192]"),
    ("stft5_kernel", "ira_stft_logbin[f64,n8192]"), ("stft6_kernel", "ira_stft_mag_db_tf[f32,n4096]"),
    ("smooth_half_split", "ira_rfft_smooth"), ("rows3_kernel", "ira_rfft_any"), ("edc_moments", "ira_edc_fits"),
    ("edc_line", "ira_edc_fits"), ("pcm16_jobs", "ira_pcm16_to_channels"),
    ("stft2_kernel<double, 1", "ira_stft_mag_db[f64,n4096,sel]"), ("stft2_kernel", "ira_stft_mag_db"),
    ("smooth_cols_kernel<0", "ira_rfft_smooth"), ("smooth_rows_kernel<0>", "ira_rfft_smooth"), ("smooth_rows_kernel<2>", "ira_rfft_smooth"),
    ("smooth_pair_split", "ira_rfft_smooth"),
    ("smooth_cols_kernel<1", "ira_band_irfft_smooth"), ("smooth_rows_kernel<1>", "ira_band_irfft_smooth"),
    ("smooth_rows_sparse", "ira_band_irfft_smooth"), ("band_compact", "ira_band_irfft_smooth"),
    ("cols_fwd_kernel<0>", "ira_rfft_any"), ("rows_kernel<1>", "ira_rfft_any"), ("cols_inv_kernel<0>", "ira_rfft_any"),
    ("pair_split_kernel", "ira_rfft_any"), ("half_split_kernel", "ira_rfft_any"),
    ("cols_fwd_kernel<1>", "ira_bluestein_filter"), ("rows_kernel<0>", "ira_bluestein_filter"),
    ("cols_fwd_kernel<2>", "ira_band_irfft"), ("cols_inv_kernel<1>", "ira_band_irfft"),
    ("ar_lag_kernel", "ira_ar_gram"), ("ar_gram_kernel", "ira_ar_gram"), ("ar_solve_dd", "ira_ar_exact"), ("ar_solve", "ira_ar_solve"), ("ar_grad", "ira_ar_refine"),
    ("ar_minnorm", "ira_ar_minnorm"), ("poly_roots", "ira_poly_roots"), ("ar_lag_dd", "ira_ar_exact"), ("ar_solve_dd", "ira_ar_exact"),
    ("order_stats", "ira_order_stats"), ("gd_uniform", "ira_group_delay"), ("gd_gradient", "ira_group_delay"),
    ("diffusion_mono", "ira_diffusion"), ("diffusion_stereo", "ira_diffusion_stereo"),
    ("edc_sums", "ira_edc_fits"), ("edc_carry", "ira_edc_fits"), ("edc_fit", "ira_edc_fits"), ("edc_emit", "ira_edc_fits"),
    ("crossing_search", "ira_curve_fits"), ("curve_fit", "ira_curve_fits"), ("peak_partial", "ira_peak_index"),
    ("peak_decode", "ira_peak_index"), ("mag_phase", "ira_spectrum_mag_phase"), ("unwrap", "ira_phase_unwrap"),
    ("stats_kernel", "ira_spectrum_stats"), ("waterfall", "ira_waterfall_rel"), ("logbin", "ira_logbin_aggregate"),
    ("host_pull", "ira_host_pull"), ("pcm16", "ira_pcm16_to_channels"),
]


def call_of(kernel):
    for sub, call in CALLS:
        if sub in kernel:
            return call
    return None


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0][:60]


kern = collections.defaultdict(dict)      # short kernel name -> {metric: value}
stats = glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    for r in csv.DictReader(open(stats[0])):
        k = short(r["Name"])
        kern[k]["calls"] = kern[k].get("calls", 0) + int(r["Calls"])
        kern[k]["total_ms"] = kern[k].get("total_ms", 0.0) + float(r["TotalDurationNs"]) / 1e6
for tag in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum"):
    files = glob.glob(os.path.join(d, tag, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    for r in csv.DictReader(open(files[0])):
        k = short(r["Kernel_Name"])
        c = r["Counter_Name"]
        kern[k][c] = kern[k].get(c, 0.0) + float(r["Counter_Value"])
        kern[k]["n_" + c] = kern[k].get("n_" + c, 0) + 1

steps = None
for k, v in kern.items():
    if "peak_decode" in k and v.get("calls"):
        steps = v["calls"]
try:
    line = json.loads(open(os.path.join(d, "stats.log")).read().strip().splitlines()[-1])
    said = int(line["steps_in_process"])
    if steps is not None and said != steps and cfg != "5":
        sys.exit(f"bench says {said} steps, rocprof saw {steps} peak_decode_kernel dispatches: something else launched a peak pick")
    steps = said
except (OSError, KeyError, ValueError, IndexError):
    pass
steps = steps or 1
rows = []
calls = collections.defaultdict(lambda: collections.defaultdict(float))
for k, v in sorted(kern.items(), key=lambda kv: -kv[1].get("total_ms", 0.0)):
    n = max(1, v.get("calls", 0))
    fetch = v.get("FETCH_SIZE", 0.0) * 1024.0 / max(1, v.get("n_FETCH_SIZE", 1))
    write = v.get("WRITE_SIZE", 0.0) * 1024.0 / max(1, v.get("n_WRITE_SIZE", 1))
    rq = v.get("TCC_EA0_RDREQ_sum", 0.0) / max(1, v.get("n_TCC_EA0_RDREQ_sum", 1))
    rq32 = v.get("TCC_EA0_RDREQ_32B_sum", 0.0) / max(1, v.get("n_TCC_EA0_RDREQ_32B_sum", 1))
    row = dict(kernel=k, launches_per_step=v.get("calls", 0) / steps, avg_us=1e3 * v.get("total_ms", 0.0) / n,
               ms_per_step=v.get("total_ms", 0.0) / steps, fetch_size_bytes=fetch, write_size_bytes=write,
               rdreq=rq, rdreq_32b=rq32, hbm_bytes_guide=2.0 * fetch + write,
               fetch_bytes_by_request=32.0 * rq32 + 128.0 * (rq - rq32))
    rows.append(row)
    call = call_of(k)
    if call:
        lps = row["launches_per_step"]
        calls[call]["ms_per_step"] += row["ms_per_step"]
        calls[call]["hbm_bytes_per_step_guide"] += lps * row["hbm_bytes_guide"]
        calls[call]["hbm_bytes_per_step_by_request"] += lps * (row["fetch_bytes_by_request"] + write)
        calls[call]["write_bytes_per_step"] += lps * write

with open(os.path.join(d, "per_kernel.csv"), "w") as f:
    cols = ["kernel", "launches_per_step", "avg_us", "ms_per_step", "fetch_size_bytes", "write_size_bytes", "rdreq", "rdreq_32b",
            "hbm_bytes_guide", "fetch_bytes_by_request"]
    f.write(",".join(cols) + "\n")
    for r in rows:
        f.write(",".join(f"\"{r[c]}\"" if c == "kernel" else f"{r[c]:.6g}" for c in cols) + "\n")
try:
    batch = json.loads(open(os.path.join(d, "stats.log")).read().strip().splitlines()[-1])["config"]["batch_per_gpu"]
except Exception:
    batch = {"report": 256, "literal": 256, "2": 256, "3": 256, "4": 256, "5": 256}.get(cfg, 64)   # channels per step
out = {"source": f"tools/profile_config.sh {cfg}: rocprofv3 --kernel-trace [--stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE | --pmc "
                 "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum], one pass each, IRA_STREAMS=1, copy-engine upload",
       "steps_in_run": steps, "batch": batch,
       "correction": "hbm_bytes_per_channel = (2*FETCH_SIZE + WRITE_SIZE)*1024 per step / batch (guide rule); the by-request "
                     "estimate (32 B x RDREQ_32B + 128 B x the rest + WRITE_SIZE) is carried beside it",
       "calls": {c: {"ms_per_step": v["ms_per_step"], "hbm_bytes_per_channel": v["hbm_bytes_per_step_guide"] / batch,
                     "hbm_bytes_per_channel_by_request": v["hbm_bytes_per_step_by_request"] / batch,
                     "write_bytes_per_channel": v["write_bytes_per_step"] / batch} for c, v in calls.items()}}
json.dump(out, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print(open(os.path.join(d, "per_kernel.csv")).read())
print(json.dumps(out["calls"], indent=1))
