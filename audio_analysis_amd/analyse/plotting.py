"""
CPU-side figure rendering (matplotlib).  Plotting is OUTSIDE the accelerated path (SURVEY.md section 2,
row 16): these helpers only exist so the plot_*_from_wav_file drop-ins still write their PNGs.  They are
deliberately small; they take the result dataclasses computed on the GPU and draw them.
"""
from __future__ import annotations

from pathlib import Path
from typing import Optional

import numpy as np

DEFAULT_FIGURE_SIZE = (10.0, 6.0)
DEFAULT_DPI = 100


def png_path(output_basename, suffix: str) -> Optional[Path]:
    """<basename stem><suffix>.png next to the basename, or None when no basename was given."""
    if output_basename is None:
        return None
    base = Path(output_basename)
    return base.with_name(f"{base.stem}{suffix}.png")


def _plt():
    import matplotlib.pyplot as plt
    return plt


def new_axes(title: Optional[str] = None, size=DEFAULT_FIGURE_SIZE, **kw):
    plt = _plt()
    fig, ax = plt.subplots(figsize=size, dpi=DEFAULT_DPI, **kw)
    if title:
        (ax if not isinstance(ax, np.ndarray) else ax.flat[0]).set_title(title)
    return fig, ax


def finish(fig, path: Optional[Path], show_interactive: bool) -> None:
    plt = _plt()
    if path is not None:
        Path(path).parent.mkdir(parents=True, exist_ok=True)
        fig.savefig(path, bbox_inches="tight")
    elif show_interactive:
        plt.show()
    plt.close(fig)


def _log_hz(ax, lo, hi, axis="x"):
    ticks = [t for t in (20, 50, 100, 200, 500, 1000, 2000, 5000, 10000, 20000) if lo <= t <= hi]
    fmt = lambda v, _p: f"{int(v / 1000)}k" if v >= 1000 else f"{int(v)}"
    import matplotlib.ticker as mt
    if axis == "x":
        ax.set_xscale("log"); ax.set_xlim(lo, hi); ax.set_xticks(ticks)
        ax.xaxis.set_major_formatter(mt.FuncFormatter(fmt)); ax.xaxis.set_minor_formatter(mt.NullFormatter())
    else:
        ax.set_yscale("log"); ax.set_ylim(lo, hi); ax.set_yticks(ticks)
        ax.yaxis.set_major_formatter(mt.FuncFormatter(fmt)); ax.yaxis.set_minor_formatter(mt.NullFormatter())


def render_ir_views(wav_path, settings, output_basename, expected_sample_rate_hz):
    """The three impulse-response PNGs of one file (reads the WAV itself: it runs in a plot worker as well)."""
    from . import impulse_response as irv
    from .io import load_wav_file
    loaded = load_wav_file(wav_path, expected_sample_rate_hz=expected_sample_rate_hz,
                           expected_channel_mode="mono_or_stereo", allow_mono_and_upmix_to_stereo=False)
    irv.plot_ir_views(loaded, settings, output_basename, False)


def render_decay(results, settings, plot_settings, title, path, show):
    if path is None and not show:
        return
    fig, ax = new_axes(title)
    for i, r in enumerate(results):
        alpha = 1.0 if i == 0 else plot_settings.secondary_channel_alpha
        step = max(1, r.edc_db.size // 20000)
        ax.plot(r.time_seconds[::step], r.edc_db[::step], alpha=alpha)
        if plot_settings.show_fit_lines:
            for f in r.fits.values():
                tt = np.array([f.start_time_seconds, f.end_time_seconds])
                ax.plot(tt, f.slope_db_per_second * tt + f.intercept_db, "--", alpha=alpha,
                        label=f"{f.name} {r.channel_name}  {f.rt60_seconds:.2f}s")
    ax.set_xlabel("Time (s)"); ax.set_ylabel("Level (dB)"); ax.set_ylim(*plot_settings.ylim_db)
    ax.grid(True, linestyle=":"); ax.legend(loc="best")
    finish(fig, path, show)


def render_spectrogram(result, settings, plot_settings, title, path, show):
    if path is None and not show:
        return
    fig, ax = new_axes(title)
    nyq = 0.5 * result.sample_rate_hz
    lo = float(np.clip(settings.f_min_hz, 1.0, nyq)); hi = float(np.clip(settings.f_max_hz, lo, nyq))
    sel = (result.frequency_hz >= lo) & (result.frequency_hz <= hi)
    mag = result.magnitude_db[sel]
    vmax = plot_settings.vmax_db if plot_settings.vmax_db is not None else float(np.percentile(mag, 99.5))
    if plot_settings.vmin_db is not None:
        vmin = plot_settings.vmin_db
    elif settings.dynamic_range_db is not None:
        vmin = vmax - float(settings.dynamic_range_db)
    else:
        vmin = float(np.percentile(mag, 5.0))
    mesh = ax.pcolormesh(result.time_seconds, result.frequency_hz[sel], mag, shading="auto", vmin=vmin, vmax=vmax)
    ax.set_xlabel("Time (s)"); ax.set_ylabel("Frequency (Hz)"); _log_hz(ax, lo, hi, "y")
    fig.colorbar(mesh, ax=ax, label="Magnitude (dB)")
    finish(fig, path, show)


def render_frequency_response(results, settings, plot_settings, title, path, show):
    if path is None and not show:
        return
    fig, ax = new_axes(title)
    nyq = 0.5 * results[0].sample_rate_hz
    lo = float(np.clip(settings.f_min_hz, 1.0, nyq)); hi = float(np.clip(settings.f_max_hz, lo, nyq))
    for i, r in enumerate(results):
        sel = (r.frequency_hz >= lo) & (r.frequency_hz <= hi)
        ax.plot(r.frequency_hz[sel], r.magnitude_db[sel], alpha=1.0 if i == 0 else plot_settings.secondary_channel_alpha,
                label=f"{r.channel_name}  peak={r.peak_frequency_hz:.0f}Hz  centroid={r.spectral_centroid_hz:.0f}Hz")
    _log_hz(ax, lo, hi, "x")
    ax.set_xlabel("Frequency (Hz)"); ax.set_ylabel("Level (dB)")
    if plot_settings.ylim_db is not None:
        ax.set_ylim(*plot_settings.ylim_db)
    ax.grid(True, which="both", linestyle=":"); ax.legend(loc="best")
    finish(fig, path, show)


def render_diffusion(results, title, path, show):
    if path is None and not show:
        return
    fig, ax = new_axes(title)
    ax.set_xlabel("Time (s)"); ax.set_ylabel("Metric (unitless)")
    ax.set_ylim(-0.05, 1.25)
    for i, r in enumerate(results):
        a = 1.0 if i == 0 else 0.7
        ax.plot(r.series.time_seconds, r.series.max_abs_autocorr, alpha=a, label=f"max|autocorr| {r.channel_name}")
        ax.plot(r.series.time_seconds, r.series.echo_density, alpha=a, linestyle="--",
                label=f"echo_density {r.channel_name}")
    if results and results[0].series.corr0 is not None and results[0].series.iacc_max is not None:
        n = min(results[0].series.time_seconds.size, results[0].series.corr0.size)
        ax.plot(results[0].series.time_seconds[:n], results[0].series.corr0[:n], linestyle=":", label="corr0 (L,R)")
        ax.plot(results[0].series.time_seconds[:n], results[0].series.iacc_max[:n], linestyle="-.",
                label="IACC max (±lag)")
    ax.grid(True, which="both", linestyle=":", linewidth=0.5); ax.legend(loc="best")
    finish(fig, path, show)


def render_group_delay(result, settings, plot_settings, title, path, show):
    if path is None and not show:
        return
    fig, ax = new_axes(title)
    ax.plot(result.frequency_hz, result.group_delay_samples,
            alpha=plot_settings.secondary_channel_alpha if result.channel_name != "L" else 1.0)
    ax.set_xscale("log")
    ax.set_xlabel("Frequency (Hz)"); ax.set_ylabel("Group delay (samples)")
    if plot_settings.show_zero_line:
        ax.axhline(0.0, linestyle="--", linewidth=1.0)
    if plot_settings.ylim_samples is not None:
        ax.set_ylim(*plot_settings.ylim_samples)
    finish(fig, path, show)


def render_filter_response(results, settings, plot_settings, title, path, show):
    if path is None and not show:
        return
    plt = _plt()
    fig, (ax_m, ax_p) = plt.subplots(2, 1, figsize=(10, 8))
    fig.suptitle(title, fontsize=12, fontweight="bold")
    nyq = 0.5 * results[0].sample_rate_hz
    lo = float(np.clip(settings.f_min_hz, 1.0, nyq)); hi = float(np.clip(settings.f_max_hz, lo, nyq))
    for i, r in enumerate(results):
        sel = (r.frequency_hz >= lo) & (r.frequency_hz <= hi)
        a = 1.0 if i == 0 else plot_settings.secondary_channel_alpha
        ax_m.plot(r.frequency_hz[sel], r.magnitude_db[sel], alpha=a,
                  label=f"{r.channel_name}  peak={r.peak_frequency_hz:.0f}Hz  @1kHz={r.magnitude_at_1khz_db:.1f}dB")
        ax_p.plot(r.frequency_hz[sel], r.phase_response[sel], alpha=a, label=r.channel_name)
    for ax, lab in ((ax_m, "Magnitude (dB)"), (ax_p, f"Phase ({'degrees' if settings.phase_mode == 'degrees' else 'radians'})")):
        ax.set_xscale("log"); ax.set_xlim(lo, hi); ax.set_xlabel("Frequency (Hz)"); ax.set_ylabel(lab)
        ax.grid(True, which="both", linestyle=":"); ax.legend(loc="best", fontsize=9)
    if plot_settings.magnitude_ylim_db is not None:
        ax_m.set_ylim(plot_settings.magnitude_ylim_db)
    if plot_settings.phase_ylim is not None:
        ax_p.set_ylim(plot_settings.phase_ylim)
    fig.tight_layout()
    finish(fig, path, show)


def render_rt60_bands(results, settings, plot_settings, title, path, show):
    if path is None and not show:
        return
    fig, ax = new_axes(title)
    bands = results[0].band_definitions
    cols = [("T30", "rt60_t30_seconds", "-")]
    if settings.include_t20:
        cols.append(("T20", "rt60_t20_seconds", "--"))
    if settings.include_edt:
        cols.append(("EDT", "edt_seconds", ":"))
    names = [b.name for b in bands]
    centres = np.array([b.centre_hz for b in bands], dtype=np.float32)
    as_bars = len(bands) <= 6
    groups = len(cols) * len(results)
    width = 0.8 / max(1, groups)
    slot = 0
    for i, ch in enumerate(results):
        a = 1.0 if i == 0 else plot_settings.secondary_channel_alpha
        for key, attr, ls in cols:
            vals = [getattr(ch.band_metrics_by_name[n], attr) for n in names]
            arr = np.array([np.nan if v is None else v for v in vals], dtype=np.float32)
            label = f"{key} {ch.channel_name}"
            if plot_settings.legend_values:
                label += "  " + "  ".join(f"{n}={'NA' if v is None else f'{v:.2f}s'}" for n, v in zip(names, vals))
            if as_bars:
                ax.bar(np.arange(len(bands)) + (slot - groups / 2) * width + width / 2, arr, width=width, alpha=a, label=label)
            else:
                ax.plot(centres, arr, linestyle=ls, marker="o", alpha=a, label=label)
            slot += 1
    if as_bars:
        ax.set_xticks(np.arange(len(bands))); ax.set_xticklabels(names); ax.set_xlabel("Band")
    else:
        ax.set_xscale("log"); ax.set_xlabel("Band centre frequency (Hz)")
    ax.set_ylabel("RT60 (seconds)")
    if plot_settings.ylim_seconds is not None:
        ax.set_ylim(*plot_settings.ylim_seconds)
    ax.grid(True, linestyle=":"); ax.legend(loc="best")
    finish(fig, path, show)


def render_waterfall(result, settings, plot_settings, title, path, show):
    if path is None and not show:
        return
    plt = _plt()
    f, t, z = result.frequency_hz, result.slice_times_seconds, result.slice_magnitude_rel_db
    if str(plot_settings.style).lower() == "2d":
        fig, ax = new_axes(title)
        for i in range(z.shape[0]):
            ax.plot(f, z[i] - i * plot_settings.ridge_offset_db, linewidth=0.8)
        _log_hz(ax, float(f[0]), float(f[-1]), "x")
        ax.set_xlabel("Frequency (Hz)"); ax.set_ylabel("Relative level (dB, offset per slice)")
    else:
        from mpl_toolkits.mplot3d import Axes3D  # noqa: F401
        fig = plt.figure(figsize=DEFAULT_FIGURE_SIZE, dpi=DEFAULT_DPI)
        ax = fig.add_subplot(111, projection="3d")
        ax.set_title(title)
        lf = np.log10(f.astype(np.float64))
        for i in range(z.shape[0] - 1, -1, -1):
            ax.plot(lf, np.full_like(lf, t[i]), z[i], linewidth=0.7)
        ticks = [v for v in (20, 50, 100, 200, 500, 1000, 2000, 5000, 10000, 20000) if f[0] <= v <= f[-1]]
        ax.set_xticks(np.log10(ticks)); ax.set_xticklabels([f"{int(v/1000)}k" if v >= 1000 else str(v) for v in ticks])
        ax.set_xlabel("Frequency (Hz)"); ax.set_ylabel("Time (s)"); ax.set_zlabel("Relative level (dB)")
        ax.view_init(elev=plot_settings.elev_deg, azim=plot_settings.azim_deg)
        if plot_settings.zlim_db is not None:
            ax.set_zlim(*plot_settings.zlim_db)
    finish(fig, path, show)


def render_modal_cloud(result, settings, plot_settings, title, path, show):
    if path is None and not show:
        return
    fig, ax = new_axes(title)
    nyq = 0.5 * result.sample_rate_hz
    lo = float(np.clip(settings.f_min_hz, 1.0, nyq)); hi = float(np.clip(settings.f_max_hz, lo, nyq))
    _log_hz(ax, lo, hi, "x")
    ax.set_xlabel("Frequency (Hz)"); ax.set_ylabel(f"RT60 estimate (s) [{result.metric.upper()}]")
    if not result.points:
        ax.text(0.5, 0.5, "No valid points (insufficient decay range).", transform=ax.transAxes, ha="center")
    else:
        fr = np.array([p.centre_hz for p in result.points]); rt = np.array([p.rt60_seconds for p in result.points])
        ax.scatter(fr, rt, s=12, alpha=0.85, label=f"{result.channel_name} ({len(result.points)} pts)")
        if plot_settings.show_median_curve and fr.size >= 8:
            lf = np.log2(fr); half = 0.5 * max(0.01, plot_settings.median_octave_window)
            keep = [(fr[i], float(np.median(rt[(lf >= lf[i] - half) & (lf <= lf[i] + half)])))
                    for i in range(fr.size) if int(np.sum((lf >= lf[i] - half) & (lf <= lf[i] + half))) >= 3]
            if len(keep) >= 4:
                ax.plot([k[0] for k in keep], [k[1] for k in keep], alpha=0.9, label=f"{result.channel_name} median")
        ax.legend(loc="best")
    if plot_settings.ylim_seconds is not None:
        ax.set_ylim(*plot_settings.ylim_seconds)
    ax.grid(True, which="both", linestyle=":")
    finish(fig, path, show)


def render_zplane(result, settings, plot_settings, title, path, show, rt60_of_radius):
    if path is None and not show:
        return
    fig, ax = new_axes(title, size=(7.5, 7.5))
    if plot_settings.show_axes:
        ax.axhline(0.0, linewidth=1.0); ax.axvline(0.0, linewidth=1.0)
    if plot_settings.show_unit_circle:
        th = np.linspace(0.0, 2.0 * np.pi, 512)
        ax.plot(np.cos(th), np.sin(th), linestyle="--", linewidth=1.0)
    if result.poles.size:
        ax.scatter(result.poles.real, result.poles.imag, marker="x", s=30, label="Poles")
    if result.zeros is not None and result.zeros.size:
        ax.scatter(result.zeros.real, result.zeros.imag, marker="o", s=18, facecolors="none", label="Zeros")
    lim = float(plot_settings.limit_radius)
    ax.set_aspect("equal", adjustable="box"); ax.set_xlim(-lim, lim); ax.set_ylim(-lim, lim)
    ax.set_xlabel("Re{z}"); ax.set_ylabel("Im{z}"); ax.legend(loc="upper right")
    if plot_settings.annotate_stats and result.poles.size:
        rad = np.abs(result.poles)
        mx, md = float(np.max(rad)), float(np.median(rad))
        ax.text(0.02, 0.02,
                f"AR order: {int(settings.ar_order)}\npoles: {result.poles.size}\n"
                f"unstable (|p|>=1): {int(np.sum(rad >= 1.0))}\nradius median: {md:.6f}\nradius max: {mx:.6f}\n"
                f"RT60~ (median r): {rt60_of_radius(min(md, 0.999999), result.sample_rate_hz):.3f} s\n"
                f"RT60~ (max r): {rt60_of_radius(min(mx, 0.999999), result.sample_rate_hz):.3f} s",
                transform=ax.transAxes, fontsize=9, va="bottom", ha="left")
    finish(fig, path, show)
