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
