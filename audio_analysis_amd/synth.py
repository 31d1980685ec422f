"""
Deterministic synthetic impulse responses.

For IR index i, channel c:
  rng = numpy.random.default_rng(1_000_003*i + c); RT60 = 0.3 + 2.7*rng.random() s;
  pre-delay d = 240 + (i mod 512) samples; x[n<d] = 0;
  x[n>=d] = g[n-d] * 10^(-3 (n-d)/(sr*RT60)), g ~ N(0,1);
  direct sound x[d] = 1.5*max|x|; peak-normalise to 0.95; float32.

Variants used by the parity suites only: one-pole low-pass colouring (AR conditioning) and a
PCM16 round trip int16(x*32767)/32768 (what the reference's C++ bundle recorder
include/analysis/recorder.hpp:49-53 followed by analyse/io.py:58-59 produces).
"""

from __future__ import annotations

from typing import Optional

import numpy as np

SAMPLE_RATE_HZ = 48_000


def synth_ir(
    index: int,
    channel: int = 0,
    num_samples: int = 480_000,
    sample_rate_hz: int = SAMPLE_RATE_HZ,
    rt60_seconds: Optional[float] = None,
    lowpass_pole: float = 0.0,
    pcm16_round_trip: bool = False,
    pre_delay: Optional[int] = None,
) -> np.ndarray:
    rng = np.random.default_rng(1_000_003 * int(index) + int(channel))
    drawn = 0.3 + 2.7 * rng.random()
    rt60 = float(drawn if rt60_seconds is None else rt60_seconds)
    d = int(240 + (int(index) % 512)) if pre_delay is None else int(pre_delay)
    d = min(d, max(0, num_samples - 1))
    n_tail = num_samples - d
    g = rng.standard_normal(n_tail)
    env = 10.0 ** (-3.0 * np.arange(n_tail, dtype=np.float64) / (float(sample_rate_hz) * rt60))
    tail = g * env
    if lowpass_pole > 0.0:
        # y[n] = (1-a) x[n] + a y[n-1]; plain recursion keeps the generator dependency-free.
        a = float(lowpass_pole)
        y = np.empty_like(tail)
        acc = 0.0
        b = 1.0 - a
        for k in range(tail.size):
            acc = b * tail[k] + a * acc
            y[k] = acc
        tail = y
    x = np.zeros(num_samples, dtype=np.float64)
    x[d:] = tail
    x[d] = 1.5 * float(np.max(np.abs(x)))
    x *= 0.95 / float(np.max(np.abs(x)))
    x32 = x.astype(np.float32)
    if pcm16_round_trip:
        q = (x32 * np.float32(32767.0)).astype(np.int16)       # truncating, like int16_t(x*32767.0f)
        x32 = (q.astype(np.float32) / np.float32(32768.0)).astype(np.float32)
    return x32


def synth_batch(first_index: int, count: int, num_samples: int, channel: int = 0, **kw) -> np.ndarray:
    """(count, num_samples) float32, IR indices first_index .. first_index+count-1."""
    out = np.empty((count, num_samples), dtype=np.float32)
    for k in range(count):
        out[k] = synth_ir(first_index + k, channel, num_samples, **kw)
    return out
