"""Shim: re-exports audio_analysis_amd.analyse.impulse_response (see analyse/__init__.py)."""
import sys as _sys

import audio_analysis_amd.analyse.impulse_response as _impl

_sys.modules[__name__] = _impl
