"""Shim: re-exports audio_analysis_amd.analyse.filterplot (see analyse/__init__.py)."""
import sys as _sys

import audio_analysis_amd.analyse.filterplot as _impl

_sys.modules[__name__] = _impl
