"""
`analyse` -- import-name shim so that code written against the reference (`from analyse.decay import ...`,
`python -m analyse.cli report ...`) runs on the GPU implementation unchanged.  Every submodule re-exports
audio_analysis_amd.analyse.<same name>; nothing is implemented here.
"""
from audio_analysis_amd.analyse import *  # noqa: F401,F403
from audio_analysis_amd.analyse import __all__  # noqa: F401
