"""Shim: `python -m analyse.cli ...` runs audio_analysis_amd.analyse.cli."""
from audio_analysis_amd.analyse.cli import build_parser, main, parse_arguments  # noqa: F401

if __name__ == "__main__":
    main()
