"""Shared pytest configuration: the `gpu` marker, golden-fixture loader, import paths."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))
os.environ.setdefault("MPLBACKEND", "Agg")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    g = np.load(REPO / "tests" / "golden" / "goldens.npz")
    meta = json.loads((REPO / "tests" / "golden" / "goldens.json").read_text())
    return g, meta["cases"], meta
