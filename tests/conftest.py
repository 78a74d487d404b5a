"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """All golden fixtures as one dict (name -> tensor) plus meta under key '__meta__'."""
    import json
    from safetensors.torch import load_file
    g = {}
    gd = os.path.join(ROOT, "tests", "golden")
    for fn in sorted(os.listdir(gd)):
        if fn.endswith(".safetensors"):
            g.update(load_file(os.path.join(gd, fn)))
    g["__meta__"] = json.load(open(os.path.join(gd, "meta.json"), encoding="utf-8"))
    return g
