import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
STREAMS = ["test_640x360", "test_1920x1080", "test_1920x1080_fullRange"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    return json.load(open(os.path.join(GOLDEN_DIR, "golden.json")))


def stream_bytes(name):
    return open(os.path.join(GOLDEN_DIR, name + ".h264"), "rb").read()


@pytest.fixture(scope="session")
def built():
    """Product library + oracle, compiled in-tree (no-ops when up to date)."""
    import h264bsd_amd
    from oracle import pyoracle
    h264bsd_amd.build()
    pyoracle.build(ref=True)
    return h264bsd_amd


_capture_cache = {}


@pytest.fixture(scope="session")
def captured(built):
    """name -> (jobs, trace, info) from the product's host parser (capture mode, no GPU)."""
    def get(name):
        if name not in _capture_cache:
            _capture_cache[name] = built.capture_stream(stream_bytes(name))
        return _capture_cache[name]
    return get
