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
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ref_values.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def renderer():
    """One context for the whole GPU session (fails loudly if the HIP library or the GPU is missing)."""
    import raytrace_amd as rt
    r = rt.Renderer(0)
    yield r
    r.close()


@pytest.fixture(autouse=True)
def trees_for_small_scenes_too(monkeypatch):
    """The library scans scenes of up to 16 spheres even when a tree is asked for (the scan is faster there).  The
    tests ask for trees on the 6-sphere default scene and on scenes of two or three spheres precisely to exercise the
    tree kernels on nodes with few children, so they lift that rule; test_small_scenes_are_scanned checks the rule
    itself."""
    monkeypatch.setenv("RT_SCAN_UP_TO", "0")
