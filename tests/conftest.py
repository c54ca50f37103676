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
