import importlib
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
# the allocation fault injector (scaldpc_debug_fail_alloc) exists only in a process started with SCALDPC_DEBUG=1; the
# library reads the variable once, so it is set before anything loads it (tests/test_abi.py checks the refusal without it)
os.environ.setdefault("SCALDPC_DEBUG", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built_artefacts():
    """Built libraries are git-ignored: make sure the HIP library (cross-compiles without a GPU)
    and the test oracle exist before any test loads them.  A failed build fails the session
    loudly -- there is nothing to fall back to."""
    lib = importlib.import_module("sca-ldpc_amd._lib")
    if not os.path.exists(lib.SO_PATH):
        lib.build()
    from oracle import pyoracle

    pyoracle.build()


@pytest.fixture(scope="session")
def scaldpc():
    return importlib.import_module("sca-ldpc_amd")


@pytest.fixture(scope="session")
def golden():
    out = {}
    for f in os.listdir(GOLDEN):
        if f.endswith(".json"):
            with open(os.path.join(GOLDEN, f)) as fh:
                out[f[:-5]] = json.load(fh)
    return out


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.lib()
    return pyoracle
