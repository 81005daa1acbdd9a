import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


# tests/ab needs the -DORBX_AB build of the library (ORB_LIB): tests/test_ab_child.py runs it in a child process
collect_ignore = [] if os.environ.get("ORB_AB_CHILD") == "1" else ["ab"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    try:                                                    # property tests explore the same examples on every run
        import hypothesis
        hypothesis.settings.register_profile("repro", derandomize=True, deadline=None, database=None)
        if not os.environ.get("ORB_HYPOTHESIS_RANDOM"):       # set it to explore fresh examples locally
            hypothesis.settings.load_profile("repro")
    except ImportError:
        pass


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name has a hyphen, so import it by string)."""
    return importlib.import_module("orb-slam3_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("orb-slam3_amd.synth")


@pytest.fixture(scope="session")
def oracle():
    import orbref
    orbref.lib()
    return orbref
