import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_oracle():
    import fmoracle
    fmoracle.build(ref=True)
    yield


@pytest.fixture(autouse=True)
def _library_options_back_to_defaults():
    """a test that sets a library option (fm.options[...]) and fails before it puts it back does not leak it into the next test"""
    yield
    try:
        from fmindex_collection_amd import capi
        for name, value in capi.OPTION_DEFAULTS.items():
            capi.set_option(name, value)
    except Exception:
        pass
