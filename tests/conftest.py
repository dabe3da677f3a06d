import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu_index_factory():
    """Creates Index handles on cuda:0 and closes them at session end."""
    import search_engine_amd as se

    made = []

    def make():
        ix = se.Index(0)
        made.append(ix)
        return ix

    yield make
    for ix in made:
        ix.close()
