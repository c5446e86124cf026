import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def models():
    from oracle import load_models
    return load_models()


@pytest.fixture(scope="session")
def orc():
    from oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def ctx(models):
    """One device context for the whole GPU session; fails loudly (no CPU fallback) when there is no device."""
    import torch  # noqa: F401  (first: so that libnp_hip.so binds to the HIP runtime torch already loaded)
    from nanopolish_amd.api import Context
    c = Context(0)
    c.register_model(models["nucleotide"], "nucleotide")
    c.register_model(models["cpg"], "cpg")
    yield c
    c.close()
