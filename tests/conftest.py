import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """PyTorch ships its own HIP runtime.  When it and libc12381_hip.so (linked against /opt/rocm) live in one process the
    runtime that initialises first must be torch's — the other order leaves torch without devices.  Only the GPU tests
    that hand torch tensors to the library need torch at all, so initialise it up front exactly when GPU tests run."""
    if any(item.get_closest_marker("gpu") for item in items):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(scope="session")
def oracle_port():
    from oracle.bindings import Oracle, build
    build()
    return Oracle("port")


@pytest.fixture(scope="session")
def oracle_ref():
    from oracle.bindings import Oracle, have_reference, build
    build()
    if not have_reference():
        pytest.skip("oracle/_ref/libc12381_ref.so not present (needs /root/reference to build)")
    return Oracle("reference")
