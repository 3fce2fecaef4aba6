import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


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
