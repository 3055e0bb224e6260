import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """Our CPU restatement (test infrastructure); built on demand by __graft_entry__.build()."""
    import oracle_lib as ol

    if not ol.available("orc"):
        import subprocess

        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"])
    return ol.load("orc")


@pytest.fixture(scope="session")
def ref():
    """The real reference (only where oracle/_ref was built, i.e. where /root/reference exists)."""
    import oracle_lib as ol

    if not ol.available("ref"):
        pytest.skip("oracle/_ref/libref3dsift.so not built (no /root/reference on this machine)")
    return ol.load("ref")


@pytest.fixture(scope="session")
def synth():
    import importlib

    return importlib.import_module("3dsift_amd.synth")


def golden(name):
    import numpy as np

    return np.load(os.path.join(HERE, "golden", name), allow_pickle=False)
