import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the product library and the oracle exist (both build in seconds); import torch once here
    (the first import on a fresh box can take minutes while the image pages in -- not inside a test)."""
    import subprocess

    import torch  # noqa: F401

    if not os.path.exists(os.path.join(ROOT, "towr_amd", "libtowr_amd.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "towr_amd", "csrc")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "libtowr_oracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
