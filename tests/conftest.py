"""pytest configuration: paths, the `gpu` marker, and one-time builds of the native pieces."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fx8010-emulator-core_amd")
for p in (ROOT, os.path.join(PKG, "python"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the checkers and the product library are built in-tree; build whatever is missing
    if not os.path.exists(os.path.join(ROOT, "oracle", "libfxoracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "port"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libfxref.so")) and os.path.isdir("/root/reference/source"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"])
    if not os.path.exists(os.path.join(PKG, "libfx8010_amd.so")):
        subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc")])


@pytest.fixture(scope="session")
def amd():
    # torch first: a process that initialises HIP through libfx8010_amd.so before torch has loaded its own HIP runtime
    # leaves torch without devices ("No HIP GPUs are available"); the other order works (bench.py does the same)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    import fx8010_amd

    fx8010_amd.load()
    return fx8010_amd


@pytest.fixture(scope="session")
def gpu(amd):
    """The product library with a usable device; fails loudly (never skips to a CPU path)."""
    if amd.device_count() < 1:
        pytest.fail("no HIP device visible: -m gpu tests must run on the GPU box")
    return amd
