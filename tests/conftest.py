import ctypes
import os
import subprocess
import sys

import pytest

try:  # load order: PyTorch brings its own HIP runtime; importing it before libagarcl_hip.so keeps ONE runtime in the
    import torch  # noqa: F401   # process (a test that initialises torch.cuda after the engine would otherwise find no GPU)
except Exception:  # torch is only needed by the tensor-facing tests
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import orabind
    orabind.build()
    return orabind


@pytest.fixture(scope="session")
def ref_lib():
    """The real reference engine (oracle/_ref/libagar_ref.so); built here when /root/reference exists."""
    from oracle import refbind
    if os.path.isdir("/root/reference/agario"):
        subprocess.check_call(["make", "-s", "-f", os.path.join(ROOT, "oracle", "Makefile"), "ref"], cwd=os.path.join(ROOT, "oracle"))
    if not refbind.available():
        pytest.skip("reference build not available")
    return refbind


@pytest.fixture(scope="session")
def emu_lib():
    """TEST-ONLY host build of the kernel source (lanes as loops); never used by the product."""
    subprocess.check_call(["make", "-s", "-f", os.path.join(ROOT, "tests", "emu", "Makefile")], cwd=ROOT)
    from agarcl_amd import _capi
    return _capi.bind(ctypes.CDLL(os.path.join(ROOT, "tests", "_build", "libagarcl_emu.so")))


@pytest.fixture(scope="session")
def hip_engine_cls():
    """The product: agarcl_amd/libagarcl_hip.so through its C ABI.  Built in-tree first if the snapshot came without it
    (hipcc, ~1.5 min); a missing library or GPU makes every -m gpu test fail loudly -- there is nothing to fall back to."""
    from agarcl_amd import _capi, build as hip_build
    if not os.path.exists(_capi.HIP_SO):
        hip_build.build()
    _capi.hip_lib()
    return _capi.BatchedEngine
