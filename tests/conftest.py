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
def golden():
    import json

    import numpy as np

    d = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(d, "ref_vectors.json")) as f:
        meta = json.load(f)
    return meta, np.load(os.path.join(d, "ref_vectors.npz"))


@pytest.fixture(scope="session")
def fake_rccl(tmp_path_factory):
    """tests/fake_rccl.c built as a shared library: the RCCL stand-in over POSIX shared memory that libmdct_hip.so binds through MDCT_RCCL_LIB"""
    import subprocess

    so = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.run(["gcc", "-O2", "-std=gnu11", "-Wall", "-Werror", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                    os.path.join(ROOT, "tests", "fake_rccl.c"), "-o", so, "-lrt", "-pthread"], check=True)
    return so
