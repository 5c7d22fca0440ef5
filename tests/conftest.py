import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    """oracle/liboracle.so -- the CPU restatement (checker only)."""
    path = os.path.join(ROOT, "oracle", "liboracle.so")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    return ctypes.CDLL(path)


@pytest.fixture(scope="session")
def cqm():
    with np.load(os.path.join(GOLDEN, "cqm_flat.npz")) as z:
        return {k: z[k] for k in z.files}


def load_golden(seed, interlaced):
    with np.load(os.path.join(GOLDEN, "l1_seed%d_i%d.npz" % (seed, interlaced))) as z:
        ins = {k[3:]: z[k] for k in z.files if k.startswith("in.")}
        outs = {k[4:]: z[k] for k in z.files if k.startswith("out.")}
    return ins, outs


@pytest.fixture(scope="session")
def hip_lib():
    """The product library through its C ABI; fails loudly if it is missing."""
    from x264_vs2008_amd import lib as L
    return L.load()


@pytest.fixture(scope="session")
def hip_lib_host():
    """The product library opened WITHOUT x264hip_init: only its host-side entries may be called (no GPU needed)."""
    from x264_vs2008_amd import lib as L
    return L.open_library()
