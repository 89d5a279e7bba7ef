import importlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_built():
    """Bring librlr_gpu.so up to date with csrc/ and include/ before anything imports it: build.py recompiles only
    what is stale (a no-op when the shipped binary is fresh; compile only, nothing touches a GPU), so a source edit can
    never be tested against an old binary.  A missing hipcc fails loudly."""
    import fcntl

    with open(os.path.join(ROOT, "rust-local-rag_amd", ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        import __graft_entry__

        __graft_entry__._load_build_module().build()


_ensure_built()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_pkg():
    """The package directory name is not an identifier -> importlib."""
    return importlib.import_module("rust-local-rag_amd")


@pytest.fixture(scope="session")
def rlr():
    return load_pkg()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O  # test infrastructure: the checker
    O.lib()
    return O


@pytest.fixture(scope="session")
def kats():
    with open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")) as f:
        return json.load(f)


def pf(x):
    """JSON cannot carry NaN/Inf: the fixtures spell them as strings."""
    if isinstance(x, str):
        return float(x.replace("Infinity", "inf"))
    return float(x)


@pytest.fixture(scope="session")
def gpu_available(rlr):
    return rlr.device_count() > 0


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)
