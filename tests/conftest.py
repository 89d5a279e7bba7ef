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
    """The library normally travels with the tree (or `__graft_entry__.build()` ran first).  If neither
    happened, compile it once here -- compile only, nothing touches a GPU -- so that collection does not
    fail on a missing .so; a missing hipcc still fails loudly."""
    so = os.path.join(ROOT, "rust-local-rag_amd", "librlr_gpu.so")
    if os.path.exists(so):
        return
    import fcntl

    with open(os.path.join(ROOT, "rust-local-rag_amd", ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        if not os.path.exists(so):
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
