import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _torch_device_first(request):
    """GPU runs: bring up torch's device context before the first ptgpu call and before any test
    starts a child process on the GPU.  torch ships its own copy of the HIP runtime; bench.py and
    the multi-GPU launchers initialise it first, so the tests do the same, once, at a defined point
    instead of in the middle of the session."""
    if any(item.get_closest_marker("gpu") for item in request.session.items):
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
            torch.zeros(1, device="cuda")
            torch.cuda.synchronize()
    yield


@pytest.fixture(scope="session")
def port():
    """The plain-C restatement (oracle/cpu_ref.c); built on demand where gcc is present."""
    import _oracle

    if not os.path.exists(_oracle.PORT_SO):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, timeout=300)
    return _oracle.Port()


@pytest.fixture(scope="session")
def golden_scene():
    import _oracle

    def load(name):
        with open(os.path.join(GOLDEN, f"scene_{name}.json")) as f:
            d = json.load(f)
        bodies = _oracle.bodies_from_list(d["bodies"])
        cam = _oracle.camera_from_list(d["camera"])
        return bodies, len(d["bodies"]), cam

    return load


def same_bits(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))
