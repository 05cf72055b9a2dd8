import json
import os
import sys

import pytest

# The pool's lanes overlap on GPU_MAX_HW_QUEUES hardware queues, which the ROCm runtime reads at the process's FIRST HIP call:
# it must be in the environment before any test touches torch / HIP (bpgpu_pool_create probes the device and refuses otherwise).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "py")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "rangeproof_v1.json")) as f:
        g = json.load(f)
    g["vc_bytes"] = b"".join(bytes.fromhex(v) for v in g["value_commitments"])
    g["label"] = g["transcript_label"].encode()
    return g


@pytest.fixture(scope="session")
def oracle():
    import pyoracle
    pyoracle.build()
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def oracle_gens_64_8(oracle):
    return oracle.Gens(64, 8)
