import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import nunet_amd  # noqa: E402,F401  (registers the package alias)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE_TFLITE = "/root/reference/dnn_model/tflite/nutls_lstm.tflite"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")
    config.addinivalue_line("markers", "reference: needs /root/reference (build container only)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def has_reference():
    return os.path.exists(REFERENCE_TFLITE)
