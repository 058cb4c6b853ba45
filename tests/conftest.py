import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def jfk_pcm():
    """The reference's jfk.wav fixture (Tests/WhisperKitTests/Resources/jfk.wav) as float32 [-1, 1)."""
    return np.load(os.path.join(GOLDEN, "jfk_pcm16.npz"))["pcm16"].astype(np.float32) / 32768.0


def golden(name):
    return np.load(os.path.join(GOLDEN, name))
