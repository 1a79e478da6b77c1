import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def full_model(cuda):
    """The Stage-2 model of the headline bench at FULL juggernautXL size (UNet 2.6 B + ControlNet 1.2 B parameters, seeded
    random init, tiled VAE 512 / 64), built once per session (~30 s of host time) and shared by the full-size tests."""
    import bench
    return bench.build_stage2(cuda, True)
