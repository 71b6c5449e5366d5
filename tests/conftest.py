import os
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "trex-gym_amd"))

ASSET_URDF = os.path.join(ROOT, "trex-gym_amd", "assets", "trex_collide.urdf")
REFERENCE_ROOT = "/root/reference"  # absent on the GPU box: only used by skip-if-missing tests


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def model():
    from oracle import trex_model
    return trex_model.compile_model(ASSET_URDF)


@pytest.fixture(scope="session")
def oracle64(model):
    from oracle import oracle as O
    return O.Oracle(model, precision="f64")


@pytest.fixture(scope="session")
def oracle32(model):
    from oracle import oracle as O
    return O.Oracle(model, precision="f32")
