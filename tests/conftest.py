import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
ASSETS = os.path.join(REPO, "gym_pybullet_drones_amd", "assets")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def urdf(model):
    return os.path.join(ASSETS, model + ".urdf")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    # the in-tree library normally travels with the snapshot; on a fresh checkout build it (hipcc is in the image)
    from gym_pybullet_drones_amd import _native
    if not os.path.exists(_native.LIB_PATH):
        _native.build(verbose=True)
    return torch.device("cuda:0")
