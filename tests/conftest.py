import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (oracle/ -- test infrastructure)."""
    from oracle import pn2_oracle
    pn2_oracle.lib()
    return pn2_oracle


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))
        return cache[name]
    return load


@pytest.fixture()
def oracle_backend(oracle):
    """Run bridgeqa_amd's Python layers over the CPU oracle (host-logic tests without a GPU)."""
    from bridgeqa_amd import pointnet2_utils
    prev = pointnet2_utils.set_backend(oracle)
    yield oracle
    pointnet2_utils.set_backend(prev)


@pytest.fixture(scope="session")
def dev():
    assert torch.cuda.is_available(), "GPU test selected but no GPU visible"
    return torch.device("cuda:0")


def scene(B, N, C=0, seed=0, room=(8.0, 8.0, 3.0)):
    g = torch.Generator().manual_seed(seed)
    xyz = torch.rand(B, N, 3, generator=g) * torch.tensor(room)
    if C == 0:
        return xyz.contiguous()
    return torch.cat([xyz, torch.randn(B, N, C, generator=g)], -1).contiguous()
