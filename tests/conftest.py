import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _load(name):
    from safetensors import safe_open

    f = safe_open(os.path.join(GOLDEN, name), "pt")
    return {k: f.get_tensor(k) for k in f.keys()}, (f.metadata() or {})


@pytest.fixture(scope="session")
def golden_operator():
    return _load("operator.safetensors")


@pytest.fixture(scope="session")
def golden_losses():
    return _load("losses.safetensors")


@pytest.fixture(scope="session")
def golden_merge():
    return _load("merge.safetensors")


@pytest.fixture(scope="session")
def golden_trajectory():
    return _load("trajectory.safetensors")


@pytest.fixture(scope="session")
def golden_structure():
    with open(os.path.join(GOLDEN, "structure.json")) as f:
        return json.load(f)


def rel_err(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.fixture(scope="session")
def relerr():
    return rel_err


def build_tiny_unet(seed=0):
    from harness.unet import UNet2DConditionModel, tiny_config

    torch.manual_seed(seed)
    unet = UNet2DConditionModel(tiny_config(32, 32, 2))
    unet.requires_grad_(False)
    return unet


@pytest.fixture
def tiny_unet_factory():
    return build_tiny_unet
