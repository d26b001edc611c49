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
def golden_pti():
    return _load("pti_trajectory.safetensors")


@pytest.fixture(scope="session")
def golden_pti_linear():
    return _load("pti_trajectory_linear.safetensors")


@pytest.fixture(scope="session")
def golden_operator_matrix():
    return _load("operator_matrix.safetensors")


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


def err_stats(a, b):
    """(whole-tensor relative L2, largest element error / rms(ref), worst ROW's relative L2).  The whole-tensor ratio
    alone lets one badly wrong edge-tile row or a handful of wrong elements through at 1e-3; the other two do not."""
    a, b = a.double().cpu(), b.double().cpu()
    d = a - b
    rms = b.pow(2).mean().sqrt().item() + 1e-30
    rel = (d.norm() / (b.norm() + 1e-30)).item()
    max_abs = d.abs().max().item() / rms if d.numel() else 0.0
    if b.dim() >= 2 and b.shape[-1] > 1:
        d2, b2 = d.reshape(-1, d.shape[-1]), b.reshape(-1, b.shape[-1])
        floor = 0.05 * rms * b2.shape[-1] ** 0.5  # rows that are ~zero are judged against the typical row
        worst = (d2.norm(dim=-1) / (b2.norm(dim=-1) + floor)).max().item() if d2.numel() else 0.0
    else:
        worst = rel
    return rel, max_abs, worst


def assert_close(got, ref, tol, what=""):
    """rel L2 < tol, every element within 8·tol of the reference's rms, every row within 4·tol relative."""
    rel, max_abs, worst = err_stats(got, ref)
    assert rel < tol, (what, "rel", rel)
    assert max_abs < 8 * tol, (what, "max_abs/rms", max_abs)
    assert worst < 4 * tol, (what, "worst_row", worst)


@pytest.fixture(scope="session")
def close():
    return assert_close


def build_tiny_unet(seed=0):
    from harness.unet import UNet2DConditionModel, tiny_config

    torch.manual_seed(seed)
    unet = UNet2DConditionModel(tiny_config(32, 32, 2))
    unet.requires_grad_(False)
    return unet


@pytest.fixture
def tiny_unet_factory():
    return build_tiny_unet
