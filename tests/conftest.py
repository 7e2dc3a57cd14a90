import json
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


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) when collected on a machine without a device and no -m filter
    # device_count() does not initialise the GPU (is_available() does): tests/test_a_dist_gpu.py starts child processes first
    if torch.cuda.device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def layer_kats():
    return np.load(os.path.join(GOLDEN, "layer_kats.npz"))


@pytest.fixture(scope="session")
def net_kats():
    return np.load(os.path.join(GOLDEN, "whole_net.npz"))


@pytest.fixture(scope="session")
def net_meta():
    with open(os.path.join(GOLDEN, "whole_net.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def dv_kats():
    """Dice-loss known answers and whole steps of the v2 net / --useDice (make_golden.py dice_v2)."""
    return np.load(os.path.join(GOLDEN, "dice_v2.npz"))


@pytest.fixture(scope="session")
def d1_kats():
    """DiceLoss single-class branch known answers (make_golden.py dice1)."""
    return np.load(os.path.join(GOLDEN, "dice1.npz"))


@pytest.fixture(scope="session")
def dv_meta():
    with open(os.path.join(GOLDEN, "dice_v2.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def pb_kats():
    """PB_FCN / trainer.py path (make_golden.py pbfcn)."""
    return np.load(os.path.join(GOLDEN, "pbfcn.npz"))


@pytest.fixture(scope="session")
def pb_meta():
    with open(os.path.join(GOLDEN, "pbfcn.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden(net_kats, net_meta, dv_kats, dv_meta):
    """tag -> (arrays, meta) across both whole-net fixture files."""
    def lookup(tag):
        return (net_kats, net_meta[tag]) if tag in net_meta else (dv_kats, dv_meta[tag])
    return lookup


def sd_hash(sd):
    import hashlib
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:16]
