import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = "/root/reference"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def yf():
    return importlib.import_module("stm32h7-yolo_amd")


@pytest.fixture(scope="session")
def golden():
    import json
    x = np.fromfile(os.path.join(GOLDEN, "golden_inputs.bin"), np.int8).reshape(-1, 56, 56, 3)
    heads = np.fromfile(os.path.join(GOLDEN, "golden_heads.bin"), np.int8).reshape(-1, 7, 7, 18)
    dump0 = np.fromfile(os.path.join(GOLDEN, "golden_dump_frame0.bin"), np.int8)
    meta = json.load(open(os.path.join(GOLDEN, "golden_meta.json")))
    return dict(inputs=x, heads=heads, dump0=dump0, meta=meta)


@pytest.fixture(scope="session")
def network(yf):
    """The library's single network instance, initialised on cuda:0 (GPU tests only)."""
    import torch
    torch.cuda.is_available()       # let torch bring up its HIP context first: asked only after the library has initialised the
                                    # GPU in this process, torch reports no device (seen when a single test file is run alone)
    net = yf.Network(device=0).init()
    yield net
    net.destroy()


def has_reference():
    return os.path.isdir(REFERENCE)
