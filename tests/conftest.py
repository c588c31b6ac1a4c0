import json
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
def golden_index():
    return json.load(open(os.path.join(GOLDEN, "index.json")))


def load_case(name):
    return np.load(os.path.join(GOLDEN, f"case_{name}.npz"))


def load_state_dict(wname):
    return dict(np.load(os.path.join(GOLDEN, f"weights_{wname}.npz")))


@pytest.fixture(scope="session")
def oracle_weights():
    from llicti_amd.weights import pack_state_dict
    from oracle import oracle as orc
    cache = {}

    def get(wname):
        if wname not in cache:
            cache[wname] = orc.Weights(pack_state_dict(load_state_dict(wname)))
        return cache[wname]
    return get


CASES = ["noise_32x32_rand", "noise_67x93_rand", "smooth_64x48_tl", "smooth_67x93_tl", "noise_33x64_tl"]
