import json
import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, "tests", "golden")
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_meta():
    with open(os.path.join(GOLDEN, "golden_meta.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def g4_pairs():
    with open(os.path.join(GOLDEN, "g4_pairs.json")) as f:
        return {(p["drug"], p["channel"]): p for p in json.load(f)["pairs"]}


def _load_table(name):
    with open(os.path.join(REPO, "data", name)) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def crumb_table():
    return _load_table("crumb_dataset.json")


@pytest.fixture(scope="session")
def synthetic_table():
    return _load_table("synthetic_dataset.json")


def rows_of(table, drug, channel):
    di, ci = table["drugs"].index(drug), table["channels"].index(channel)
    sel = [k for k in range(len(table["dose"])) if table["drug_idx"][k] == di and table["channel_idx"][k] == ci]
    return (np.array([table["experiment"][k] for k in sel]), np.array([table["dose"][k] for k in sel]),
            np.array([table["response"][k] for k in sel]))


@pytest.fixture(scope="session")
def oracle_pair(crumb_table, synthetic_table):
    """(file, drug, channel) -> oracle PairData built from the packed tables."""
    from oracle import pyhillfit_oracle as orc
    cache = {}

    def get(drug, channel, which="crumb"):
        key = (which, drug, channel)
        if key not in cache:
            table = crumb_table if which == "crumb" else synthetic_table
            cache[key] = orc.pair_from_rows(*rows_of(table, drug, channel))
        return cache[key]
    return get
