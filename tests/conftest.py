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


def reference_posteriors(model, temperature=1.0):
    """The reference sampler's long-run posterior of every Crumb pair: (names, mean [d+1][210], standard error [d+1][210], sd).

    G5c = ONE reference chain per pair (PyHillTemp.do_mcmc, 200 000 iterations, seed 1) with its batch-means standard error.  Where
    G5d holds several independent reference chains of the same case (tests/golden/make_golden_posteriors_reseed.py: the weakly
    informative pairs whose single chain is not enough), the pooled mean of those replaces the single chain's, with the larger
    of the two standard errors one can form from them: scatter between the seeds' means / sqrt(n), and mean batch-means s.e. / sqrt(n)."""
    with open(os.path.join(GOLDEN, "g5c_posteriors_all_pairs_model_%d.json" % model)) as f:
        g5c = json.load(f)
    names = [(w["drug"], w["channel"]) for w in g5c]
    mean = np.array([w["mean"] for w in g5c]).T
    se = np.array([w["batch_means_se"] for w in g5c]).T
    sd = np.array([w["sd"] for w in g5c]).T
    reseeded = []
    # G5d: the closed list of cases with 96 reference seeds each; G5e (round 4): the posterior-width follow-up, 32 seeds for each of the
    # twelve (pair, model) cases whose sd ratio against G5c's single chain lay outside [0.8, 1.25] (make_golden_posteriors_reseed.py --widths)
    # G5f (round 5): 256 fresh seeds for the two pairs G5e left outside the width band (--tails; rule written before the runs); read last,
    # it stands in for those two pairs' G5e entries
    for fixture in ("g5d_posteriors_reseeded.json", "g5e_posterior_widths_reseeded.json", "g5f_posterior_tails_reseeded.json"):
        path = os.path.join(GOLDEN, fixture)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            for e in json.load(f):
                if e["model"] == model and e["temperature"] == temperature and (e["drug"], e["channel"]) in names:
                    q = names.index((e["drug"], e["channel"]))
                    n = len(e["seeds"])
                    mean[:, q] = e["mean"]
                    se[:, q] = np.maximum(e["se_between_seeds"], np.array(e["se_single_chain_batch_means"]) / np.sqrt(n))
                    sd[:, q] = e["sd"]
                    if names[q] not in reseeded:
                        reseeded.append(names[q])
    return names, mean, se, sd, reseeded
