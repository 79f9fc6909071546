"""SURVEY.md section 8(d)'s synthetic scaling set S3: P generated (drug, channel) pairs for the benchmark's scaling runs
(BASELINE.json north_star: "samples/sec on synthetic drug x channel x chain batches reported at 1, 2, 4 and 8 GPUs").

The reference ships its own small synthetic file (data/synthetic_data.csv, data/readme.md:6-9: responses of a Hill curve plus Gaussian
noise, clipped to the measurable range); S3 follows the same recipe at the Crumb set's shape — three experiments of four doses per
pair — with as many pairs as a node of GPUs needs: the reference's unit of parallel work is the pair (python/PyHillFit.py:978-1003),
and P = 1 680 pairs x 4 096 chains are 107 520 blocks of 64 chains: 13 440 per GPU at 8, no ragged last round.

    rng = numpy.random.default_rng(12345)
    pIC50 ~ U(3, 9), Hill ~ U(0.5, 2), sigma ~ U(2, 10)                       one draw per pair, in this order, each a vector of P
    doses  = IC50 * 10^linspace(-2, 2, 4)  (uM; IC50 = 10^(6 - pIC50))         the same four doses in the pair's three experiments
    y      = clip(100 (1 - 1 / (1 + (dose / IC50)^Hill)) + sigma z, 0, 100)     z ~ N(0, 1), one [P][3][4] array drawn last
so that about 30 % of the responses are censored (exactly 0 or exactly 100), the Crumb set's rate (768 + 15 of 2 585).
Deterministic: the same P gives the same data on every rank (no broadcast needed, though bench.py still routes it through one)."""
import numpy as np

SEED = 12345
NUM_EXPTS, DOSES_PER_EXPT = 3, 4


def generate(num_pairs, seed=SEED):
    """-> list of num_pairs entries, each a list of 3 experiments, each an array [4][2] of (concentration in uM, % inhibition):
    the shape pyhillfit_amd.doseresponse.load_crumb_data returns for a Crumb pair"""
    rng = np.random.default_rng(seed)
    P = int(num_pairs)
    pic50 = rng.uniform(3.0, 9.0, P)
    hill = rng.uniform(0.5, 2.0, P)
    sigma = rng.uniform(2.0, 10.0, P)
    z = rng.standard_normal((P, NUM_EXPTS, DOSES_PER_EXPT))
    ic50 = 10.0 ** (6.0 - pic50)
    doses = ic50[:, None] * 10.0 ** np.linspace(-2.0, 2.0, DOSES_PER_EXPT)[None, :]                  # [P][4]
    pred = 100.0 * (1.0 - 1.0 / (1.0 + (doses / ic50[:, None]) ** hill[:, None]))                   # doseresponse.py:84-85
    y = np.clip(pred[:, None, :] + sigma[:, None, None] * z, 0.0, 100.0)                             # [P][3][4]
    out = []
    for p in range(P):
        out.append([np.stack([doses[p], y[p, e]], axis=1) for e in range(NUM_EXPTS)])
    return out, {"pic50": pic50, "hill": hill, "sigma": sigma}


def single_level_pairs(experiments_per_pair):
    """[(concs, responses)] per pair, experiments concatenated in order (python/PyHillFit.py:661-677): what PackedPoints takes"""
    return [(np.concatenate([e[:, 0] for e in ex]), np.concatenate([e[:, 1] for e in ex])) for ex in experiments_per_pair]


def censoring_rate(experiments_per_pair):
    y = np.concatenate([e[:, 1] for ex in experiments_per_pair for e in ex])
    return float(np.mean((y == 0.0) | (y == 100.0)))
