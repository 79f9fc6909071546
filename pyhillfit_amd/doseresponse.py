"""Host-side mirror of the reference's model library interface (python/doseresponse.py), GPU-backed.

Same names, argument meaning and return values as the reference module for everything the sampling step
touches — ``setup``, ``list_drug_channel_options``, ``load_crumb_data``, ``define_model``, ``log_target``,
``log_data_likelihood``, ``log_priors``, ``compute_pi_bit_of_log_likelihood`` and the output-path helpers —
so the drivers (pyhillfit_amd/PyHillFit.py, PyHillTemp.py) and the parity tests read like the reference's.
The arithmetic itself is NOT here: ``log_target`` & co. pack their arguments and call the HIP batch
evaluator (phf_single_level_log_target); there is no CPU implementation to fall back to.

Module-level state (``df``-like table, ``drugs``, ``channels``, ``num_params`` ...) mirrors the reference's
globals (doseresponse.py:31-37,250-279) because the drivers and downstream scripts read them as ``dr.xxx``.
"""
import csv
import json
import os

import numpy as np

# ---- constants: python/doseresponse.py:8-28 ------------------------------------------------------------------
sigma_uniform_lower = 1e-3
pic50_exp_rate = 0.2
pic50_exp_scale = 1. / pic50_exp_rate
pic50_exp_lower = -3.
hill_uniform_lower = 0.
hill_uniform_upper = 10.
sigma_shape = 5.
sigma_mode = 6.
sigma_loc = 1e-3
sigma_scale = scales = (sigma_mode - sigma_loc) / (sigma_shape - 1.)
n = 40      # temperature ladder: (i/n)**c, i = 0..n   (doseresponse.py:27-28, PyHillTemp.py:151)
c = 3

# ---- module state, set by setup() / define_model() -----------------------------------------------------------
file_name = None
dir_name = None
table = None          # columnar rows (the reference keeps a pandas DataFrame `df`)
drugs = None
channels = None
num_params = None
model_number = None
file_labels = None
labels = None


class Table(object):
    """Rows of a PyHillFit input file: Compound,Channel,Experiment,Dose,Response (data/readme.md)."""

    def __init__(self, drug, channel, experiment, dose, response):
        self.drug = np.asarray(drug, dtype=object)
        self.channel = np.asarray(channel, dtype=object)
        self.experiment = np.asarray(experiment, dtype=np.int64)
        self.dose = np.asarray(dose, dtype=np.float64)
        self.response = np.asarray(response, dtype=np.float64)

    @staticmethod
    def _unique(a):
        seen, out = set(), []
        for x in a:
            if x not in seen:
                seen.add(x); out.append(x)
        return np.array(out, dtype=object)

    @classmethod
    def from_csv(cls, path):
        cols = [[], [], [], [], []]
        with open(path, newline="") as f:
            rd = csv.reader(f)
            next(rd)                                  # header line (doseresponse.py:35 skiprows=1)
            for row in rd:
                if not row:
                    continue
                cols[0].append(row[0]); cols[1].append(row[1]); cols[2].append(int(float(row[2])))
                cols[3].append(float(row[3])); cols[4].append(float(row[4]) if row[4].strip() else np.nan)
        return cls(*cols)

    @classmethod
    def from_packed_json(cls, path):
        """data/*.json: the same rows, columnar (written by tests/golden/make_golden.py)."""
        with open(path) as f:
            t = json.load(f)
        return cls([t["drugs"][i] for i in t["drug_idx"]], [t["channels"][i] for i in t["channel_idx"]],
                   t["experiment"], t["dose"], t["response"])

    def to_csv(self, path):
        with open(path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Compound", "Channel", "Experiment", "Dose", "Response"])
            for k in range(len(self.dose)):
                w.writerow([self.drug[k], self.channel[k], int(self.experiment[k]), repr(float(self.dose[k])),
                            repr(float(self.response[k]))])


def setup(given_file):
    """doseresponse.py:31-37 — read the data file, set file_name/dir_name/drugs/channels."""
    global file_name, dir_name, table, drugs, channels
    file_name = given_file
    dir_name = given_file.split('/')[-1][:-4] if not given_file.endswith(".json") else given_file.split('/')[-1][:-5]
    table = Table.from_packed_json(given_file) if given_file.endswith(".json") else Table.from_csv(given_file)
    drugs = Table._unique(table.drug)
    channels = Table._unique(table.channel)


def setup_from_table(given_file, tbl):
    """setup() from rows already in memory (multi-GPU runs: rank 0 reads the file, the others receive the table)"""
    global file_name, dir_name, table, drugs, channels
    file_name = given_file
    dir_name = given_file.split('/')[-1][:-4] if not given_file.endswith(".json") else given_file.split('/')[-1][:-5]
    table = tbl
    drugs = Table._unique(table.drug)
    channels = Table._unique(table.channel)


def list_drug_channel_options(args_all):
    """doseresponse.py:40-57 — everything with -a, otherwise the interactive menu."""
    if args_all:
        return drugs, channels
    print("\nDrugs:\n")
    for i in range(len(drugs)):
        print("{}. {}".format(i + 1, drugs[i]))
    drug_indices = [x - 1 for x in map(int, input("\nSelect drug numbers: ").split())]
    assert 0 <= len(drug_indices) <= len(drugs)
    print("\nChannels:\n")
    for i in range(len(channels)):
        print("{}. {}".format(i + 1, channels[i]))
    channel_indices = [x - 1 for x in map(int, input("\nSelect channel numbers: ").split())]
    assert 0 <= len(channel_indices) <= len(channels)
    return [drugs[i] for i in drug_indices], [channels[i] for i in channel_indices]


def load_crumb_data(drug, channel):
    """doseresponse.py:60-67 — (num_expts = largest experiment label, zero-based labels, [(n_i,2) arrays])."""
    sel = (table.drug == drug) & (table.channel == channel)
    experiment_numbers = np.array(Table._unique(table.experiment[sel]), dtype=np.int64)
    if len(experiment_numbers) == 0:
        raise ValueError("no rows for {} + {}".format(drug, channel))
    num_expts = max(experiment_numbers)
    experiments = []
    for expt in experiment_numbers:
        m = sel & (table.experiment == expt)
        experiments.append(np.column_stack([table.dose[m], table.response[m]]))
    experiment_numbers = experiment_numbers - 1
    return num_expts, experiment_numbers, experiments


def concatenate_experiments(num_expts, experiments):
    """PyHillFit.py:661-665 / PyHillTemp.py:132-136."""
    concs = np.concatenate([experiments[i][:, 0] for i in range(num_expts)])
    responses = np.concatenate([experiments[i][:, 1] for i in range(num_expts)])
    return concs, responses


def response_masks(responses):
    """PyHillFit.py:675-677 — exact floating-point equality, like the reference."""
    responses = np.asarray(responses)
    return responses == 0, responses == 100, (0 < responses) & (responses < 100)


def compute_pi_bit_of_log_likelihood(y):
    """doseresponse.py:299-301."""
    return 0.5 * len(y) * np.log(2 * np.pi)


def pic50_to_ic50(pic50):
    """doseresponse.py:87-88 (host convenience; the kernels work in log space)."""
    return 10 ** (6 - pic50)


def ic50_to_pic50(ic50):
    """doseresponse.py:90-91."""
    return 6 - np.log10(ic50)


def trapezium_rule(x, y):
    """doseresponse.py:192-193."""
    x, y = np.asarray(x), np.asarray(y)
    return 0.5 * np.sum((x[1:] - x[:-1]) * (y[1:] + y[:-1]))


def temperature_ladder(rungs_n=None, power=None):
    """PyHillTemp.py:151."""
    nn = n if rungs_n is None else rungs_n
    cc = c if power is None else power
    return (np.arange(nn + 1.) / nn) ** cc


def define_model(model):
    """doseresponse.py:250-279 — choose Hill fixed to 1 (#1) or free (#2)."""
    global num_params, model_number, file_labels, labels
    if model == 1:
        num_params, labels, file_labels = 2, [r"$pIC50$", r"$\sigma$"], ['pIC50', 'sigma']
    elif model == 2:
        num_params, labels, file_labels = 3, [r"$pIC50$", r"$Hill$", r"$\sigma$"], ['pIC50', 'Hill', 'sigma']
    else:
        raise ValueError("model must be 1 or 2")
    model_number = model


# ---- packing for the kernels ----------------------------------------------------------------------------------
def merge_replicates(concs, y):
    """Entries of one masked group: the points that share a concentration (exact equality, first-appearance order)
    become (conc, mean response, count); also returns sum over groups of sum_j (y_j - mean)^2, the part of the sum of
    squares that does not depend on the curve (see include/pyhillfit_amd.h, phf_points).  Correctly rounded (fsum)."""
    import math
    groups, order = {}, []
    for cval, yval in zip(concs, y):
        k = float(cval)
        if k not in groups:
            groups[k] = []; order.append(k)
        groups[k].append(float(yval))
    cc, yy, ww, within = [], [], [], []
    for k in order:
        v = groups[k]
        m = v[0] if len(v) == 1 else math.fsum(v) / len(v)
        cc.append(k); yy.append(m); ww.append(float(len(v)))
        if len(v) > 1:
            within.extend((a - m) * (a - m) for a in v)
    return np.array(cc), np.array(yy), np.array(ww), math.fsum(within)


class PackedPoints(object):
    """numpy image of include/pyhillfit_amd.h `phf_points` for a list of pairs."""

    def __init__(self, pairs, merge=True):
        """pairs: list of (concs, responses) arrays, one per (drug, channel) pair, file order.
        merge: replicate points at one concentration become one weighted entry (same likelihood, less arithmetic)."""
        self.num_pairs = len(pairs)
        groups = []
        for concs, y in pairs:
            concs = np.asarray(concs, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
            if concs.shape != y.shape:
                raise ValueError("concs/responses length mismatch")
            is0, is100, other = response_masks(y)
            parts, ss_within = [], 0.0
            for mask in (other, is0, is100):
                if merge:
                    cc, yy, ww, ss = merge_replicates(concs[mask], y[mask])
                    ss_within += ss
                else:
                    cc, yy, ww = concs[mask], y[mask], np.ones(int(mask.sum()))
                parts.append((cc, yy, ww))
            groups.append((np.concatenate([p_[0] for p_ in parts]), np.concatenate([p_[1] for p_ in parts]),
                           np.concatenate([p_[2] for p_ in parts]), [len(p_[0]) for p_ in parts], int(other.sum()), ss_within, len(y)))
        self.stride = max(1, max(len(g[0]) for g in groups))
        self.ln_conc = np.zeros((self.num_pairs, self.stride))
        self.response = np.zeros((self.num_pairs, self.stride))
        self.weight = np.zeros((self.num_pairs, self.stride))
        self.counts = np.zeros((self.num_pairs, 4), dtype=np.int32)
        self.pi_bit = np.zeros(self.num_pairs)
        self.extra = np.zeros((self.num_pairs, 2))
        for p, (cc, yy, ww, ks, n_other_points, ss_within, ntot) in enumerate(groups):
            with np.errstate(divide="ignore"):
                self.ln_conc[p, :len(cc)] = np.log(cc)
            self.response[p, :len(yy)] = yy
            self.weight[p, :len(ww)] = ww
            self.counts[p] = (ks[0], ks[1], ks[2], ntot)
            self.pi_bit[p] = 0.5 * ntot * np.log(2 * np.pi)
            self.extra[p] = (float(n_other_points), ss_within)


def pack_single_level(drug_channel_pairs, merge=True):
    """[(drug, channel), ...] of the loaded table -> PackedPoints (PyHillFit.py:654-683 per pair)."""
    out = []
    for drug, channel in drug_channel_pairs:
        num_expts, _, experiments = load_crumb_data(drug, channel)
        out.append(concatenate_experiments(num_expts, experiments))
    return PackedPoints(out, merge=merge)


# ---- GPU-evaluated model functions with the reference's signatures ---------------------------------------------
def _eval_on_gpu(y, where_y_0, where_y_100, where_y_other, concs, params, t, pi_bit, want):
    from . import sampler
    y = np.asarray(y, dtype=np.float64)
    # rebuild responses consistent with the masks the caller passes (the reference trusts the masks, :244-247)
    yy = np.where(where_y_0, 0.0, np.where(where_y_100, 100.0, np.where(where_y_other, y, -1.0)))
    pts = PackedPoints([(np.asarray(concs, dtype=np.float64), yy)])
    pts.pi_bit[0] = pi_bit
    params = np.atleast_2d(np.asarray(params, dtype=np.float64))
    lik, pri = sampler.log_target_batch(pts, model_number, np.zeros(len(params), dtype=np.int32),
                                        np.full(len(params), float(t)), params)
    res = {"lik": lik, "prior": pri, "target": lik + pri}[want]
    return float(res[0]) if res.shape[0] == 1 else res


def log_data_likelihood(y, where_y_0, where_y_100, where_y_other, concs, params, t, pi_bit):
    """doseresponse.py:203-248 (selected by define_model)."""
    return _eval_on_gpu(y, where_y_0, where_y_100, where_y_other, concs, params, t, pi_bit, "lik")


def log_priors(params):
    """doseresponse.py:166-184."""
    d = np.zeros(1)
    return _eval_on_gpu(d, d == 1, d == 1, d == 1, np.ones(1), params, 0, 0.0, "prior")


def log_target(y, where_y_0, where_y_100, where_y_other, concs, params, t, pi_bit):
    """doseresponse.py:187-189."""
    return _eval_on_gpu(y, where_y_0, where_y_100, where_y_other, concs, params, t, pi_bit, "target")


# ---- output-path helpers (the chain-file contract): doseresponse.py:70-82,101-141 ----------------------------
output_root = "output"


def py2_str(x):
    """How the reference's Python 2 renders a number inside '{}'.format(...) (12 significant digits for floats):
    the temperature in the chain paths is formatted this way (doseresponse.py:120,127), e.g. 1, 0.0, 1.5625e-05."""
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    s = '%.12g' % float(x)
    if not any(ch in s for ch in '.en'):
        s += '.0'
    return s


def _clean(name):
    return name.replace('/', '_') if '/' in name else name


def _mk(*dirs):
    for d in dirs:
        os.makedirs(d, exist_ok=True)       # ranks of one run share leaf directories: no exists()-then-create race


def hierarchical_output_dirs_and_chain_file(drug, channel, Ne=0):
    """doseresponse.py:70-82."""
    drug, channel = _clean(drug), _clean(channel)
    output_dir = '{}/{}/hierarchical/{}/{}/{}_expts/'.format(output_root, dir_name, drug, channel, Ne)
    chain_dir, figs_dir = output_dir + 'chain/', output_dir + 'figures/'
    _mk(output_dir, chain_dir, figs_dir)
    chain_file = chain_dir + '{}_{}_{}_hierarchical_chain.txt'.format(dir_name, drug, channel)
    return drug, channel, output_dir, chain_dir, figs_dir, chain_file


def nonhierarchical_chain_file_and_figs_dir(model, drug, channel, temperature, make_dirs=True):
    """doseresponse.py:115-128.  (make_dirs=False: only name the files — rank 0 lists the chain files other ranks wrote)"""
    drug, channel = _clean(drug), _clean(channel)
    temperature = py2_str(temperature)
    output_dir = '{}/{}/single-level/{}/{}/model_{}/temperature_{}/'.format(output_root, dir_name, drug, channel, model, temperature)
    chain_dir, images_dir = output_dir + 'chain/', output_dir + 'figures/'
    if make_dirs:
        _mk(output_dir, chain_dir, images_dir)
    chain_file = chain_dir + '{}_{}_model_{}_temp_{}_chain_single-level.txt'.format(drug, channel, model, temperature)
    return drug, channel, chain_file, images_dir


def hierarchical_posterior_predictive_cdf_files(drug, channel, Ne):
    """doseresponse.py:93-99 (names as cleaned by hierarchical_output_dirs_and_chain_file)."""
    cdf_dir = '{}/{}/hierarchical/{}/{}/{}_expts/cdfs/'.format(output_root, dir_name, drug, channel, Ne)
    _mk(cdf_dir)
    return (cdf_dir + '{}_{}_{}_posterior_predictive_hill_cdf.txt'.format(dir_name, drug, channel),
            cdf_dir + '{}_{}_{}_posterior_predictive_pic50_cdf.txt'.format(dir_name, drug, channel))


def hierarchical_hill_and_pic50_samples_for_AP_file(drug, channel):
    """doseresponse.py:101-106."""
    output_dir = '{}/{}/hierarchical/posterior_predictive_hill_pic50_samples/'.format(output_root, dir_name)
    _mk(output_dir)
    return output_dir + '{}_{}_{}_hill_pic50_samples.txt'.format(dir_name, drug, channel)


def alpha_mu_downsampling(drug, channel):
    """doseresponse.py:130-135."""
    output_dir = '{}/{}/hierarchical/alpha_mu_samples/'.format(output_root, dir_name)
    _mk(output_dir)
    return output_dir + '{}_{}_hill_pic50_samples.txt'.format(drug, channel)
