"""Chain files: the output contract of the sampling step (python/PyHillFit.py:861-867,423-426,514-525;
python/PyHillTemp.py:165-169).  Text via np.savetxt's default '%.18e', '#' comment headers, so the reference's
downstream readers (np.loadtxt with usecols, last column = log-target) work unchanged."""
import numpy as np


def drop_burn_in(chain, burn_in_fraction):
    """PyHillFit.py:861-864 — Python-2 integer division: burn = saved_iterations / burn_fraction."""
    return chain[chain.shape[0] // int(burn_in_fraction):]


def save_single_level_chain(chain_file, chain, drug, channel, model):
    """PyHillFit.py:865-867.  Columns: model 2 -> pIC50, Hill, sigma, log-target; model 1 -> pIC50, sigma, log-target
    (the reference's header text says "(Hill,pIC50,sigma,log-target)"; its code writes the order used here)."""
    cols = "(pIC50,Hill,sigma,log-target)" if model == 2 else "(pIC50,sigma,log-target)"
    with open(chain_file, 'w') as outfile:
        outfile.write('# Nonhierarchical MCMC output for {} + {}: {}\n'.format(drug, channel, cols))
        np.savetxt(outfile, chain)


def save_tempered_chain(chain_file, chain):
    """PyHillTemp.py:169 — no header."""
    np.savetxt(chain_file, chain)


def save_hierarchical_chain(chain_file, chain):
    """PyHillFit.py:423-426,514-515 — two header lines, then the FULL chain (burn-in included).
    Parameter order as the code stores it: alpha, beta, mu, s, pIC50_1, Hill_1, ..., sigma, log-target."""
    with open(chain_file, 'w') as outfile:
        outfile.write("# Hill ~ log-logistic(alpha,beta), pIC50 ~ logistic(mu,s)\n")
        outfile.write("# alpha, beta, mu, s, pic50_1, hill_1, pic50_2, hill_2, ..., pic50_Ne, hill_Ne, sigma, log-target\n")
        np.savetxt(outfile, chain)


def save_alpha_mu_samples(samples_file, chain, num_samples, burn, drug, channel, rng):
    """PyHillFit.py:519-525 — num_samples random post-burn rows, columns (alpha, mu)."""
    indices = rng.randint(burn, chain.shape[0], num_samples)
    with open(samples_file, 'w') as outfile:
        outfile.write('# {} (alpha,mu) samples from hierarchical MCMC for {} + {}\n'.format(num_samples, drug, channel))
        np.savetxt(outfile, chain[indices][:, [0, 2]])


def save_best_fit_params(path, theta0, model):
    """PyHillFit.py:739-746 (read back by assemble_BFs.py:62-63 with np.loadtxt)."""
    with open(path, "w") as outfile:
        outfile.write("# least-squares best fit params (start point of the MCMC)\n")
        outfile.write("# pIC50, sigma, (Hill=1, not included)\n" if model == 1 else "# pIC50, Hill, sigma\n")
        np.savetxt(outfile, [theta0])
