"""TEST INFRASTRUCTURE (oracle/): exact posteriors of a 2-state Gaussian HMM by ENUMERATING every state path.

frisk_amd/hmm.py restates the model the reference asks of hmmlearn (frisk/__init__.py L1539-1541: GaussianHMM(n_components=2,
covariance_type="full").fit on the stacked KLD track; L769: predict per scaffold).  hmmlearn is absent here, so the model's
arithmetic cannot be pinned to the third party itself; what CAN be pinned is that forward-backward, the M step and Viterbi
compute what their definitions say.  For n <= 14 observations all 2^n paths are enumerated in extended precision (Python
floats via math.fsum): the likelihood is the sum of the path probabilities, a posterior is the share of the paths through a
state, the best path is the arg max - no recursion, no scaling, nothing shared with the code under test.  The closed-form EM
step on those posteriors (hmmlearn's defaults: flat Dirichlet priors, means_weight 0, covars_prior 1e-2, covars_weight 1) is
what one round of Baum-Welch must return.  Only tests/ may import this module.
"""
import itertools
import math


def _density(x, mean, covar):
    return math.exp(-0.5 * (math.log(2.0 * math.pi) + math.log(covar) + (x - mean) ** 2 / covar))


def enumerate_paths(x, means, covars, startprob, transmat):
    """{likelihood, loglik, gamma[t][i], xi[i][j] (summed over t), best_path, best_logp, runner_up_logp} by brute force."""
    n = len(x)
    assert 1 <= n <= 14
    dens = [[_density(x[t], means[i], covars[i]) for i in (0, 1)] for t in range(n)]
    probs = {}
    for path in itertools.product((0, 1), repeat=n):
        p = startprob[path[0]] * dens[0][path[0]]
        for t in range(1, n):
            p *= transmat[path[t - 1]][path[t]] * dens[t][path[t]]
        probs[path] = p
    like = math.fsum(probs.values())
    gamma = [[math.fsum(p for path, p in probs.items() if path[t] == i) / like for i in (0, 1)] for t in range(n)]
    xi = [[math.fsum(p for path, p in probs.items() for t in range(1, n) if path[t - 1] == i and path[t] == j) / like
           for j in (0, 1)] for i in (0, 1)]
    ranked = sorted(probs.items(), key=lambda kv: -kv[1])
    return {"likelihood": like, "loglik": math.log(like), "gamma": gamma, "xi": xi, "best_path": list(ranked[0][0]),
            "best_logp": math.log(ranked[0][1]) if ranked[0][1] > 0 else -math.inf,
            "runner_up_logp": math.log(ranked[1][1]) if len(ranked) > 1 and ranked[1][1] > 0 else -math.inf}


def em_step(x, means, covars, startprob, transmat, covars_prior=1e-2):
    """One exact Baum-Welch round from the enumerated posteriors: (means, covars, startprob, transmat, loglik of the input model)."""
    e = enumerate_paths(x, means, covars, startprob, transmat)
    n = len(x)
    g = e["gamma"]
    w = [math.fsum(g[t][i] for t in range(n)) for i in (0, 1)]
    new_means = [math.fsum(g[t][i] * x[t] for t in range(n)) / w[i] for i in (0, 1)]
    new_covars = [(covars_prior + math.fsum(g[t][i] * (x[t] - new_means[i]) ** 2 for t in range(n))) / w[i] for i in (0, 1)]
    new_start = [g[0][0] / (g[0][0] + g[0][1]), g[0][1] / (g[0][0] + g[0][1])]
    new_trans = []
    for i in (0, 1):
        row = e["xi"][i][0] + e["xi"][i][1]
        new_trans.append([e["xi"][i][0] / row, e["xi"][i][1] / row] if row > 0 else [0.5, 0.5])
    return new_means, new_covars, new_start, new_trans, e["loglik"]
