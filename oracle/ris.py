"""ORACLE -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the RIS weight formulas demonstrated by the reference's notebook
(restir_di/RIS_Test/ris_test.ipynb, cell 3 `sampleRIS`, raw file lines 67-104), written against
the formulas (not the notebook text):

    candidates x_i ~ p_i  (first half from p1, second half from p2)
    w_i = f(x_i) / p_i(x_i)
    pick y = x_k with probability w_k / sum(w)
    W_biased = sum(w) / (M f(y))                       <- what Reservoir-based ReSTIR calls
                                                          recPDFEstimate with weight 1/M
                                                          (optix_restir_di_kernels.cu:118, 268)
    W_naive  = sum(w) / (|{i : p_i(y) > 0}| f(y))      <- "naive unbiased" (useMIS_RIS = false)
    W_mis    = p_k(y) / sum_i p_i(y) * sum(w) / f(y)   <- MIS weights (useMIS_RIS = true, :210-258)

The proposal pairs: A: p1 = U[0,1), p2 = U[0,0.5);  B: p2 = 1.998 on [0,0.5), 0.002 on [0.5,1).
"""
import numpy as np


def target_density(x):
    return 2.0 - 2.0 * x


class ProposalPair:
    def __init__(self, name):
        assert name in ("A", "B")
        self.name = name

    def sample_p1(self, u):
        return u

    def p1(self, x):
        return np.ones_like(x)

    def sample_p2(self, u):
        if self.name == "A":
            return 0.5 * u
        return np.where(u < 0.999, 0.5 * (u / 0.999), 0.5 + 0.5 * (u - 0.999) / 0.001)

    def p2(self, x):
        if self.name == "A":
            return np.where(x < 0.5, 2.0, 0.0)
        return np.where(x < 0.5, 1.998, 0.002)


def ris_weights(pair, us, indices):
    """us: (M, K) uniforms; indices: (K,) the resampled candidate index per sample."""
    M, K = us.shape
    half = M // 2
    cand = np.concatenate([pair.sample_p1(us[:half]), pair.sample_p2(us[half:])], axis=0)
    with np.errstate(divide="ignore", invalid="ignore"):
        w = np.concatenate([target_density(cand[:half]) / pair.p1(cand[:half]),
                            target_density(cand[half:]) / pair.p2(cand[half:])], axis=0)
    y = cand[indices, np.arange(K)]
    f = target_density(y)
    sum_w = np.sum(w, axis=0)
    biased = sum_w / (M * f)
    nz = np.where(pair.p1(y) > 0, half, 0) + np.where(pair.p2(y) > 0, half, 0)
    naive = sum_w / (nz * f)
    denom = half * pair.p1(y) + half * pair.p2(y)
    mis_w = np.where(indices < half, pair.p1(y), pair.p2(y)) / denom
    mis = mis_w * sum_w / f
    return cand, w, y, biased, naive, mis
