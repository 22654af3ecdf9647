"""Synthetic workloads of SURVEY.md section 8(d) (host-side numpy, seeds only).

The recipe mirrors what ``BayesianModelCombination.orthogonalize`` feeds the
sampler (reference bmc.py:106-122, inference_utils.py:164): a gaussian model
matrix, rows centred, thin SVD, the leading left singular vectors as a
column-major design matrix, and the ``train()`` default prior
(reference bmc.py:168-171).
"""
import hashlib

import numpy as np


def synth_problem(n, k_models, kept, seed, noise=0.1, dtype=np.float64):
    rng = np.random.Generator(np.random.PCG64(seed))
    F = rng.standard_normal((n, k_models))
    Fc = F - F.mean(axis=1)[:, None]
    U, S, Vt = np.linalg.svd(Fc, full_matrices=False)
    X = np.asfortranarray(U[:, :kept])
    S_hat = S[:kept]
    beta_true = rng.standard_normal(kept)
    y = X @ beta_true + noise * rng.standard_normal(n)
    prior = (np.zeros(kept), np.diag(S_hat ** 2), 1.0, 0.02)
    if dtype != np.float64:
        X = np.asfortranarray(X.astype(dtype))
        y = y.astype(dtype)
    return dict(F=F, X=X, y=y, S_hat=S_hat, Vt=Vt[:kept], prior=prior,
                beta_true=beta_true)


def sha256(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
