"""Noisy observations from a sample path: counterpart of auxiliary/collect_obs.py:8.

Host code on purpose: the work is picking M rows of the path and adding M (or d x M) random
numbers, and the random numbers have to come from ``numpy.random`` in the reference's order
for a seeded run to give the reference's observations.
"""
import numpy as np

__all__ = ["collect_obs"]


def collect_obs(xt, T, n_Obs=1, R=0.2, hMask=None):
    """Observation indices and values at equidistant times (auxiliary/collect_obs.py:8-92).

    xt (N, D) sample path, T (N,) time window, n_Obs observations per time unit, R noise
    variance (scalar), variances (d,) or diagonal covariance (d, d), hMask the observed
    components.  Returns (obsX, obsY).  As in the reference, a scalar R adds ONE random
    number per observation time to all its components (``randn(M, 1)``, :74), a matrix R
    independent ones (``randn(d, M)``, :86)."""
    dt = np.diff(T)[0]
    if n_Obs > 1.0 / dt:
        raise ValueError('Observation density exceeds the number of samples.')
    M = int(np.floor(np.abs(T[0] - T[-1]) * n_Obs))
    N = T.shape[0]
    idx = np.linspace(0, N, M + 2, dtype=int)
    obsX = np.array(idx[1:-1].tolist())
    obsY = np.take(xt, obsX, axis=0)
    if hMask is not None:
        obsY = obsY[:, hMask]
    d = obsY.shape[1]
    if np.isscalar(R):
        obsY = obsY + np.sqrt(R) * np.random.randn(M, 1)
    else:
        R = np.asarray(R)
        if R.ndim == 1:
            R = np.diag(R)
        obsY = obsY + np.sqrt(R).dot(np.random.randn(d, M)).T
    return obsX, obsY
