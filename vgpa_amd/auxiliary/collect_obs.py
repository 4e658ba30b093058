"""Noisy observations from a sample path: counterpart of auxiliary/collect_obs.py:8.

Host code on purpose: the work is picking M rows of the path and adding M (or d x M) random
numbers.  The only things taken over from the reference are what a seeded run depends on:
the integer observation grid and the shape and order of the ``numpy.random`` draws.
"""
import numpy as np

__all__ = ["collect_obs", "observation_grid", "observation_noise"]


def observation_grid(N, span, density):
    """Row indices of the observations: the interior points of an equidistant integer grid of
    floor(span * density) + 2 points over [0, N] (auxiliary/collect_obs.py:50-59).  The two end
    points, 0 and N, are never observed."""
    count = int(np.floor(span * density))
    return np.linspace(0, N, count + 2, dtype=int)[1:-1].copy()


def observation_noise(count, width, R):
    """(count, width) zero-mean Gaussian noise for `count` observations of `width` components.

    A plain number R is a variance shared by all components AND draws one random number per
    observation time, added to every component of that observation (``randn(count, 1)``,
    auxiliary/collect_obs.py:73-75).  An array R -- variances (width,) or a covariance
    (width, width) whose element-wise square root is applied, which is a factor of R only
    when R is diagonal (:78-87) -- draws independent numbers as ``randn(width, count)``."""
    if np.isscalar(R):
        return np.sqrt(R) * np.random.randn(count, 1)
    scale = np.sqrt(np.asarray(R, dtype=np.float64))
    draws = np.random.randn(width, count)
    # a vector of standard deviations scales the rows; a matrix multiplies (same numbers for a
    # diagonal one: the products with its zeros add nothing)
    shaped = scale @ draws if scale.ndim == 2 else scale.reshape(-1, 1) * draws
    return shaped.T


def collect_obs(xt, T, n_Obs=1, R=0.2, hMask=None):
    """Observation indices and values at equidistant times (auxiliary/collect_obs.py:8-92).

    xt (N, D) sample path (NumPy array or a tensor of this package's device paths), T (N,) time
    window, n_Obs observations per time unit, R noise variance (number), variances (d,) or
    diagonal covariance (d, d), hMask the observed components.  Returns (obsX, obsY)."""
    T = np.asarray(T, dtype=np.float64).reshape(-1)
    if n_Obs > 1.0 / (T[1] - T[0]):       # more observations than grid points per time unit
        raise ValueError('Observation density exceeds the number of samples.')
    obsX = observation_grid(T.size, abs(T[0] - T[-1]), n_Obs)
    path = xt.detach().cpu().numpy() if hasattr(xt, "detach") else np.asarray(xt)
    clean = path[obsX] if hMask is None else path[obsX][:, hMask]
    return obsX, clean + observation_noise(obsX.size, clean.shape[1], R)
