"""Sample paths on the device behind the reference's ``system_path`` signatures.

The random numbers are drawn on the host from ``numpy.random`` with the calls, shapes and
order of the reference (dynamics/sys_*.py), so a seeded run consumes the generator the same
way and returns the same path; the recursion itself (burn-in and Euler-Maruyama steps) runs
in ``vgpa_sde_path``.
"""
import numpy as np

from .. import _lib


def noise_scale(sig0, dt, D):
    """K = sqrt(sig0 dt) as a (D,) vector of per-component scales.  The reference forms
    ``ek[t].dot(K.T)`` (sys_lorenz_96_ut.py:106): a scalar K scales every component, a
    diagonal K its own component; a K with off-diagonal entries would mix components."""
    K = np.sqrt(np.asarray(sig0, dtype=np.float64) * dt)
    if K.ndim == 0:
        return np.full(D, float(K))
    if K.ndim == 2 and K.shape == (D, D):
        if np.count_nonzero(K - np.diag(np.diagonal(K))):
            raise ValueError("system_path: the noise (co)variance must be a scalar or a diagonal matrix")
        return np.array(np.diagonal(K), dtype=np.float64)
    raise ValueError(f"system_path: noise (co)variance of shape {K.shape} for D={D}")


def run(system, x0, theta, kdiag, ek, dt, burn=0, dtau=0.0, add0=None):
    """One path: x0 (D,), ek (N,D) -> X (N,D) as a NumPy array."""
    from .. import engine
    d = engine.to_dev
    x0 = np.asarray(x0, dtype=np.float64).reshape(1, -1)
    D = x0.shape[1]
    X = engine.sde_paths(system, d(x0), d(np.atleast_1d(np.asarray(theta, dtype=np.float64))), d(kdiag),
                         d(np.asarray(ek, dtype=np.float64).reshape(1, -1, D)), dt, burn, dtau,
                         None if add0 is None else d(np.asarray(add0, dtype=np.float64).reshape(1, D)))
    return X[0].cpu().numpy()

