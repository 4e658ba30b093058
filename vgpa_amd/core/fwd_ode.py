"""Forward moment ODEs -- counterpart of core/fwd_ode.py:48 ``solver``."""
import numpy as np

from .. import engine
from . import _facade

__all__ = ["solver"]


def solver(A, b, m0, S0, Sigma, dt=0.01, method="EULER"):
    """m(t), S(t) from (m0, S0) under dm/dt = -A m + b, dS/dt = -A S - S A' + Sigma
    with EULER / HEUN / RK2 / RK4 (case-insensitive; anything else raises
    ValueError('Unknown method of integration'), core/fwd_ode.py:82).  The 1-D path
    is taken when A is a vector or has a single column (core/fwd_ode.py:51-54) and
    returns (N,1) arrays; otherwise (N,D) and (N,D,D)."""
    mid = engine.method_id(method)
    A = np.asarray(A, dtype=np.float64)
    D = 1 if A.ndim == 1 else A.shape[1]
    N = np.asarray(b).shape[0]
    Ad, bd = _facade.dev_batch1(A, D, "m"), _facade.dev_batch1(b, D, "v")
    if D == 1:
        m0d = engine.to_dev(np.asarray(m0, dtype=np.float64).reshape(1, 1))
        S0d = engine.to_dev(np.asarray(S0, dtype=np.float64).reshape(1))
        Sg = engine.to_dev(np.asarray(Sigma, dtype=np.float64).reshape(1))
    else:
        m0d = engine.to_dev(np.asarray(m0, dtype=np.float64).reshape(1, D))
        S0d = engine.to_dev(np.asarray(S0, dtype=np.float64).reshape(1, D, D))
        Sg = engine.to_dev(np.asarray(Sigma, dtype=np.float64).reshape(D, D))
    m, S = engine.fwd_scan(mid, Ad, bd, m0d, S0d, Sg, dt)
    return _facade.host_vec(m, D), _facade.host_mat(S, D)
