"""Backward Lagrange-multiplier ODEs -- counterpart of core/bwd_ode.py:50 ``solver``."""
import numpy as np

from .. import engine
from . import _facade

__all__ = ["solver"]


def solver(A, dEsde_dm, dEsde_dS, dEobs_dm, dEobs_dS, dt=0.01, method="EULER"):
    """lam(t), Psi(t) integrated backwards from zero at t = N-1, with the
    observation jumps dEobs_dm[t-1], dEobs_dS[t-1] added after each step
    (core/bwd_ode.py:124-127).  Same method names, 1-D detection, output shapes
    and ValueError as core/bwd_ode.py:50-83."""
    mid = engine.method_id(method)
    A = np.asarray(A, dtype=np.float64)
    D = 1 if A.ndim == 1 else A.shape[1]
    Ad = _facade.dev_batch1(A, D, "m")
    gm, jm = _facade.dev_batch1(dEsde_dm, D, "v"), _facade.dev_batch1(dEobs_dm, D, "v")
    gS, jS = _facade.dev_batch1(dEsde_dS, D, "m"), _facade.dev_batch1(dEobs_dS, D, "m")
    sym = D > 1 and all(np.array_equal(x, np.swapaxes(x, -1, -2))
                        for x in (np.asarray(dEsde_dS), np.asarray(dEobs_dS)))
    lam, Psi = engine.bwd_scan(mid, Ad, gm, gS, jm, dt, dEobs_dS=jS, symmetric=sym)
    return _facade.host_vec(lam, D), _facade.host_mat(Psi, D)
