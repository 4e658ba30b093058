"""Expected SDE energy for the Lorenz-96 (D-dimensional) with unscented-transform expectations.

Counterpart of ``energy_mode`` in dynamics/sys_lorenz_96_ut.py:131 and of ``system_path`` (dynamics/sys_lorenz_96_ut.py:43);
plotting is outside this package.
"""
import numpy as np

from .. import _lib
from . import _path
from ._energy import energy_mode as _energy_mode

__all__ = ["system_path", "energy_mode"]


def system_path(T, D, sig0, thet0):
    """A realisation of the stochastic Lorenz-96 system on the time window T -> (N, D)
    (dynamics/sys_lorenz_96_ut.py:43-111): start at thet0 with component D/2 perturbed by
    1e-3, 50 000 deterministic burn-in steps of 1e-3, then Euler-Maruyama with
    ``np.random.randn(N, D)``."""
    print('Lorenz 1996 - trajectory')
    dt = np.diff(T)[0]
    N = len(T)
    x0 = np.array([thet0] * D, dtype='float64')
    x0[int(D / 2)] += 1.0e-3
    kdiag = _path.noise_scale(sig0, dt, D)
    ek = np.random.randn(N, D)
    return _path.run(_lib.SYS_L96_UT, x0, thet0, kdiag, ek, dt, burn=50000, dtau=1.0e-3)


def energy_mode(A, b, m, S, sDyn):
    """(Esde, Ef, Edf, dEsde_dm, dEsde_dS, dEsde_dth, dEsde_dSig) for the variational
    parameters A, b and marginal moments m, S; ``sDyn`` needs 'dt', 'Sig', 'theta' and
    'obsX' (the time integrals stop at the last observation, auxiliary/numerical.py:275-281)."""
    return _energy_mode(_lib.SYS_L96_UT, A, b, m, S, sDyn)


energy_mode._vgpa_system = _lib.SYS_L96_UT
