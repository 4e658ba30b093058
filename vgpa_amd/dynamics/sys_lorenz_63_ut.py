"""Expected SDE energy for the Lorenz-63 with unscented-transform expectations.

Counterpart of ``energy_mode`` in dynamics/sys_lorenz_63_ut.py:132 and of ``system_path`` (dynamics/sys_lorenz_63_ut.py:49);
plotting is outside this package.
"""
import numpy as np

from .. import _lib
from . import _path
from ._energy import energy_mode as _energy_mode

__all__ = ["system_path", "energy_mode"]


def system_path(T, sig0, thet0):
    """A realisation of the stochastic Lorenz-63 system on the time window T -> (N, 3)
    (dynamics/sys_lorenz_63_ut.py:49-113): 8000 deterministic burn-in steps of 1e-3 from
    (1, 1, 1), then Euler-Maruyama with ``np.random.randn(N, 3)``."""
    print('Lorenz 1963 - trajectory')
    dt = np.diff(T)[0]
    N = len(T)
    kdiag = _path.noise_scale(sig0, dt, 3)
    ek = np.random.randn(N, 3)
    return _path.run(_lib.SYS_L63_UT, np.ones(3), thet0, kdiag, ek, dt, burn=8000, dtau=1.0e-3)


def energy_mode(A, b, m, S, sDyn):
    """(Esde, Ef, Edf, dEsde_dm, dEsde_dS, dEsde_dth, dEsde_dSig) for the variational
    parameters A, b and marginal moments m, S; ``sDyn`` needs 'dt', 'Sig', 'theta' and
    'obsX' (the time integrals stop at the last observation, auxiliary/numerical.py:275-281)."""
    return _energy_mode(_lib.SYS_L63_UT, A, b, m, S, sDyn)


energy_mode._vgpa_system = _lib.SYS_L63_UT
