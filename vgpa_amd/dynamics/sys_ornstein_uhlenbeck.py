"""Expected SDE energy for the Ornstein-Uhlenbeck (1-D), f(x) = -theta x.

Counterpart of ``energy_mode`` in dynamics/sys_ornstein_uhlenbeck.py:83 and of ``system_path`` (dynamics/sys_ornstein_uhlenbeck.py:15);
plotting is outside this package.
"""
import numpy as np

from .. import _lib
from . import _path
from ._energy import energy_mode as _energy_mode

__all__ = ["system_path", "energy_mode"]


def system_path(T, sig0, thet0):
    """A realisation of the Ornstein-Uhlenbeck process (mean 0) on the time window T -> (N, 1)
    (dynamics/sys_ornstein_uhlenbeck.py:15-61), started at 0, with ``np.random.randn(N, 1)``."""
    print(" OU - trajectory\n")
    dt = np.diff(T)[0]
    N = len(T)
    ek = np.random.randn(N, 1)
    return _path.run(_lib.SYS_OU, np.zeros(1), thet0, _path.noise_scale(sig0, dt, 1), ek, dt)


def energy_mode(A, b, m, S, sDyn):
    """(Esde, Ef, Edf, dEsde_dm, dEsde_dS, dEsde_dth, dEsde_dSig) for the variational
    parameters A, b and marginal moments m, S; ``sDyn`` needs 'dt', 'Sig', 'theta' and
    'obsX' (the time integrals stop at the last observation, auxiliary/numerical.py:275-281)."""
    return _energy_mode(_lib.SYS_OU, A, b, m, S, sDyn)


energy_mode._vgpa_system = _lib.SYS_OU
