"""Expected SDE energy for the double-well (1-D), f(x) = 4 x (theta - x^2).

Counterpart of ``energy_mode`` in dynamics/sys_double_well.py:91 and of ``system_path`` (dynamics/sys_double_well.py:14);
plotting is outside this package.
"""
import numpy as np

from .. import _lib
from . import _path
from ._energy import energy_mode as _energy_mode

__all__ = ["system_path", "energy_mode"]


def system_path(T, sig0, thet0):
    """A realisation of the double-well system on the time window T -> (N, 1)
    (dynamics/sys_double_well.py:14-75).  Like the reference it draws from a generator
    seeded with 6771 and restores the caller's generator state afterwards; the first
    value comes from the two-component equilibrium mixture."""
    print(" DW - trajectory\n")
    dt = np.diff(T)[0]
    N = len(T)
    r0 = np.random.get_state()
    np.random.seed(6771)
    if np.random.rand(1) > 0.5:
        x0 = +thet0 + np.sqrt(0.5 * sig0 * dt) * np.random.randn(1)
    else:
        x0 = -thet0 + np.sqrt(0.5 * sig0 * dt) * np.random.randn(1)
    ek = np.random.randn(N, 1)
    np.random.set_state(r0)
    return _path.run(_lib.SYS_DW, x0, thet0, _path.noise_scale(sig0, dt, 1), ek, dt)


def energy_mode(A, b, m, S, sDyn):
    """(Esde, Ef, Edf, dEsde_dm, dEsde_dS, dEsde_dth, dEsde_dSig) for the variational
    parameters A, b and marginal moments m, S; ``sDyn`` needs 'dt', 'Sig', 'theta' and
    'obsX' (the time integrals stop at the last observation, auxiliary/numerical.py:275-281)."""
    return _energy_mode(_lib.SYS_DW, A, b, m, S, sDyn)


energy_mode._vgpa_system = _lib.SYS_DW
