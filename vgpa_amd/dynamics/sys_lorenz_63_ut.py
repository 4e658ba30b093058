"""Expected SDE energy for the Lorenz-63 with unscented-transform expectations.

Counterpart of ``energy_mode`` in dynamics/sys_lorenz_63_ut.py:132; trajectory generation and plotting
of that file are outside this package.
"""
from .. import _lib
from ._energy import energy_mode as _energy_mode

__all__ = ["energy_mode"]


def energy_mode(A, b, m, S, sDyn):
    """(Esde, Ef, Edf, dEsde_dm, dEsde_dS, dEsde_dth, dEsde_dSig) for the variational
    parameters A, b and marginal moments m, S; ``sDyn`` needs 'dt', 'Sig', 'theta' and
    'obsX' (the time integrals stop at the last observation, auxiliary/numerical.py:275-281)."""
    return _energy_mode(_lib.SYS_L63_UT, A, b, m, S, sDyn)


energy_mode._vgpa_system = _lib.SYS_L63_UT
