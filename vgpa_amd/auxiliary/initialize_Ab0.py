"""Starting values of the variational parameters: counterpart of auxiliary/initialize_Ab0.py:9.

The cubic smoothing splines through the observations (SciPy FITPACK) are fitted on the host,
one per state dimension; A0(t), b0(t) are filled in on the device (``vgpa_init_ab0``).
``initialize_Ab0_batched`` does the same for P problems and leaves the arrays on the device in
the (P,N,D,D) / (P,N,D) layout the evaluation takes.
"""
import numpy as np
from scipy import interpolate

__all__ = ["initialize_Ab0", "initialize_Ab0_batched"]


def _spline_means(sde_struct):
    """mt0 (N, D): per dimension a cubic UnivariateSpline through the observations with the
    first and the last one repeated at the ends of the window (initialize_Ab0.py:46-56, :74-77)."""
    Tw, obsX, obsY = sde_struct['Tw'], sde_struct['obsX'], sde_struct['obsY']
    obsZ = np.column_stack((obsY[0], obsY.T, obsY[-1])).T
    Tx = [Tw[0]]
    Tx.extend(Tw[obsX])
    Tx.extend([Tw[-1]])
    D = sde_struct['D']
    mt0 = np.zeros((sde_struct['N'], D))
    for i in range(D):
        mt0[:, i] = interpolate.UnivariateSpline(Tx, obsZ[:, i].T, k=3)(Tw)
    return mt0


def initialize_Ab0(S0, sde_struct):
    """[A0, b0] concatenated into one (K, 1) vector (auxiliary/initialize_Ab0.py:9-97):
    A0(t) = 0.5 diag(Sigma) / diag(S0), b0(t) = dm/dt + A0 m along the spline means m(t)."""
    Sigma, N, D, dt = sde_struct['Sig'], sde_struct['N'], sde_struct['D'], sde_struct['dt']
    mt0 = _spline_means(sde_struct)
    if D == 1:
        A0 = 0.5 * (Sigma / S0) * np.ones((N, 1))
        b0 = mt0[:, 0]
    else:
        from .. import engine
        d = engine.to_dev
        A0, b0 = engine.init_ab0(d(mt0[None]), d(np.array(Sigma.diagonal())), d(np.array(S0.diagonal())), dt)
        A0, b0 = A0[0].cpu().numpy(), b0[0].cpu().numpy()
    return np.concatenate((A0.ravel()[:, np.newaxis], b0.ravel()[:, np.newaxis]))


def initialize_Ab0_batched(S0, sde_structs):
    """P problems (a list of ``sde_struct`` dictionaries with equal N and D > 1; S0 (D,D) shared
    or a list of P): device tensors A0 (P,N,D,D), b0 (P,N,D)."""
    from .. import engine
    P = len(sde_structs)
    D = sde_structs[0]['D']
    if D == 1:
        raise ValueError("initialize_Ab0_batched: D > 1 (the 1-D case is a host formula, see initialize_Ab0)")
    mt0 = np.stack([_spline_means(s) for s in sde_structs])
    sig = np.stack([np.array(s['Sig'].diagonal()) for s in sde_structs])
    S0s = [S0] * P if isinstance(S0, np.ndarray) else list(S0)
    s0 = np.stack([np.array(x.diagonal()) for x in S0s])
    d = engine.to_dev
    return engine.init_ab0(d(mt0), d(sig), d(s0), sde_structs[0]['dt'])
