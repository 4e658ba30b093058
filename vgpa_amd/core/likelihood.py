"""Gaussian observation energy -- counterpart of core/likelihood.py@13 ``gauss``."""
import numpy as np

from .. import engine
from . import _facade

__all__ = ["gauss"]


def gauss(m, S, sde_struct):
    """Returns (Eobs, dEobs_dm, dEobs_dS, dEobs_dR) with the reference's shapes:
    (N,1) arrays for D == 1 (core/likelihood.py@52-105), (N,D), (N,D,D), (N,d) for
    D > 1 (@108-179).  dEobs_dR is not used by the free-energy path: it is zero for
    D > 1 as in the reference (@176) and formed on the host from the inputs for
    D == 1 (@102)."""
    sde = sde_struct
    D, N = int(sde["D"]), int(sde["N"])
    dummy0 = np.zeros(D) if D > 1 else np.zeros(1)
    dummyS = np.eye(D) if D > 1 else 1.0
    prob = engine.make_problem(-1, dict(sde, px0=dict(mu0=dummy0, tau0=dummyS)), dummy0, dummyS, P=1)
    md, Sd = _facade.dev_batch1(m, D, "v"), _facade.dev_batch1(S, D, "m")
    Eobs, dm, jumpS = engine.obs_terms(prob, md, Sd)
    idx = np.asarray(sde["obsX"]).reshape(-1)
    dEobs_dm = _facade.host_vec(dm, D)
    E = float(Eobs[0].item())
    if D == 1:
        dEobs_dS = np.zeros((N, 1))
        dEobs_dS[idx] = float(jumpS[0].item())
        m_, S_ = np.asarray(m, dtype=np.float64).reshape(N, 1), np.asarray(S, dtype=np.float64).reshape(N, 1)
        Y, Ri = np.asarray(sde["obsY"], dtype=np.float64).reshape(-1, 1), 1.0 / sde["Rig"]
        dEobs_dR = np.zeros((N, 1))
        dEobs_dR[idx] = -0.5 * Ri * (Y**2 - 2 * Y * m_[idx] + (m_**2 + S_)[idx]) + 0.5 * Ri
        return E, dEobs_dm, dEobs_dS, dEobs_dR
    dEobs_dS = np.zeros((N, D, D))
    dEobs_dS[idx] = jumpS.cpu().numpy()
    d = np.asarray(sde["obsY"]).shape[1]
    return E, dEobs_dm, dEobs_dS, np.zeros((N, d))
