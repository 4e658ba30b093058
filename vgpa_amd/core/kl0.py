"""Initial-state KL energy -- counterpart of core/kl0.py:9 ``gauss``."""
import numpy as np

from .. import engine

__all__ = ["gauss"]


def gauss(m0, S0, lam, Psi, sde_struct):
    """Returns (KL0, dKL0_dm0, dKL0_dS0).  KL0 comes from the device kernel and keeps
    the reference's dependence on the shape of ``m0`` ((D,) broadcasts |z0|^2 into the
    matrix, (1,D) gives the outer product; core/kl0.py:88-90) and the 1-D form with
    -log(S0) (core/kl0.py:60).  The two gradients are not used by the free-energy
    path (core/varFreeEnergy.py:107-108); they are formed on the host from lam[0],
    Psi[0] as in core/kl0.py:63-66, :93-96."""
    D = int(sde_struct["D"])
    mu0, tau0 = sde_struct["px0"]["mu0"], sde_struct["px0"]["tau0"]
    m0a = np.asarray(m0, dtype=np.float64)
    fake = dict(D=D, N=2, dt=1.0, ode_method="EULER", obsX=np.array([1]), theta=0.0,
                obsY=np.zeros((1, D)), H=(np.eye(D) if D > 1 else 1.0), Sig=(np.eye(D) if D > 1 else 1.0),
                Rig=(np.eye(D) if D > 1 else 1.0), px0=sde_struct["px0"])
    prob = engine.make_problem(-1, fake, m0a, S0, P=1)
    val = float(engine.kl0(prob)[0].item())
    z0 = m0a - mu0
    if D == 1:
        KL0 = np.full(np.shape(z0), val) if np.ndim(z0) else np.float64(val)
        return KL0, lam[0] + z0 / tau0, Psi[0] + 0.5 * (1.0 / tau0 - 1.0 / S0)
    tau0i, S0i = np.linalg.inv(tau0), np.linalg.inv(S0)
    return np.float64(val), lam[0, :] + np.linalg.solve(tau0, z0.T).T, Psi[0, :, :] + 0.5 * (tau0i - S0i)
