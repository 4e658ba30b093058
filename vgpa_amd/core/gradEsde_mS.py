"""core/gradEsde_mS.py of the reference on the device (vgpa_grad_esde_ms)."""
import numpy as np

from .. import engine

__all__ = ["gradEsde_mS"]


def gradEsde_mS(x, F, mt, St, At, bt, diagSigI, sDyn=None):
    """core/gradEsde_mS.py:8-68: per-sample gradient terms of the SDE energy w.r.t. the
    marginal mean and covariance, (K, D + D*D).  ``F(x, sDyn)`` is the caller's drift
    and is evaluated on the host; the solves with St and the outer products run in one
    kernel (one wavefront per sample)."""
    x = np.asarray(x, dtype=np.float64)
    K, D = x.shape
    Fx = np.asarray(F(x, sDyn), dtype=np.float64).reshape(K, D)
    dev = engine.to_dev
    out = engine.grad_esde_ms(dev(x), dev(Fx), dev(np.reshape(mt, D)), dev(np.reshape(St, (D, D))),
                              dev(np.reshape(At, (D, D))), dev(np.reshape(bt, D)), dev(np.reshape(diagSigI, D)))
    return out.cpu().numpy()
