"""auxiliary/momGauss.py of the reference on the device (vgpa_mom_gauss)."""
import numpy as np

from .. import engine

_DERIV = {None: 0, "Dm": 1, "DS": 2}


def momGauss(m, S, order=0, dmS=None):
    """auxiliary/momGauss.py:5-107: uncentred moment of order 0..8 of N(m, S), or its
    derivative w.r.t. m ('Dm') or S ('DS'), element-wise; (N,1) in, (N,1) out.
    ValueError for any other order (:104).  Any other ``dmS`` gives None, as there."""
    if not (isinstance(order, (int, np.integer)) and 0 <= order <= 8):
        raise ValueError("Unknown Higher Order Gaussian Moment!")
    if order > 0 and dmS not in _DERIV:
        return None
    ma = np.asarray(m, dtype=np.float64)
    Sa = np.broadcast_to(np.asarray(S, dtype=np.float64), ma.shape)
    out = engine.mom_gauss(engine.to_dev(ma.reshape(-1)), engine.to_dev(Sa.reshape(-1)), int(order),
                           _DERIV.get(dmS, 0)).cpu().numpy()
    # the constant entries of the table are built as (N,1) arrays there (:38, :43, :45, :53)
    constant = order == 0 or (order == 1 and dmS is not None) or (order == 2 and dmS == "DS")
    return out.reshape(ma.shape[0], -1)[:, :1] if constant else out.reshape(ma.shape)
