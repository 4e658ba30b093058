"""Shared body of the five ``energy_mode(A, b, m, S, sDyn)`` entry points."""
import numpy as np

from .. import _lib, engine
from ..core import _facade


def energy_mode(system, A, b, m, S, sDyn):
    """NumPy in, device kernels, NumPy out, with the reference's output shapes
    (e.g. dynamics/sys_lorenz_96_ut.py:258): Esde (1,), Ef (N,D), Edf (N,D,D),
    dEsde_dm (N,D), dEsde_dS (N,D,D), dEsde_dth, dEsde_dSig; (N,1) arrays and (1,)
    hyper-gradients for the 1-D systems."""
    A = np.asarray(A, dtype=np.float64)
    D = 1 if (A.ndim == 1 or A.shape[1] == 1 and A.ndim == 2) else A.shape[1]
    N = A.shape[0]
    dt = float(sDyn["dt"])
    idx = sDyn.get("obsX", None)
    w = engine.to_dev(engine.trapz_weights(N, idx))
    if D == 1:
        SigInv = engine.to_dev(np.array([1.0 / float(np.asarray(sDyn["Sig"]).reshape(-1)[0])]))
        sid = SigInv
        theta = engine.to_dev(np.asarray(sDyn["theta"], dtype=np.float64).reshape(1))
    else:
        SI = np.linalg.inv(np.asarray(sDyn["Sig"], dtype=np.float64))   # sys_lorenz_96_ut.py:184
        SigInv, sid = engine.to_dev(SI), engine.to_dev(np.diag(SI).copy())
        theta = engine.to_dev(np.asarray(sDyn["theta"], dtype=np.float64).reshape(-1))
    Ad, Sd = _facade.dev_batch1(A, D, "m"), _facade.dev_batch1(S, D, "m")
    bd, md = _facade.dev_batch1(b, D, "v"), _facade.dev_batch1(m, D, "v")
    out = engine.energy_terms(system, Ad, bd, md, Sd, SigInv, sid, theta, w, dt, want_hyper=True)
    engine.check_status(out["status"])
    Esde = out["Esde"].cpu().numpy().reshape(1)
    dth = out["dth"][0].cpu().numpy().reshape(-1)
    dSig = out["dSig"][0].cpu().numpy()
    if D == 1:
        dSig = dSig.reshape(1)
    return (Esde, _facade.host_vec(out["Ef"], D), _facade.host_mat(out["Edf"], D),
            _facade.host_vec(out["dEdm"], D), _facade.host_mat(out["dEdS"], D), dth, dSig)
