"""Gradient of the Lagrangian w.r.t. A(t), b(t) -- counterpart of core/gradL_Ab.py:11-180."""
import numpy as np

from .. import engine
from . import _facade
from .varFreeEnergy import _evaluate

__all__ = ["gradL_Ab"]


def gradL_Ab(Ab, mParam, evalF=False, outer=False):
    """Returns the (n,1) column [dt*dL/dA.ravel(); dt*dL/db.ravel()].

    With ``evalF`` the free energy is evaluated first on a private copy of
    ``mParam`` (core/gradL_Ab.py:36-41), so the caller's dictionary is untouched;
    otherwise 'mt', 'St', 'lamt', 'Psit', 'Efx', 'Edf' are read from ``mParam`` and
    must come from ``varFreeEnergy`` at the same ``Ab`` (core/gradL_Ab.py:140-149).

    ``outer=True`` (not the reference's behaviour) replaces the two scalar inner
    products of core/gradL_Ab.py:167,170 by outer products.
    """
    sde = mParam["sde_struct"]
    N, D = int(sde["N"]), int(sde["D"])
    if evalF:
        prob, Ad, bd, st = _evaluate(Ab, dict(mParam))
        m, S, lam, Psi, Ef, Edf = st.m, st.S, st.lam, st.Psi, st.Ef, st.Edf
    else:
        A, b = _facade.unpack_Ab(Ab, N, D)
        Ad, bd = _facade.dev_batch1(A, D, "m"), _facade.dev_batch1(b, D, "v")
        cached = _facade.recall_published(mParam, Ad, bd)
        if cached is not None:
            # the arrays in mParam are the ones varFreeEnergy just published for this point
            prob, st = cached
            m, S, lam, Psi, Ef, Edf = st.m, st.S, st.lam, st.Psi, st.Ef, st.Edf
        else:
            prob = engine.make_problem(_facade.system_of(mParam["fun"]), sde, mParam["m0"], mParam["S0"], P=1)
            m, lam, Ef = (_facade.dev_batch1(mParam[k], D, "v") for k in ("mt", "lamt", "Efx"))
            S, Psi, Edf = (_facade.dev_batch1(mParam[k], D, "m") for k in ("St", "Psit", "Edf"))
    import torch
    nA, nb = (N, N) if D == 1 else (N * D * D, N * D)
    flat = torch.empty(nA + nb, dtype=torch.float64, device=Ad.device)
    engine.grad_assemble(prob, Ad, bd, m, S, lam, Psi, Ef, Edf, outer=outer,
                         out=(flat[:nA].view(Ad.shape), flat[nA:].view(bd.shape)))
    return _facade.to_host([flat])[0].reshape(-1, 1)
