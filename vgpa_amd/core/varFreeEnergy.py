"""Variational free energy F(A, b) -- counterpart of core/varFreeEnergy.py:9-113."""
import numpy as np

from .. import engine
from . import _facade

__all__ = ["varFreeEnergy"]


def _evaluate(Ab, mParam, want_hyper=False):
    """Run the device pipeline for one problem; returns (problem, A, b, state)."""
    sde = mParam["sde_struct"]
    N, D = int(sde["N"]), int(sde["D"])
    A, b = _facade.unpack_Ab(Ab, N, D)
    fun = mParam["fun"]
    system = _facade.system_of(fun)
    prob = engine.make_problem(system, sde, mParam["m0"], mParam["S0"], P=1,
                               fun=fun if system < 0 else None)
    Ad = _facade.dev_batch1(A, D, "m")
    bd = _facade.dev_batch1(b, D, "v")
    st = engine.free_energy(prob, Ad, bd, want_hyper=want_hyper)
    engine.check_status(st.status)
    return prob, Ad, bd, st


def _publish(mParam, st, D):
    """The side effects of the reference (core/varFreeEnergy.py:84-104): host arrays
    that drivers index and deep-copy."""
    mParam["mt"] = _facade.host_vec(st.m, D)
    mParam["St"] = _facade.host_mat(st.S, D)
    mParam["Efx"] = _facade.host_vec(st.Ef, D)
    mParam["Edf"] = _facade.host_mat(st.Edf, D)
    mParam["lamt"] = _facade.host_vec(st.lam, D)
    mParam["Psit"] = _facade.host_mat(st.Psi, D)


def varFreeEnergy(Ab, mParam):
    """F = KL0 + Eobs + Esde for the variational parameters ``Ab`` (A then b, flat).

    ``mParam`` needs 'm0', 'S0', 'fun', 'sde_struct' (core/smoothing.py:59-64) and
    receives 'mt', 'St', 'Efx', 'Edf', 'lamt', 'Psit' as NumPy arrays.  Returns a
    Python float (core/varFreeEnergy.py:111).
    """
    prob, _, _, st = _evaluate(Ab, mParam)
    _publish(mParam, st, prob.D)
    return float(st.F[0].item())
