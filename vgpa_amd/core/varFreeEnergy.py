"""Variational free energy F(A, b) -- counterpart of core/varFreeEnergy.py:9-113."""
import numpy as np

from .. import engine
from . import _facade

__all__ = ["varFreeEnergy", "finite_difference_batched"]


def _evaluate(Ab, mParam, want_hyper=False, on_ready=None):
    """Run the device pipeline for one problem; returns (problem, A, b, state)."""
    sde = mParam["sde_struct"]
    N, D = int(sde["N"]), int(sde["D"])
    A, b = _facade.unpack_Ab(Ab, N, D)
    fun = mParam["fun"]
    system = _facade.system_of(fun)
    prob = _facade.cached_problem(system, sde, mParam["m0"], mParam["S0"], fun=fun if system < 0 else None)
    Ad = _facade.dev_batch1(A, D, "m")
    bd = _facade.dev_batch1(b, D, "v")
    st = engine.free_energy(prob, Ad, bd, want_hyper=want_hyper, on_ready=on_ready)
    engine.check_status(st.status)
    return prob, Ad, bd, st


_PUBLISHED = (("mt", "m"), ("St", "S"), ("Efx", "Ef"), ("Edf", "Edf"), ("lamt", "lam"), ("Psit", "Psi"))


def _publish(mParam, st, D, host=None):
    """The side effects of the reference (core/varFreeEnergy.py:84-104): host arrays
    that drivers index and deep-copy.  ``host``: arrays already copied out by name
    (_facade.AsyncPublisher); what is missing there is copied now."""
    host = dict(host or {})
    missing = [(k, n) for k, n in _PUBLISHED if n not in host]
    if missing:
        got = _facade.to_host([getattr(st, n)[0] for _, n in missing])
        host.update({n: a for (_, n), a in zip(missing, got)})
    for k, n in _PUBLISHED:
        a = host[n]
        mParam[k] = a.reshape(-1, 1) if D == 1 else a


def varFreeEnergy(Ab, mParam):
    """F = KL0 + Eobs + Esde for the variational parameters ``Ab`` (A then b, flat).

    ``mParam`` needs 'm0', 'S0', 'fun', 'sde_struct' (core/smoothing.py:59-64) and
    receives 'mt', 'St', 'Efx', 'Edf', 'lamt', 'Psit' as NumPy arrays.  Returns a
    Python float (core/varFreeEnergy.py:111).
    """
    # the six arrays are copied to the host on a second stream while the later kernels still run
    pub = _facade.AsyncPublisher()
    prob, Ad, bd, st = _evaluate(Ab, mParam, on_ready=pub.ready)
    _publish(mParam, st, prob.D, pub.collect())
    if prob.system >= 0:
        _facade.remember_published(mParam, prob, Ad, bd, st)
    return float(st.F[0].item())


def finite_difference_batched(x, mParam, h=1.0e-6, max_bytes=1 << 28):
    """Central differences of F along every coordinate of ``x`` (the work of
    auxiliary/numerical.py:132-144) with the 2n perturbed points evaluated as device
    batches: each batch holds k coordinates, rows 0..k-1 at x + h e_i and rows k..2k-1
    at x - h e_i.  Returns the (n,1) column of the reference."""
    import torch
    sde = mParam["sde_struct"]
    N, D = int(sde["N"]), int(sde["D"])
    fun = mParam["fun"]
    system = _facade.system_of(fun)
    x = np.asarray(x, dtype=np.float64).reshape(-1)
    n = x.size
    if system < 0:
        # a host plug-in gains nothing from batching: one point at a time
        grad = np.zeros((n, 1))
        e = np.zeros(n)
        for i in range(n):
            e[i] = h
            grad[i] = 0.5 * (varFreeEnergy(x + e, dict(mParam)) - varFreeEnergy(x - e, dict(mParam))) / h
            e[i] = 0.0
        return grad
    nA = N * D * D
    per = max(1, min(n, int(max_bytes // (16 * n))))
    base = engine.to_dev(x)
    grad = torch.empty(n, dtype=torch.float64, device=base.device)
    shape_A = (N,) if D == 1 else (N, D, D)
    shape_b = (N,) if D == 1 else (N, D)
    for start in range(0, n, per):
        k = min(per, n - start)
        X = base.repeat(2 * k, 1)
        rows = torch.arange(k, device=base.device)
        X[rows, start + rows] += h
        X[k + rows, start + rows] -= h
        A = X[:, :nA].reshape((2 * k,) + shape_A).contiguous()
        b = X[:, nA:].reshape((2 * k,) + shape_b).contiguous()
        prob = engine.make_problem(system, sde, mParam["m0"], mParam["S0"], P=2 * k)
        st = engine.free_energy(prob, A, b, want_edf=False, need_adjoint=False)
        engine.check_status(st.status)
        grad[start:start + k] = 0.5 * (st.F[:k] - st.F[k:]) / h
    return grad.cpu().numpy().reshape(n, 1)
