"""Reference-signature numerical helpers (auxiliary/numerical.py of the reference)
on top of the HIP library: ``ut_approx`` and ``mytrapz`` run on the device,
``finiteDiff`` batches its 2n evaluations when the function is this package's
``varFreeEnergy``.  ``chol_inv`` / ``log_det`` act on the run's constant noise
matrices (a few dozen entries, once per run) and stay on the host."""
import numpy as np
import torch

from .. import _lib, engine

__all__ = ["chol_inv", "log_det", "finiteDiff", "ut_approx", "mytrapz", "safelog"]


def chol_inv(A):
    """auxiliary/numerical.py:10-48: (inverse of A, inverse of its Cholesky factor);
    a vector means a diagonal matrix, a scalar gives (1/A, 1/sqrt(A))."""
    if A is None:
        print("Input matrix is None!")
        return None, None
    if np.isscalar(A):
        return 1.0 / A, 1.0 / np.sqrt(A)
    A = np.asarray(A, dtype=np.float64)
    if A.ndim == 1:
        A = np.diag(A)
    Ri = np.linalg.inv(np.linalg.cholesky(A))
    return Ri.dot(Ri.T), Ri


def log_det(A):
    """auxiliary/numerical.py:51-86: log det A from the Cholesky factor."""
    if A is None:
        print("Input matrix is None!")
        return None
    if np.isscalar(A):
        return np.log(A)
    A = np.asarray(A, dtype=np.float64)
    if A.ndim == 1:
        A = np.diag(A)
    return 2.0 * np.sum(np.log(np.linalg.cholesky(A).diagonal()))


def safelog(x=None):
    """auxiliary/numerical.py:287-346: log with the argument clipped to [1e-300, 1e300].
    Off the hot path (no caller on it); kept for drivers that import it."""
    if x is None:
        print("[safelog] in debug mode ... exiting:", end="")
        return None
    return np.log(np.clip(np.asarray(x, dtype=np.float64), 1.0e-300, 1.0e+300))


def mytrapz(fxt, dt, idx=None):
    """auxiliary/numerical.py:231-284 as a weighted device reduction
    (vgpa_reduce_energy): dt * sum_t w[t] fxt[t], w from ``engine.trapz_weights``.
    fxt (N,) -> scalar, (N,c) -> (c,)."""
    f = np.asarray(fxt, dtype=np.float64)
    N = f.shape[0]
    x = engine.to_dev(f.reshape(1, N, -1))
    w = engine.to_dev(engine.trapz_weights(N, idx))
    out = engine.reduce_energy(x, w, float(dt))[0].cpu().numpy()
    return out.reshape(f.shape[1:]) if f.ndim > 1 else float(out[0])


def ut_approx(f, xbar, Pxx, *args):
    """auxiliary/numerical.py:153-228.  Sigma points and the weighted moments are
    computed on the device; ``f`` is the caller's Python function and is applied to
    the (2D+1, D) sigma points on the host, as in the reference.  Returns
    (ybar (1,C), Pyy (C,C)); raises numpy.linalg.LinAlgError when neither Cholesky
    factorisation exists (:199-202)."""
    xb = np.asarray(xbar, dtype=np.float64)
    D = xb.shape[0] if xb.ndim == 1 else xb.shape[1]
    chi, status = engine.ut_sigma_points(engine.to_dev(xb.reshape(1, D)), engine.to_dev(np.asarray(Pxx).reshape(1, D, D)))
    if int(status.item()) & _lib.ST_CHOL_FAILED:
        raise np.linalg.LinAlgError("Matrix is not positive definite")
    Y = np.asarray(f(chi[0].cpu().numpy(), *args), dtype=np.float64)
    Y = Y.reshape(2 * D + 1, -1)
    ybar, Pyy = engine.ut_moments(engine.to_dev(Y[None]))
    return ybar.cpu().numpy(), Pyy[0].cpu().numpy()


def finiteDiff(fun, x, mParam, h=1.0e-6):
    """auxiliary/numerical.py:89-150: central differences of ``fun`` along every
    coordinate.  When ``fun`` is this package's varFreeEnergy the 2n perturbed points
    are evaluated as device batches instead of one by one."""
    from ..core import varFreeEnergy as vfe
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    x = x.reshape(n, 1)
    print("Checking numerically the gradient...")
    if fun is vfe.varFreeEnergy:
        return vfe.finite_difference_batched(x, mParam, h)
    gradN = np.zeros((n, 1))
    e = np.zeros((n, 1))
    for i in range(n):
        e[i] = 1.0
        gradN[i] = 0.5 * (fun(x + h * e, mParam) - fun(x - h * e, mParam)) / h
        e[i] = 0.0
        if i % 100 == 0:
            print("Done: {:.3}%".format(i / n * 100))
    return gradN
