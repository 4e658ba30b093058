"""Drift-parameter estimation for P problems at once, on the device.

Counterpart of the outer loop of lorenz-63-parameter-estimation.py:189-277
(``estimate_paramters``): smooth with the current parameters, take dEsde/dtheta at the
optimum, move the parameters down the gradient with a fixed rate, repeat until the free
energy or the parameters stop changing.  Here every problem of the batch follows that loop
with its own parameters; the inner smoothing is ``scg_batched`` and the hyper-parameter
gradient is a by-product of the energy kernels (``engine.free_energy(want_hyper=True)``).
"""
__all__ = ["estimate_parameters_batched"]


class _DeviceLoopOps:
    """The two evaluations the loop needs, on the device: the inner smoothing
    (``scg_batched``) and dEsde/dtheta at its optimum (a by-product of the energy kernels)."""

    def __init__(self, prob, scg_nit):
        self.prob, self.scg_nit = prob, scg_nit

    def _bind(self, theta):
        self.prob.theta = theta.reshape(self.prob.theta.shape).contiguous()

    def smooth(self, theta, A_start, b_start):
        from .optimize import scg_batched
        self._bind(theta)
        return scg_batched(self.prob, A_start, b_start, nit=self.scg_nit)[:3]

    def dth(self, theta, A, b):
        from .. import engine
        prob = self.prob
        self._bind(theta)
        st = engine.free_energy(prob, A, b, want_hyper=True, want_edf=prob.D == 1 or prob.system < 0,
                                need_adjoint=False)
        return st.dth.reshape(prob.P, -1)


def estimate_parameters_batched(prob, A0, b0, rate=0.01, nit=250, scg_nit=150, tol=1.0e-6, warm_start=False,
                                disp=False, ops=None, theta0=None):
    """Gradient descent on the drift parameters of the P problems of ``prob``.

    prob.theta (P, n_theta) holds the starting parameters and is updated in place of the
    problem (``prob.theta`` is rebound to the current values).  A0, b0: starting variational
    parameters of the inner smoothing; the reference starts every smoothing from the same
    initial values (``run_smoothing`` rebuilds them, :183-187), ``warm_start=True`` continues
    from the previous optimum instead.  Per problem the loop ends when |F - F_old| <= tol or
    ||theta - theta_old||_2 <= tol (:267), when the gradient is exactly zero (:213), or after
    ``nit`` updates.

    ``ops`` replaces the two device evaluations (an object with ``smooth(theta, A, b) ->
    (A, b, F)`` and ``dth(theta, A, b) -> (P, n)``; then ``prob`` may be None and ``theta0``
    (P, n) gives the starting parameters): it lets the loop logic be driven by another
    evaluator.

    Returns dict(theta (P,n), F (P,), A, b, costs (n_outer + 1, P), grads (n_outer, P, n),
    thetas (n_outer + 1, P, n), iterations (P,)): ``costs`` is the free energy after every
    smoothing, as the reference's list ``costs``; ``thetas`` the parameters each of them
    was run with."""
    import torch

    if ops is None:
        ops = _DeviceLoopOps(prob, scg_nit)
    theta = (prob.theta if theta0 is None else theta0).clone()
    P = A0.shape[0]
    theta = theta.reshape(P, -1)
    n = theta.shape[1]
    active = torch.ones((P,), dtype=torch.bool, device=theta.device)
    iterations = torch.zeros((P,), dtype=torch.int64, device=theta.device)

    def smooth(A_start, b_start):
        return ops.smooth(theta, A_start, b_start)

    A, b, F = smooth(A0, b0)
    costs, grads, thetas = [F.clone()], [], [theta.clone()]
    F_old, theta_old = F.clone(), theta.clone()
    for it in range(nit):
        if not bool(active.any().item()):
            break
        g = ops.dth(theta, A, b)
        if g.shape[1] != n:
            raise ValueError(f"estimate_parameters_batched: dEsde/dtheta has {g.shape[1]} components per problem, "
                             f"theta has {n}")
        grads.append(g.clone())
        active = active & (g != 0).any(dim=1)                        # :213
        theta = torch.where(active.unsqueeze(1), theta - rate * g, theta)
        A_new, b_new, F_new = smooth(A if warm_start else A0, b if warm_start else b0)
        # problems that have finished keep their results
        keep = active.reshape((P,) + (1,) * (A.dim() - 1))
        A = torch.where(keep, A_new, A)
        b = torch.where(active.reshape((P,) + (1,) * (b.dim() - 1)), b_new, b)
        F = torch.where(active, F_new, F)
        costs.append(F.clone())
        thetas.append(theta.clone())
        dF = (F - F_old).abs()
        dth = torch.linalg.vector_norm(theta - theta_old, ord=2, dim=1)
        iterations += active.long()
        if disp:
            print(f"outer {it}: min F {float(F.min().item()):.4f}  active {int(active.sum().item())}")
        F_old, theta_old = F.clone(), theta.clone()
        active = active & ~((dF <= tol) | (dth <= tol))              # :267
    if prob is not None:
        prob.theta = theta.reshape(prob.theta.shape).contiguous()
    return dict(theta=theta, F=F, A=A, b=b, costs=torch.stack(costs), iterations=iterations,
                thetas=torch.stack(thetas),
                grads=torch.stack(grads) if grads else torch.zeros((0, P, n), dtype=theta.dtype, device=theta.device))
