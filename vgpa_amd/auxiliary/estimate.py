"""Drift-parameter estimation for P problems at once, on the device.

Counterpart of the outer loop of lorenz-63-parameter-estimation.py:189-277
(``estimate_paramters``): smooth with the current parameters, take dEsde/dtheta at the
optimum, move the parameters down the gradient with a fixed rate, repeat until the free
energy or the parameters stop changing.  Here every problem of the batch follows that loop
with its own parameters; the inner smoothing is ``scg_batched`` and the hyper-parameter
gradient is a by-product of the energy kernels (``engine.free_energy(want_hyper=True)``).
"""
__all__ = ["estimate_parameters_batched"]


def estimate_parameters_batched(prob, A0, b0, rate=0.01, nit=250, scg_nit=150, tol=1.0e-6, warm_start=False,
                                disp=False):
    """Gradient descent on the drift parameters of the P problems of ``prob``.

    prob.theta (P, n_theta) holds the starting parameters and is updated in place of the
    problem (``prob.theta`` is rebound to the current values).  A0, b0: starting variational
    parameters of the inner smoothing; the reference starts every smoothing from the same
    initial values (``run_smoothing`` rebuilds them, :183-187), ``warm_start=True`` continues
    from the previous optimum instead.  Per problem the loop ends when |F - F_old| <= tol or
    ||theta - theta_old||_2 <= tol (:267), when the gradient is exactly zero (:213), or after
    ``nit`` updates.

    Returns dict(theta (P,n), F (P,), A, b, costs (n_outer + 1, P), grads (n_outer, P, n),
    iterations (P,)): ``costs`` is the free energy after every smoothing, as the reference's
    list ``costs``."""
    import torch
    from .. import engine
    from .optimize import scg_batched

    P = prob.P
    theta = prob.theta.reshape(P, -1).clone()
    n = theta.shape[1]
    active = torch.ones((P,), dtype=torch.bool, device=theta.device)
    iterations = torch.zeros((P,), dtype=torch.int64, device=theta.device)

    def smooth(A_start, b_start):
        prob.theta = theta.reshape(prob.theta.shape).contiguous()
        return scg_batched(prob, A_start, b_start, nit=scg_nit)[:3]

    A, b, F = smooth(A0, b0)
    costs, grads = [F.clone()], []
    F_old, theta_old = F.clone(), theta.clone()
    for it in range(nit):
        if not bool(active.any().item()):
            break
        st = engine.free_energy(prob, A, b, want_hyper=True, want_edf=prob.D == 1 or prob.system < 0,
                                need_adjoint=False)
        g = st.dth.reshape(P, -1)
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
        dF = (F - F_old).abs()
        dth = torch.linalg.vector_norm(theta - theta_old, ord=2, dim=1)
        iterations += active.long()
        if disp:
            print(f"outer {it}: min F {float(F.min().item()):.4f}  active {int(active.sum().item())}")
        F_old, theta_old = F.clone(), theta.clone()
        active = active & ~((dF <= tol) | (dth <= tol))              # :267
    prob.theta = theta.reshape(prob.theta.shape).contiguous()
    return dict(theta=theta, F=F, A=A, b=b, costs=torch.stack(costs), iterations=iterations,
                grads=torch.stack(grads) if grads else torch.zeros((0, P, n), dtype=theta.dtype, device=theta.device))
