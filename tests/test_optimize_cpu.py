"""The optimisers of vgpa_amd.auxiliary.optimize on the CPU, with the NumPy oracle as
the function being minimised: a recorded run of the reference's optim_SCG (through the
reference's own varFreeEnergy / gradL_Ab) must be reproduced step by step, by the
scalar optimiser and by every problem of the batched one."""
import contextlib
import io

import numpy as np
import pytest
import torch

from conftest import load_golden, sde_from_golden, initial_moments, relerr
from oracle import vgpa_oracle as orc
from vgpa_amd.auxiliary import optimize

RUNS = [("ou", "scg_ou.npz"), ("l63", "scg_l63.npz")]
# The recorded Ornstein-Uhlenbeck run stalls after 30 iterations: every later trial step
# changes F by less than its rounding error, so whether it is accepted depends on the
# last bit of the dot products (NumPy and torch sum in different orders).  A batched run
# is compared row by row up to there, and by its final value after.
RESOLVED_ROWS = {"scg_ou.npz": 30, "scg_l63.npz": 25}


def _oracle_pair(system, sde, m0, S0):
    def f(x, mParam):
        out = orc.evaluate(system, x, m0, S0, sde)
        mParam["state"] = out
        return out["F"]

    def gf(x, mParam, evalF=False):
        if evalF:
            return orc.evaluate(system, x, m0, S0, sde)["grad"].reshape(-1, 1)
        return mParam["state"]["grad"].reshape(-1, 1)
    return f, gf


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize("system,fname", RUNS)
def test_optim_scg_replays_the_recorded_reference_run(system, fname):
    g = load_golden(fname)
    sde = sde_from_golden(g)
    m0, S0 = initial_moments(g)
    f, gf = _oracle_pair(system, sde, m0, S0)
    nit = int(g["nit"])
    x, fx, mParam, stat = _quiet(optimize.optim_SCG, f, g["Ab"], gf,
                                 dict(nit=nit, xtol=1e-6, ftol=1e-8, disp=False, lmin=True), {})
    assert stat["f_eval"] == int(g["f_eval"]) and stat["g_eval"] == int(g["g_eval"])
    assert stat["MaxIt"] == int(g["MaxIt"])
    assert stat["Fx"].shape == g["Fx"].shape
    assert relerr(stat["Fx"], g["Fx"]) <= 1e-9
    assert relerr(stat["beta"], g["beta"]) <= 1e-9
    assert relerr(stat["gradx"], g["gradx"]) <= 1e-7
    assert relerr(fx, g["fx"]) <= 1e-9
    assert x.shape == g["x"].shape and relerr(x, g["x"]) <= 1e-7


def _batched_evaluator(system, sdes, m0, S0, N, D):
    def evaluate(A, b, need_grad):
        F, G = [], []
        for p, sde in enumerate(sdes):
            Ab = np.concatenate((A[p].numpy().ravel(), b[p].numpy().ravel()))
            out = orc.evaluate(system, Ab, m0, S0, sde, with_grad=need_grad)
            F.append(out["F"])
            if need_grad:
                G.append(out["grad"].ravel())
        return torch.tensor(F, dtype=torch.float64), (torch.from_numpy(np.stack(G)) if need_grad else None)
    return evaluate


@pytest.mark.parametrize("system,fname", RUNS)
def test_scg_batched_follows_the_same_trajectory_per_problem(system, fname):
    """Problem 0 is the recorded run; problem 1 has other observations, so the two
    take different accept / reject decisions and must not disturb each other."""
    g = load_golden(fname)
    sde = sde_from_golden(g)
    m0, S0 = initial_moments(g)
    N, D = int(g["N"]), int(g["D"])
    sde2 = dict(sde, obsY=np.array(sde["obsY"]) * 0.5 + 0.2)
    A0, b0 = orc.unpack_Ab(g["Ab"], N, D)
    shpA, shpb = ((2, N), (2, N)) if D == 1 else ((2, N, D, D), (2, N, D))
    A = torch.from_numpy(np.broadcast_to(A0.reshape(shpA[1:]), shpA).copy())
    b = torch.from_numpy(np.broadcast_to(b0.reshape(shpb[1:]), shpb).copy())
    nit = int(g["nit"])
    ev = _batched_evaluator(system, [sde, sde2], m0, S0, N, D)
    Aout, bout, F, stat = optimize.scg_batched(None, A, b, nit=nit, evaluate=ev)
    # problem 0 against the recorded reference run
    K = RESOLVED_ROWS[fname]
    assert relerr(stat["Fx"][:K, 0].numpy(), g["Fx"].ravel()[:K]) <= 1e-9
    assert relerr(stat["beta"][:K, 0].numpy(), g["beta"].ravel()[:K]) <= 1e-9
    assert relerr(stat["gradx"][:K, 0].numpy(), g["gradx"].ravel()[:K]) <= 1e-7
    if K == nit:
        assert int(stat["f_eval"][0]) == int(g["f_eval"]) and int(stat["g_eval"][0]) == int(g["g_eval"])
        assert int(stat["MaxIt"][0]) == int(g["MaxIt"])
    assert relerr(F[0].item(), g["fx"]) <= 1e-9
    x0 = np.concatenate((Aout[0].numpy().ravel(), bout[0].numpy().ravel()))
    assert relerr(x0, g["x"].ravel()) <= 1e-7
    # problem 1 against the scalar optimiser on the same problem
    f, gf = _oracle_pair(system, sde2, m0, S0)
    x, fx, _, st1 = _quiet(optimize.optim_SCG, f, g["Ab"], gf, dict(nit=nit), {})
    assert relerr(stat["Fx"][:K, 1].numpy(), st1["Fx"].ravel()[:K]) <= 1e-9
    assert relerr(stat["beta"][:K, 1].numpy(), st1["beta"].ravel()[:K]) <= 1e-9
    if K == nit:
        assert int(stat["f_eval"][1]) == st1["f_eval"] and int(stat["MaxIt"][1]) == st1["MaxIt"]
    assert relerr(F[1].item(), fx) <= 1e-8


def test_scg_batched_freezes_finished_problems():
    """Quadratic bowls of different conditioning: each problem stops by its own
    tolerance test, with the same counters as the scalar optimiser, while the
    others keep iterating."""
    rng = np.random.default_rng(4)
    n, P = 6, 3
    Hs = []
    for p in range(P):
        Q = np.linalg.qr(rng.standard_normal((n, n)))[0]
        Hs.append(Q.dot(np.diag(np.linspace(1.0, 10.0 ** (p + 0.5), n))).dot(Q.T))
    c = rng.standard_normal((P, n))
    x0 = rng.standard_normal((P, n))

    def evaluate(A, b, need_grad):
        x = torch.cat((A.reshape(P, -1), b.reshape(P, -1)), dim=1).numpy()
        F = [0.5 * x[p].dot(Hs[p]).dot(x[p]) - c[p].dot(x[p]) for p in range(P)]
        G = np.stack([Hs[p].dot(x[p]) - c[p] for p in range(P)])
        return torch.tensor(F, dtype=torch.float64), torch.from_numpy(G)

    A, b = torch.from_numpy(x0[:, :4].copy()), torch.from_numpy(x0[:, 4:].copy())
    Aout, bout, F, stat = optimize.scg_batched(None, A, b, nit=80, evaluate=evaluate)
    its = []
    for p in range(P):
        f = lambda x, mp: float(0.5 * x.ravel().dot(Hs[p]).dot(x.ravel()) - c[p].dot(x.ravel()))   # noqa: E731
        gf = lambda x, mp, e=False: (Hs[p].dot(x.ravel()) - c[p]).reshape(-1, 1)                     # noqa: E731
        x, fx, _, st = _quiet(optimize.optim_SCG, f, x0[p].copy(), gf, dict(nit=80), {})
        # once the minimum is reached to rounding the accept / reject decisions are noise;
        # the first n rows (one conjugate-gradient sweep) and the end point are not
        assert relerr(stat["Fx"][:n, p].numpy(), st["Fx"].ravel()[:n]) <= 1e-9
        assert relerr(stat["beta"][:n, p].numpy(), st["beta"].ravel()[:n]) <= 1e-9
        assert relerr(F[p].item(), fx) <= 1e-10
        got = np.concatenate((Aout[p].numpy().ravel(), bout[p].numpy().ravel()))
        assert relerr(got, x.ravel()) <= 1e-6
        k = int(stat["MaxIt"][p])
        assert k < 80 and np.all(stat["Fx"][k + 1:, p].numpy() == 0.0)   # frozen after it stopped
        its.append(k)
    assert len(set(its)) > 1   # they really stop at different iterations


def test_backtrack_and_cg_minimise_a_quadratic():
    Hm = np.diag([1.0, 4.0, 9.0])
    f = lambda x, mp: float(0.5 * x.ravel().dot(Hm).dot(x.ravel()))   # noqa: E731
    gf = lambda x, mp, e=False: Hm.dot(x.reshape(-1, 1))               # noqa: E731
    eta, cnt = optimize.backtrack(f, np.ones((3, 1)), f(np.ones(3), {}), -np.ones((3, 1)) / np.sqrt(3), 0.15, 0.5, {})
    assert eta == 0.15 and cnt == 1
    x, fx, _, stat = _quiet(optimize.optim_CG, f, np.array([1.0, -2.0, 0.5]), gf, dict(nit=200), {})
    assert fx < 1e-3 and stat["f_eval"] > stat["g_eval"]


def test_estimation_loop_replays_the_recorded_reference_run():
    """Three updates of the drift-parameter loop of lorenz-63-parameter-estimation.py:189-277,
    recorded from the reference (its smoothing, its energy_mode, its update): the loop logic of
    estimate_parameters_batched, driven by the oracle, must visit the same parameters, free
    energies and gradients.  The second problem starts elsewhere and must not disturb it."""
    from vgpa_amd.auxiliary.estimate import estimate_parameters_batched
    g = load_golden("estimate_l63_ut.npz")
    sde = sde_from_golden(g)
    m0, S0 = initial_moments(g)
    N, D, P = int(g["N"]), int(g["D"]), 2
    A0, b0 = orc.unpack_Ab(g["Ab"], N, D)
    scg_nit = int(g["scg_nit"])

    class OracleOps:
        def _sdes(self, theta):
            return [dict(sde, theta=theta[p].numpy().copy()) for p in range(P)]

        def smooth(self, theta, A, b):
            ev = _batched_evaluator("l63_ut", self._sdes(theta), m0, S0, N, D)
            return optimize.scg_batched(None, A, b, nit=scg_nit, evaluate=ev)[:3]

        def dth(self, theta, A, b):
            rows = []
            for p, s in enumerate(self._sdes(theta)):
                Ab = np.concatenate((A[p].numpy().ravel(), b[p].numpy().ravel()))
                rows.append(np.asarray(orc.evaluate("l63_ut", Ab, m0, S0, s, with_grad=False)["dEsde_dth"]).ravel())
            return torch.from_numpy(np.stack(rows))

    theta0 = torch.from_numpy(np.stack([g["thetas"][0], g["thetas"][0] * np.array([1.05, 0.9, 1.0])]))
    A = torch.from_numpy(np.broadcast_to(A0, (P,) + A0.shape).copy())
    b = torch.from_numpy(np.broadcast_to(b0, (P,) + b0.shape).copy())
    out = estimate_parameters_batched(None, A, b, rate=float(g["rate"]), nit=3, scg_nit=scg_nit, tol=0.0,
                                      ops=OracleOps(), theta0=theta0)
    assert relerr(out["thetas"][:, 0].numpy(), g["thetas"]) <= 1e-9
    assert relerr(out["costs"][:, 0].numpy(), g["costs"]) <= 1e-9
    assert relerr(out["grads"][:, 0].numpy(), g["grads"]) <= 1e-8
    assert relerr(out["thetas"][-1, 1].numpy(), g["thetas"][-1]) > 1e-3
