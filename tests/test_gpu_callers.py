"""The pieces either side of the hot path, on the device: the stand-alone unscented
transform, gradEsde_mS, momGauss, mytrapz, the optimisers driving the HIP path, the
batched finite differences and the module aliases for the reference's drivers."""
import contextlib
import io
import sys

import numpy as np
import pytest

from conftest import load_golden, relerr, sde_from_golden, initial_moments
from oracle import vgpa_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-9


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


@pytest.mark.parametrize("tag", ["l63", "l96", "l63_npd", "l96_npd"])
def test_ut_approx_and_grad_esde_ms_vs_golden(tag):
    from vgpa_amd.auxiliary.numerical import ut_approx
    from vgpa_amd.core.gradEsde_mS import gradEsde_mS
    g = load_golden("units.npz")
    St, mt, At, bt = (g[f"ut.{tag}.{k}"] for k in ("St", "mt", "At", "bt"))
    dSigI, theta = g[f"ut.{tag}.dSigI"], g[f"ut.{tag}.theta"]
    D = mt.size
    drift = (lambda x: orc.lorenz63(x, theta)) if D == 3 else (lambda x: orc.lorenz96(x, float(theta)))
    f1 = lambda x, At, bt: (drift(x) + x.dot(At.T) - np.tile(bt, (x.shape[0], 1))) ** 2   # noqa: E731
    f2 = lambda x, _: drift(x)                                                              # noqa: E731
    mbar, Pyy = ut_approx(f1, mt, St, At, bt)
    assert mbar.shape == g[f"ut.{tag}.mbar"].shape and Pyy.shape == g[f"ut.{tag}.Pyy"].shape
    assert relerr(mbar, g[f"ut.{tag}.mbar"]) <= TOL
    assert relerr(Pyy, g[f"ut.{tag}.Pyy"]) <= TOL
    # the gradient call of the energy functions: ut_approx over gradEsde_mS
    dmS, _ = ut_approx(gradEsde_mS, mt, St, f2, mt, St, At, bt, dSigI)
    assert dmS.shape == g[f"ut.{tag}.dmS"].shape
    assert relerr(dmS, g[f"ut.{tag}.dmS"]) <= TOL
    # and gradEsde_mS row by row against the oracle's literal version
    chi, w, _ = orc.ut_sigma_points(mt, St)
    rows = gradEsde_mS(chi, f2, mt, St, At, bt, dSigI)
    dmt, dSt = orc.grad_esde_ms(chi, drift(chi), mt, St, At, bt, dSigI)
    ref = np.concatenate((dmt, dSt.reshape(chi.shape[0], -1)), axis=1)
    assert rows.shape == ref.shape and relerr(rows, ref) <= TOL


def test_ut_approx_raises_when_no_factorisation_exists():
    from vgpa_amd.auxiliary.numerical import ut_approx
    P = np.diag([0.3, -0.2, 0.4])
    with pytest.raises(np.linalg.LinAlgError):
        ut_approx(lambda x: x, np.zeros(3), P)


def test_mom_gauss_table_vs_golden():
    from vgpa_amd.auxiliary.momGauss import momGauss
    g = load_golden("units.npz")
    m, S = g["mom.m"], g["mom.S"]
    for order in range(9):
        for flag in (None, "Dm", "DS"):
            got = momGauss(m, S, order, flag)
            want = g[f"mom.{order}.{flag}"]
            assert got.shape == want.shape, (order, flag)
            assert relerr(got, want) <= 1e-12, (order, flag)
    with pytest.raises(ValueError, match="Unknown Higher Order Gaussian Moment"):
        momGauss(m, S, 9)


def test_mytrapz_vs_golden():
    from vgpa_amd.auxiliary.numerical import mytrapz
    g = load_golden("units.npz")
    for col in ("f1", "f3"):
        f = g[f"trapz.{col}"]
        got = mytrapz(f, 0.01)
        assert np.shape(got) == g[f"trapz.{col}.none"].shape and relerr(got, g[f"trapz.{col}.none"]) <= 1e-12
        for k in ("a", "b", "c"):
            got = mytrapz(f, 0.01, g[f"trapz.idx.{k}"])
            assert relerr(got, g[f"trapz.{col}.{k}"]) <= 1e-12, (col, k)


def _facade_params(system, g):
    from vgpa_amd import dynamics   # noqa: F401
    import importlib
    mod = dict(ou="sys_ornstein_uhlenbeck", dw="sys_double_well", l63="sys_lorenz_63", l63_ut="sys_lorenz_63_ut",
               l96_ut="sys_lorenz_96_ut")[system]
    dyn = importlib.import_module("vgpa_amd.dynamics." + mod)
    sde = sde_from_golden(g)
    m0, S0 = initial_moments(g)
    return dict(m0=m0, S0=S0, fun=dyn.energy_mode, sde_struct=sde)


@pytest.mark.parametrize("system,fname,rows", [("ou", "scg_ou.npz", 30), ("l63", "scg_l63.npz", 25)])
def test_optim_scg_through_the_device_replays_the_reference_run(system, fname, rows):
    """optim_SCG(varFreeEnergy, x0, gradL_Ab, ...) exactly as core/smoothing.py:83 calls
    it, every evaluation on the GPU.  Rows past ``rows`` of the Ornstein-Uhlenbeck run
    are below the rounding error of F (the run has stalled) and are not compared."""
    from vgpa_amd.auxiliary.optimize import optim_SCG
    from vgpa_amd.core.varFreeEnergy import varFreeEnergy
    from vgpa_amd.core.gradL_Ab import gradL_Ab
    g = load_golden(fname)
    mParam = _facade_params(system, g)
    nit = int(g["nit"])
    x, fx, mParam, stat = _quiet(optim_SCG, varFreeEnergy, g["Ab"], gradL_Ab,
                                 dict(nit=nit, xtol=1e-6, ftol=1e-8, disp=False, lmin=True), mParam)
    assert relerr(stat["Fx"][:rows], g["Fx"][:rows]) <= TOL
    assert relerr(stat["beta"][:rows], g["beta"][:rows]) <= TOL
    assert relerr(stat["gradx"][:rows], g["gradx"][:rows]) <= 1e-7
    assert relerr(fx, g["fx"]) <= TOL
    if rows == nit:
        assert stat["f_eval"] == int(g["f_eval"]) and stat["g_eval"] == int(g["g_eval"])
        assert stat["MaxIt"] == int(g["MaxIt"])
        assert relerr(x, g["x"]) <= 1e-7
    assert "mt" in mParam and mParam["mt"].shape[0] == int(g["N"])


@pytest.mark.parametrize("system,fname,rows", [("ou", "scg_ou.npz", 30), ("l63", "scg_l63.npz", 25)])
def test_scg_batched_on_device_vs_reference_run(system, fname, rows):
    """Three copies of the recorded problem, the middle one with other observations:
    copies 0 and 2 must reproduce the recorded run, whatever copy 1 does."""
    import torch
    from vgpa_amd import engine, _lib
    from vgpa_amd.auxiliary.optimize import scg_batched
    g = load_golden(fname)
    sde = sde_from_golden(g)
    m0, S0 = initial_moments(g)
    N, D, P = int(g["N"]), int(g["D"]), 3
    obsY = np.array(sde["obsY"], dtype=np.float64).reshape(len(sde["obsX"]), -1)
    sde["obsY"] = np.stack([obsY, 0.5 * obsY + 0.2, obsY])
    prob = engine.make_problem(_lib.SYSTEMS[system], sde, m0, S0, P=P)
    A0, b0 = orc.unpack_Ab(g["Ab"], N, D)
    shpA, shpb = ((P, N), (P, N)) if D == 1 else ((P, N, D, D), (P, N, D))
    A = engine.to_dev(np.broadcast_to(A0.reshape(shpA[1:]), shpA))
    b = engine.to_dev(np.broadcast_to(b0.reshape(shpb[1:]), shpb))
    nit = int(g["nit"])
    Aout, bout, F, stat = scg_batched(prob, A, b, nit=nit)
    Fx, beta = stat["Fx"].cpu().numpy(), stat["beta"].cpu().numpy()
    for p in (0, 2):
        assert relerr(Fx[:rows, p], g["Fx"].ravel()[:rows]) <= TOL
        assert relerr(beta[:rows, p], g["beta"].ravel()[:rows]) <= TOL
        assert relerr(F[p].item(), g["fx"]) <= TOL
        if rows == nit:
            assert int(stat["f_eval"][p]) == int(g["f_eval"]) and int(stat["MaxIt"][p]) == int(g["MaxIt"])
            xp = np.concatenate((Aout[p].reshape(-1).cpu().numpy(), bout[p].reshape(-1).cpu().numpy()))
            assert relerr(xp, g["x"].ravel()) <= 1e-7
    assert relerr(Fx[:rows, 1], g["Fx"].ravel()[:rows]) > 1e-3
    assert torch.equal(stat["Fx"][:, 0], stat["Fx"][:, 2])   # identical problems, identical arithmetic


def test_finite_differences_batched_equal_one_by_one():
    from vgpa_amd.auxiliary.numerical import finiteDiff
    from vgpa_amd.core.varFreeEnergy import varFreeEnergy
    g = load_golden("sys_l63.npz")
    mParam = _facade_params("l63", g)
    x = np.array(g["Ab"], dtype=np.float64).reshape(-1, 1)
    got = _quiet(finiteDiff, varFreeEnergy, x, mParam)
    assert got.shape == x.shape
    rng = np.random.default_rng(0)
    h = 1e-6
    for i in rng.choice(x.size, 12, replace=False):
        e = np.zeros_like(x)
        e[i] = h
        want = 0.5 * (varFreeEnergy(x + e, dict(mParam)) - varFreeEnergy(x - e, dict(mParam))) / h
        assert abs(got[i, 0] - want) <= 1e-6 * max(1.0, abs(want))


def test_compat_aliases_bind_the_reference_module_names():
    import vgpa_amd.compat as compat
    import vgpa_amd.core.varFreeEnergy as ours
    saved = {k: v for k, v in sys.modules.items() if k.split(".")[0] in ("core", "dynamics", "auxiliary")}
    try:
        names = compat.install()
        assert "core.varFreeEnergy" in names and "dynamics.sys_lorenz_96_ut" in names
        from core.varFreeEnergy import varFreeEnergy          # the import lines of core/smoothing.py:5-10
        from core.gradL_Ab import gradL_Ab                     # noqa: F401
        from auxiliary.optimize import optim_SCG               # noqa: F401
        from dynamics import sys_lorenz_63 as dyn
        assert varFreeEnergy is ours.varFreeEnergy
        g = load_golden("sys_l63.npz")
        mParam = dict(m0=initial_moments(g)[0], S0=initial_moments(g)[1], fun=dyn.energy_mode,
                      sde_struct=sde_from_golden(g))
        assert abs(varFreeEnergy(g["Ab"], mParam) - float(g["EULER.F"])) <= TOL * abs(float(g["EULER.F"]))
    finally:
        compat.uninstall()
        for k in [k for k in sys.modules if k.split(".")[0] in ("core", "dynamics", "auxiliary")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_parameter_estimation_loop_batched():
    """The outer loop of lorenz-63-parameter-estimation.py:189-277 for a batch: every problem
    must follow the loop it would follow alone, and the gradient it descends is the
    dEsde_dth of ``energy_mode`` at the smoothed solution."""
    import torch
    from vgpa_amd import engine, _lib
    from vgpa_amd.auxiliary.estimate import estimate_parameters_batched
    from vgpa_amd.dynamics import sys_lorenz_63_ut
    g = load_golden("sys_l63_ut.npz")
    sde = sde_from_golden(g)
    m0, S0 = initial_moments(g)
    N, D = int(g["N"]), int(g["D"])
    A0, b0 = orc.unpack_Ab(g["Ab"], N, D)
    thetas = np.stack([np.array(sde["theta"]), np.array(sde["theta"]) * np.array([1.1, 0.95, 1.05])])
    dev = lambda x, P: engine.to_dev(np.broadcast_to(x, (P,) + x.shape))   # noqa: E731
    kw = dict(rate=1e-3, nit=2, scg_nit=6, tol=0.0)
    prob = engine.make_problem(_lib.SYS_L63_UT, sde, m0, S0, P=2, theta=thetas)
    out = estimate_parameters_batched(prob, dev(A0, 2), dev(b0, 2), **kw)
    assert out["costs"].shape == (3, 2) and out["grads"].shape == (2, 2, 3)
    assert torch.equal(out["iterations"].cpu(), torch.tensor([2, 2]))
    assert torch.equal(prob.theta.reshape(2, 3), out["theta"])
    for p in range(2):
        single = engine.make_problem(_lib.SYS_L63_UT, sde, m0, S0, P=1, theta=thetas[p:p + 1])
        one = estimate_parameters_batched(single, dev(A0, 1), dev(b0, 1), **kw)
        assert relerr(out["theta"][p].cpu().numpy(), one["theta"][0].cpu().numpy()) <= 1e-12
        assert relerr(out["costs"][:, p].cpu().numpy(), one["costs"][:, 0].cpu().numpy()) <= 1e-12
    # first update: theta1 = theta0 - rate * dEsde_dth(energy_mode at the first optimum)
    single = engine.make_problem(_lib.SYS_L63_UT, sde, m0, S0, P=1, theta=thetas[1:2])
    first = estimate_parameters_batched(single, dev(A0, 1), dev(b0, 1), rate=1e-3, nit=0, scg_nit=6)
    st = engine.free_energy(single, first["A"], first["b"], want_hyper=False)
    sde1 = dict(sde, theta=thetas[1])
    dth = sys_lorenz_63_ut.energy_mode(first["A"][0].cpu().numpy(), first["b"][0].cpu().numpy(),
                                       st.m[0].cpu().numpy(), st.S[0].cpu().numpy(), sde1)[5]
    assert relerr(out["grads"][0, 1].cpu().numpy(), np.asarray(dth).ravel()) <= TOL
    step = out["grads"][0, 1].cpu().numpy() * 1e-3
    assert step.max() > 0 or step.min() < 0
    second = thetas[1] - step
    # the parameters after the first of the two updates are not returned; check through the
    # second gradient being evaluated at them: rerun one update and compare
    one_update = estimate_parameters_batched(
        engine.make_problem(_lib.SYS_L63_UT, sde, m0, S0, P=1, theta=thetas[1:2]), dev(A0, 1), dev(b0, 1),
        rate=1e-3, nit=1, scg_nit=6, tol=0.0)
    assert relerr(one_update["theta"][0].cpu().numpy(), second) <= 1e-12


def test_parameter_estimation_loop_vs_reference_run():
    """Three updates of lorenz-63-parameter-estimation.py:189-277 recorded from the reference
    (tests/golden/estimate_l63_ut.npz: the reference's smoothing, energy_mode and update):
    the device loop must visit the same parameters, free energies and gradients."""
    from vgpa_amd import engine, _lib
    from vgpa_amd.auxiliary.estimate import estimate_parameters_batched
    g = load_golden("estimate_l63_ut.npz")
    sde = sde_from_golden(g)
    m0, S0 = initial_moments(g)
    N, D, P = int(g["N"]), int(g["D"]), 2
    A0, b0 = orc.unpack_Ab(g["Ab"], N, D)
    thetas = np.stack([g["thetas"][0] * np.array([1.05, 0.9, 1.0]), g["thetas"][0]])
    dev = lambda x: engine.to_dev(np.broadcast_to(x, (P,) + x.shape))   # noqa: E731
    prob = engine.make_problem(_lib.SYS_L63_UT, sde, m0, S0, P=P, theta=thetas)
    out = estimate_parameters_batched(prob, dev(A0), dev(b0), rate=float(g["rate"]), nit=3,
                                      scg_nit=int(g["scg_nit"]), tol=0.0)
    assert relerr(out["thetas"][:, 1].cpu().numpy(), g["thetas"]) <= TOL
    assert relerr(out["costs"][:, 1].cpu().numpy(), g["costs"]) <= TOL
    assert relerr(out["grads"][:, 1].cpu().numpy(), g["grads"]) <= 1e-8
    assert relerr(out["theta"][1].cpu().numpy(), g["thetas"][-1]) <= TOL


def test_finite_differences_vs_reference_vector():
    """finiteDiff(varFreeEnergy, x, mParam) recorded from the reference
    (auxiliary/numerical.py:89-150, 984 evaluations) against the batched device version.
    Each entry is a difference of two free energies of size 1e2 over 2e-6, so a relative
    error of 1e-12 in F allows 1e-4 * 1e-12 / 1e-6 ... in absolute terms 1e-7."""
    from vgpa_amd.auxiliary.numerical import finiteDiff
    from vgpa_amd.core.varFreeEnergy import varFreeEnergy
    g = load_golden("finite_diff_l63.npz")
    mParam = _facade_params("l63", g)
    got = _quiet(finiteDiff, varFreeEnergy, np.array(g["Ab"], dtype=np.float64), mParam)
    assert got.shape == g["finite_diff"].shape
    assert np.abs(got - g["finite_diff"]).max() <= 1e-7


def test_captured_evaluation_replays_the_eager_one():
    """engine.CapturedEvaluation: the evaluation as a HIP graph gives the eager results bit for bit,
    also after the parameters were changed in place, for a 1-D system and for Lorenz-63."""
    import torch
    from vgpa_amd import engine, _lib
    for fname, system in (("sys_dw.npz", _lib.SYS_DW), ("sys_l63_ut.npz", _lib.SYS_L63_UT)):
        g = load_golden(fname)
        sde = sde_from_golden(g, "EULER")
        m0, S0 = initial_moments(g)
        N, D = sde["N"], sde["D"]
        A, b = orc.unpack_Ab(g["Ab"], N, D)
        prob = engine.make_problem(system, sde, m0, S0, P=1)
        Ad = engine.to_dev(A.reshape((1, N) if D == 1 else (1, N, D, D)))
        bd = engine.to_dev(b.reshape((1, N) if D == 1 else (1, N, D)))
        cap = engine.CapturedEvaluation(prob, Ad, bd, want_hyper=True)
        for scale in (1.0, 1.05):
            Ad.copy_(engine.to_dev(scale * A.reshape(Ad.shape)))
            st, gA, gb = cap.replay()
            torch.cuda.synchronize()
            F, gAc, gbc = st.F.clone(), gA.clone(), gb.clone()
            st2, gA2, gb2 = engine.free_energy_and_gradient(prob, Ad, bd, want_hyper=True)
            assert torch.equal(F, st2.F) and torch.equal(gAc, gA2) and torch.equal(gbc, gb2)
            assert torch.equal(st.dth, st2.dth)
            if scale == 1.0:
                assert abs(float(F[0]) - float(g["EULER.F"])) <= 1e-9 * abs(float(g["EULER.F"]))
