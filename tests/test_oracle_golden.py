"""The CPU oracle against fixtures produced by running the reference itself
(oracle/make_golden.py).  Tolerance 1e-12 max-norm relative: the oracle is a
restatement of the same fp64 arithmetic, only BLAS summation order differs."""
import numpy as np
import pytest

from conftest import load_golden, relerr, sde_from_golden, initial_moments
from oracle import vgpa_oracle as orc

TOL = 1e-12
SYSTEMS = [("ou", "sys_ou.npz"), ("dw", "sys_dw.npz"), ("l63", "sys_l63.npz"),
           ("l63_ut", "sys_l63_ut.npz"), ("l96_ut", "sys_l96_ut_d6.npz"),
           ("l96_ut", "sys_l96_ut_d40.npz"), ("l96_ut", "sys_l96_ut_d16.npz")]


@pytest.mark.parametrize("system,fname", SYSTEMS)
@pytest.mark.parametrize("method", orc.METHODS)
def test_full_evaluation(system, fname, method):
    g = load_golden(fname)
    sde = sde_from_golden(g, method)
    m0, S0 = initial_moments(g)
    out = orc.evaluate(system, g["Ab"], m0, S0, sde)
    assert abs(out["F"] - float(g[f"{method}.F"])) <= TOL * abs(float(g[f"{method}.F"]))
    assert relerr(out["grad"], g[f"{method}.grad"]) <= TOL
    for key in ("Eobs", "Esde", "KL0", "dEsde_dth", "dEsde_dSig"):
        assert relerr(out[key], g[f"{method}.{key}"]) <= TOL, key
    for key in ("m", "S", "lam", "Psi", "Ef", "Edf", "dEsde_dm", "dEsde_dS", "dEobs_dm"):
        gk = f"{method}.{key}"
        if gk in g:
            assert relerr(out[key], g[gk]) <= TOL, key
    if f"{method}.dEobs_dS_row" in g:
        row = g[f"{method}.dEobs_dS_row"]
        for t in sde["obsX"]:
            assert relerr(out["dEobs_dS"][t], row) <= TOL


def test_trapz_literal_and_weights():
    g = load_golden("units.npz")
    for f in ("f1", "f3"):
        fx = g[f"trapz.{f}"]
        assert relerr(orc.mytrapz(fx, 0.01), g[f"trapz.{f}.none"]) <= TOL
        w = orc.trapz_weights(fx.shape[0])
        assert relerr(0.01 * w.dot(fx), g[f"trapz.{f}.none"]) <= TOL
        for k in "abc":
            idx = g[f"trapz.idx.{k}"]
            ref = g[f"trapz.{f}.{k}"]
            assert relerr(np.asarray(orc.mytrapz(fx, 0.01, idx)).reshape(ref.shape), ref) <= TOL
            w = orc.trapz_weights(fx.shape[0], idx)
            assert relerr((0.01 * w.dot(fx)).reshape(ref.shape), ref) <= TOL


def test_mom_gauss_table():
    g = load_golden("units.npz")
    m, S = g["mom.m"], g["mom.S"]
    for order in range(9):
        for flag in (None, "Dm", "DS"):
            got = np.broadcast_to(orc.mom_gauss(m, S, order, flag), m.shape)
            assert relerr(got, g[f"mom.{order}.{flag}"]) <= TOL
    with pytest.raises(ValueError):
        orc.mom_gauss(m, S, 9)


def test_kl0_shape_dependence():
    g = load_golden("units.npz")
    sde = dict(D=5, px0=dict(mu0=g["kl0.mu0"], tau0=g["kl0.tau0"]))
    vec = orc.kl0_gauss(g["kl0.m0"], g["kl0.S0"], sde)
    row = orc.kl0_gauss(g["kl0.m0"][None, :], g["kl0.S0"], sde)
    assert relerr(vec, g["kl0.vec"]) <= TOL
    assert relerr(row, g["kl0.row"]) <= TOL
    assert abs(float(vec) - float(row)) > 1e-3       # the two shapes really differ
    sde1 = dict(D=1, px0=dict(mu0=0.3, tau0=0.5))
    assert relerr(orc.kl0_gauss(np.array([0.45]), 0.25, sde1), g["kl0.1d"]) <= TOL


@pytest.mark.parametrize("tag", ["l63", "l96", "l63_npd", "l96_npd"])
def test_unscented_single_step(tag):
    g = load_golden("units.npz")
    St, mt, At, bt = (g[f"ut.{tag}.{k}"] for k in ("St", "mt", "At", "bt"))
    dSigI, theta = g[f"ut.{tag}.dSigI"], g[f"ut.{tag}.theta"]
    D = mt.size
    drift = (lambda x: orc.lorenz63(x, theta)) if D == 3 else (lambda x: orc.lorenz96(x, float(theta)))
    chi, w, flag = orc.ut_sigma_points(mt, St)
    assert int(flag) == (1 if tag.endswith("npd") else 0)
    Fx = drift(chi)
    r2 = (Fx + chi.dot(At.T) - bt) ** 2
    assert relerr(w.dot(r2), g[f"ut.{tag}.mbar"]) <= TOL
    dmt, dSt = orc.grad_esde_ms(chi, Fx, mt, St, At, bt, dSigI)
    dmS = np.concatenate((w.dot(dmt), np.einsum("k,kij->ij", w, dSt).ravel()))
    assert relerr(dmS, g[f"ut.{tag}.dmS"]) <= 1e-11
    # the complete ut_approx, second output included
    f1 = lambda x, At, bt: (drift(x) + x.dot(At.T) - bt) ** 2   # noqa: E731
    ybar, Pyy = orc.ut_approx(f1, mt, St, At, bt)
    assert ybar.shape == g[f"ut.{tag}.mbar"].shape and Pyy.shape == g[f"ut.{tag}.Pyy"].shape
    assert relerr(ybar, g[f"ut.{tag}.mbar"]) <= TOL
    assert relerr(Pyy, g[f"ut.{tag}.Pyy"]) <= 1e-11


@pytest.mark.parametrize("system,fname", [("l63_ut", "l63_ut_nonpd.npz"), ("l96_ut", "l96_ut_nonpd_d6.npz"),
                                          ("l96_ut", "l96_ut_nonpd_d40.npz")])
def test_energy_mode_nonpd_fallback(system, fname):
    g = load_golden(fname)
    sde = sde_from_golden(g)
    N, D = sde["N"], sde["D"]
    A, b = orc.unpack_Ab(g["Ab"], N, D)
    out = orc.energy_mode(system, A, b, g["m"], g["S"], sde)
    names = ["Esde", "Ef", "Edf", "dEsde_dm", "dEsde_dS", "dEsde_dth", "dEsde_dSig"]
    for n, v in zip(names, out[:7]):
        assert relerr(v, g[n]) <= 1e-11, n
    assert out[7]["flag"].sum() == len(range(1, N, 3))


def test_unknown_method_raises():
    g = load_golden("sys_ou.npz")
    sde = sde_from_golden(g, "LEAPFROG")
    m0, S0 = initial_moments(g)
    with pytest.raises(ValueError, match="Unknown method of integration"):
        orc.evaluate("ou", g["Ab"], m0, S0, sde)


def test_observation_terms_dense_operators():
    """core/likelihood.py@108-179 with a dense H and a non-diagonal R."""
    g = load_golden("likelihood_dense.npz")
    sde = dict(D=int(g["D"]), N=int(g["N"]), dt=0.01, Rig=g["Rig"], H=g["H"], obsX=g["obsX"], obsY=g["obsY"])
    Eobs, dm, dS = orc.likelihood_gauss(g["m"], g["S"], sde)
    assert relerr(Eobs, g["Eobs"]) <= TOL
    assert relerr(dm, g["dEobs_dm"]) <= TOL and relerr(dS, g["dEobs_dS"]) <= TOL
    assert not g["dEobs_dR"].any()


def test_finite_differences_vs_reference():
    """auxiliary/numerical.py:89-150 on the analytic Lorenz-63 fixture.  A difference of two
    free energies of size 1e2 over 2e-6: absolute agreement to 1e-7."""
    g = load_golden("finite_diff_l63.npz")
    sde = sde_from_golden(g)
    m0, S0 = initial_moments(g)
    got = orc.finite_diff(lambda x: orc.evaluate("l63", x, m0, S0, sde, with_grad=False)["F"], g["Ab"])
    assert got.shape == g["finite_diff"].shape
    assert np.abs(got - g["finite_diff"]).max() <= 1e-7
