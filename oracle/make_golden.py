#!/usr/bin/env python3
"""Generate golden input/output vectors by running the reference implementation.

TEST INFRASTRUCTURE ONLY.  This script imports the NumPy reference from
``/root/reference`` (read-only, never copied) and records, for seeded random
inputs, every output and intermediate of the free-energy / gradient path into
small ``.npz`` files under ``tests/golden/``.  It runs only in the build
container (the reference does not exist on the GPU box); the committed
fixtures are what travels.

The reference targets an old NumPy, so three removed names are patched in
before import (``np.asscalar``, ``np.int``, ``np.math``); no reference file is
modified.

Usage:  python oracle/make_golden.py [--out tests/golden]
"""
import argparse
import math
import os
import sys
import warnings
from copy import deepcopy

import numpy as np

REF = "/root/reference"

METHODS = ["EULER", "HEUN", "RK2", "RK4"]


def _import_reference():
    np.asscalar = lambda a: np.asarray(a).item()  # noqa: E731
    np.int = int
    np.math = math
    import matplotlib
    matplotlib.use("Agg")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    warnings.simplefilter("ignore")
    mods = {}
    from core import varFreeEnergy, gradL_Ab, fwd_ode, bwd_ode, likelihood, kl0, gradEsde_mS
    from auxiliary import numerical, momGauss, optimize
    from dynamics import (sys_ornstein_uhlenbeck, sys_double_well, sys_lorenz_63,
                          sys_lorenz_63_ut, sys_lorenz_96_ut)
    mods.update(varFreeEnergy=varFreeEnergy, gradL_Ab=gradL_Ab, fwd_ode=fwd_ode,
                bwd_ode=bwd_ode, likelihood=likelihood, kl0=kl0, gradEsde_mS=gradEsde_mS,
                numerical=numerical, momGauss=momGauss, optimize=optimize,
                ou=sys_ornstein_uhlenbeck, dw=sys_double_well, l63=sys_lorenz_63,
                l63_ut=sys_lorenz_63_ut, l96_ut=sys_lorenz_96_ut)
    return mods


# --------------------------------------------------------------------------
# Problem builders (inputs are stored in the fixture, so no RNG dependence
# remains on the consumer side).
# --------------------------------------------------------------------------
def _obs_indices(N, rng, M):
    """Irregular, sorted, unique observation indices in [1, N-2]; the first
    two are adjacent so one trapezoid segment has a single sample."""
    pool = np.arange(2, N - 1)
    idx = np.sort(rng.choice(pool, size=M, replace=False))
    if idx[1] != idx[0] + 1:
        idx[1] = idx[0] + 1
        idx = np.unique(idx)
    return idx.astype(np.int64)


def build_1d(system, N, seed):
    rng = np.random.default_rng(seed)
    dt = 0.01
    if system == "ou":
        Sig, theta = 1.0, 2.0
    else:
        Sig, theta = 0.8, 1.0
    R = 0.04
    S0 = 0.25
    tau0 = 0.5
    a0 = 0.5 * Sig / S0
    A = a0 + 0.3 * rng.standard_normal((N, 1))
    b = 1.5 * rng.standard_normal((N, 1))
    obsX = _obs_indices(N, rng, max(4, N // 12))
    obsY = 1.2 * rng.standard_normal((obsX.size, 1))
    mu0 = float(rng.standard_normal())
    m0 = np.array([mu0 + 0.1 * rng.standard_normal()])
    sde = dict(Sig=Sig, theta=theta, Rig=R, D=1, H=1.0, obsX=obsX, obsY=obsY,
               px0=dict(mu0=mu0, tau0=tau0), dt=dt, N=N, ode_method="EULER",
               checkGradf=False, Tw=np.arange(N) * dt)
    Ab = np.concatenate((A, b))
    return Ab, m0, S0, sde


def build_nd(system, D, N, seed, dt=0.01):
    rng = np.random.default_rng(seed)
    if system in ("l63", "l63_ut"):
        Sig = 10.0 * np.eye(D)
        theta = np.array([10.0, 28.0, 8.0 / 3.0])
        R = 2.0 * np.eye(D)
        bscale, mscale = 5.0, 3.0
    else:
        Sig = 4.0 * np.eye(D)
        theta = 8.0
        R = 1.0 * np.eye(D)
        bscale, mscale = 5.0, 2.0
    S0 = 0.25 * np.eye(D) + 0.02 * _sym(rng.standard_normal((D, D))) / np.sqrt(D)
    a0 = 0.5 * Sig[0, 0] / 0.25
    A = a0 * np.eye(D)[None] + 0.3 * rng.standard_normal((N, D, D))
    b = bscale * rng.standard_normal((N, D))
    obsX = _obs_indices(N, rng, max(3, N // 6))
    obsY = mscale * rng.standard_normal((obsX.size, D))
    mu0 = mscale * rng.standard_normal(D)
    m0 = mu0 + 0.1 * rng.standard_normal(1)
    tau0 = 0.5 * np.eye(D)
    sde = dict(Sig=Sig, theta=theta, Rig=R, D=D, H=np.eye(D), obsX=obsX, obsY=obsY,
               px0=dict(mu0=mu0, tau0=tau0), dt=dt, N=N, ode_method="EULER",
               checkGradf=False, Tw=np.arange(N) * dt)
    Ab = np.concatenate((A.ravel()[:, None], b.ravel()[:, None]))
    return Ab, m0, S0, sde


def _sym(X):
    return 0.5 * (X + X.T)


def _flatten_inputs(Ab, m0, S0, sde):
    out = dict(Ab=Ab, m0=np.asarray(m0), S0=np.asarray(S0), Sig=np.asarray(sde["Sig"]),
               theta=np.asarray(sde["theta"]), Rig=np.asarray(sde["Rig"]),
               D=np.int64(sde["D"]), N=np.int64(sde["N"]), H=np.asarray(sde["H"]),
               obsX=np.asarray(sde["obsX"]), obsY=np.asarray(sde["obsY"]),
               mu0=np.asarray(sde["px0"]["mu0"]), tau0=np.asarray(sde["px0"]["tau0"]),
               dt=np.float64(sde["dt"]))
    return out


def run_full(ref, system, Ab, m0, S0, sde, method, full=True):
    """One free-energy + gradient evaluation through the reference, with all
    intermediates recomputed through the reference's own sub-functions."""
    sde = deepcopy(sde)
    sde["ode_method"] = method
    dyn = ref[system]
    mParam = dict(m0=m0, S0=S0, fun=dyn.energy_mode, sde_struct=sde)
    F = ref["varFreeEnergy"].varFreeEnergy(Ab, mParam)
    g = ref["gradL_Ab"].gradL_Ab(Ab, mParam)
    g2 = ref["gradL_Ab"].gradL_Ab(Ab, mParam, True)
    assert np.array_equal(g, g2)
    out = dict(F=np.float64(F), grad=g)
    N, D = sde["N"], sde["D"]
    if D == 1:
        A, b = Ab[:N], Ab[N:]
    else:
        A, b = Ab[:N * D * D].reshape(N, D, D), Ab[N * D * D:].reshape(N, D)
    m, S = mParam["mt"], mParam["St"]
    Eobs, dEobs_dm, dEobs_dS, _ = ref["likelihood"].gauss(m, S, sde)
    Esde, Ef, Edf, dEdm, dEdS, dth, dSig = dyn.energy_mode(A, b, m, S, sde)
    KL0, _, _ = ref["kl0"].gauss(m0, S0, mParam["lamt"], mParam["Psit"], sde)
    out.update(Eobs=np.asarray(Eobs), Esde=np.asarray(Esde), KL0=np.asarray(KL0),
               dEsde_dth=np.asarray(dth), dEsde_dSig=np.asarray(dSig))
    if full:
        out.update(m=m, S=S, lam=mParam["lamt"], Psi=mParam["Psit"], Ef=Ef, Edf=Edf,
                   dEsde_dm=dEdm, dEsde_dS=dEdS, dEobs_dm=dEobs_dm)
        # dEobs_dS is zero except at observation rows, all equal: keep one row.
        out["dEobs_dS_row"] = np.asarray(dEobs_dS[sde["obsX"][0]])
        assert all(np.array_equal(dEobs_dS[t], dEobs_dS[sde["obsX"][0]]) for t in sde["obsX"])
        mask = np.ones(N, bool)
        mask[sde["obsX"]] = False
        assert not np.any(dEobs_dS[mask])
    return out


def golden_systems(ref, outdir):
    cases = [
        ("ou", lambda: build_1d("ou", 101, 11), True),
        ("dw", lambda: build_1d("dw", 101, 12), True),
        ("l63", lambda: build_nd("l63", 3, 41, 13), True),
        ("l63_ut", lambda: build_nd("l63_ut", 3, 41, 14), True),
        ("l96_ut", lambda: build_nd("l96_ut", 6, 21, 15), True),     # small-D Lorenz-96
    ]
    for system, builder, full in cases:
        Ab, m0, S0, sde = builder()
        rec = _flatten_inputs(Ab, m0, S0, sde)
        for method in METHODS:
            res = run_full(ref, system, Ab, m0, S0, sde, method, full=full)
            for k, v in res.items():
                rec[f"{method}.{k}"] = v
        tag = system if system != "l96_ut" else "l96_ut_d6"
        np.savez_compressed(os.path.join(outdir, f"sys_{tag}.npz"), **rec)
        print("wrote", tag, {m: rec[f"{m}.F"] for m in METHODS})

    # Lorenz-96 at the headline dimension D=40 (N=7): full intermediates for
    # Euler, F + gradient only for the other integrators (keeps the file small).
    Ab, m0, S0, sde = build_nd("l96_ut", 40, 7, 16, dt=0.005)
    rec = _flatten_inputs(Ab, m0, S0, sde)
    for method in METHODS:
        res = run_full(ref, "l96_ut", Ab, m0, S0, sde, method, full=(method == "EULER"))
        for k, v in res.items():
            if method == "EULER" and k == "Edf":
                continue  # reproducible from m; checked in the D=6 case
            rec[f"{method}.{k}"] = v
    np.savez_compressed(os.path.join(outdir, "sys_l96_ut_d40.npz"), **rec)
    print("wrote l96_ut_d40", {m: rec[f"{m}.F"] for m in METHODS})


def golden_units(ref, outdir):
    num = ref["numerical"]
    rng = np.random.default_rng(101)
    rec = {}

    # mytrapz: with / without observation indices, 1 and several columns.
    f1 = rng.standard_normal((37, 1))
    f3 = rng.standard_normal((37, 3))
    idxs = {"a": np.array([3, 4, 11, 30]), "b": np.array([1, 2, 3, 36]), "c": np.array([5])}
    rec["trapz.f1"], rec["trapz.f3"] = f1, f3
    rec["trapz.f1.none"] = np.asarray(num.mytrapz(f1, 0.01))
    rec["trapz.f3.none"] = np.asarray(num.mytrapz(f3, 0.01))
    for k, idx in idxs.items():
        rec[f"trapz.idx.{k}"] = idx
        rec[f"trapz.f1.{k}"] = np.asarray(num.mytrapz(f1, 0.01, idx))
        rec[f"trapz.f3.{k}"] = np.asarray(num.mytrapz(f3, 0.01, idx))

    # momGauss: every order and derivative flag.
    mg = ref["momGauss"].momGauss
    m = rng.standard_normal((9, 1))
    S = 0.1 + rng.random((9, 1))
    rec["mom.m"], rec["mom.S"] = m, S
    for order in range(9):
        for flag in (None, "Dm", "DS"):
            rec[f"mom.{order}.{flag}"] = np.asarray(mg(m, S, order, flag), dtype=np.float64)

    # KL0 with (D,) and (1,D) initial means (the result depends on the shape).
    D = 5
    S0 = 0.25 * np.eye(D) + 0.03 * _sym(rng.standard_normal((D, D)))
    tau0 = 0.5 * np.eye(D) + 0.05 * _sym(rng.standard_normal((D, D)))
    mu0 = rng.standard_normal(D)
    m0 = mu0 + 0.3 * rng.standard_normal(D)
    lam = np.zeros((2, D))
    Psi = np.zeros((2, D, D))
    sde = dict(D=D, px0=dict(mu0=mu0, tau0=tau0))
    rec["kl0.S0"], rec["kl0.tau0"], rec["kl0.mu0"], rec["kl0.m0"] = S0, tau0, mu0, m0
    rec["kl0.vec"] = np.asarray(ref["kl0"].gauss(m0, S0, lam, Psi, sde)[0])
    rec["kl0.row"] = np.asarray(ref["kl0"].gauss(m0[None, :], S0, lam, Psi, sde)[0])
    sde1 = dict(D=1, px0=dict(mu0=0.3, tau0=0.5))
    rec["kl0.1d"] = np.asarray(ref["kl0"].gauss(np.array([0.45]), 0.25, np.zeros((2, 1)),
                                                 np.zeros((2, 1)), sde1)[0])

    # ut_approx + gradEsde_mS single-step vectors at D=3 and D=40, plus a
    # non-positive-definite covariance at D=3 and D=6 (diagonal fallback).
    for tag, dyn, D in (("l63", ref["l63_ut"], 3), ("l96", ref["l96_ut"], 40),
                        ("l63_npd", ref["l63_ut"], 3), ("l96_npd", ref["l96_ut"], 6)):
        theta = np.array([10.0, 28.0, 8.0 / 3.0]) if D == 3 else 8.0
        drift = dyn.lorenz63 if D == 3 else dyn.lorenz96
        X = rng.standard_normal((D, D))
        St = 0.3 * np.eye(D) + 0.1 * X.dot(X.T) / D
        if tag.endswith("npd"):
            St = St - 0.6 * np.outer(np.ones(D), np.ones(D)) / D  # one negative eigenvalue
            St[np.diag_indices(D)] = np.abs(St[np.diag_indices(D)]) + 0.05
            assert np.linalg.eigvalsh(St).min() < 0
        mt = 2.0 * rng.standard_normal(D)
        At = 8.0 * np.eye(D) + 0.3 * rng.standard_normal((D, D))
        bt = 3.0 * rng.standard_normal(D)
        dSigI = 1.0 / (2.0 + rng.random(D))
        f1 = lambda x, At, bt: (drift(x, theta) + x.dot(At.T) - np.tile(bt, (x.shape[0], 1))) ** 2  # noqa
        f2 = lambda x, _: drift(x, theta)  # noqa: E731
        mbar, Pyy = num.ut_approx(f1, mt, St, At, bt)
        dmS, _ = num.ut_approx(ref["gradEsde_mS"].gradEsde_mS, mt, St, f2, mt, St, At, bt, dSigI)
        rec.update({f"ut.{tag}.St": St, f"ut.{tag}.mt": mt, f"ut.{tag}.At": At,
                    f"ut.{tag}.bt": bt, f"ut.{tag}.dSigI": dSigI, f"ut.{tag}.theta": np.asarray(theta),
                    f"ut.{tag}.mbar": mbar, f"ut.{tag}.Pyy": Pyy, f"ut.{tag}.dmS": dmS})
    np.savez_compressed(os.path.join(outdir, "units.npz"), **rec)
    print("wrote units")


def golden_nonpd_energy(ref, outdir):
    """A Lorenz-63 (unscented) energy_mode call where some S(t) are not
    positive definite, so the diagonal Cholesky fallback is taken."""
    rng = np.random.default_rng(202)
    D, N = 3, 12
    Ab, m0, S0, sde = build_nd("l63_ut", D, N, 17)
    A = Ab[:N * D * D].reshape(N, D, D)
    b = Ab[N * D * D:].reshape(N, D)
    m = 2.0 * rng.standard_normal((N, D))
    S = np.zeros((N, D, D))
    for t in range(N):
        X = rng.standard_normal((D, D))
        S[t] = 0.3 * np.eye(D) + 0.1 * X.dot(X.T)
        if t % 3 == 1:
            S[t] -= 0.5 * np.ones((D, D))
            S[t][np.diag_indices(D)] = np.abs(S[t][np.diag_indices(D)]) + 0.05
    out = ref["l63_ut"].energy_mode(A, b, m, S, sde)
    names = ["Esde", "Ef", "Edf", "dEsde_dm", "dEsde_dS", "dEsde_dth", "dEsde_dSig"]
    rec = _flatten_inputs(Ab, m0, S0, sde)
    rec.update(m=m, S=S)
    rec.update({n: np.asarray(v) for n, v in zip(names, out)})
    np.savez_compressed(os.path.join(outdir, "l63_ut_nonpd.npz"), **rec)
    print("wrote l63_ut_nonpd")

    # The same for Lorenz-96 at a small and at the benchmark dimension.
    for D, N, seed in ((6, 12, 31), (40, 6, 32)):
        rng = np.random.default_rng(300 + D)
        Ab, m0, S0, sde = build_nd("l96_ut", D, N, seed)
        A = Ab[:N * D * D].reshape(N, D, D)
        b = Ab[N * D * D:].reshape(N, D)
        m = 2.0 * rng.standard_normal((N, D))
        S = np.zeros((N, D, D))
        for t in range(N):
            X = rng.standard_normal((D, D))
            S[t] = 0.3 * np.eye(D) + 0.1 * X.dot(X.T) / D
            if t % 3 == 1:
                S[t] -= 0.6 * np.ones((D, D)) / D
                S[t][np.diag_indices(D)] = np.abs(S[t][np.diag_indices(D)]) + 0.05
                assert np.linalg.eigvalsh(S[t]).min() < 0
        out = ref["l96_ut"].energy_mode(A, b, m, S, sde)
        rec = _flatten_inputs(Ab, m0, S0, sde)
        rec.update(m=m, S=S)
        rec.update({n: np.asarray(v) for n, v in zip(names, out)})
        np.savez_compressed(os.path.join(outdir, f"l96_ut_nonpd_d{D}.npz"), **rec)
        print(f"wrote l96_ut_nonpd_d{D}")


def golden_scg(ref, outdir):
    """One SCG smoothing run on the Ornstein-Uhlenbeck configuration
    (T=10, dt=0.01): pins that an optimiser driving this path follows the same
    trajectory."""
    import io
    import contextlib
    N = 1001
    Ab, m0, S0, sde = build_1d("ou", N, 21)
    # A smooth start (constant A, zero b) as a spline initialisation would give.
    Ab = np.concatenate((np.full((N, 1), 0.5 * sde["Sig"] / S0), np.zeros((N, 1))))
    rng = np.random.default_rng(22)
    M = 20
    sde["obsX"] = np.linspace(0, N, M + 2, dtype=int)[1:-1]
    sde["obsY"] = 0.7 * rng.standard_normal((M, 1))
    mParam = dict(m0=m0, S0=S0, fun=ref["ou"].energy_mode, sde_struct=sde)
    nit = 60
    opts = dict(nit=nit, xtol=1e-6, ftol=1e-8, disp=False, lmin=True)
    with contextlib.redirect_stdout(io.StringIO()):
        x, fx, mParam, stat = ref["optimize"].optim_SCG(
            ref["varFreeEnergy"].varFreeEnergy, Ab, ref["gradL_Ab"].gradL_Ab, opts, mParam)
    rec = _flatten_inputs(Ab, m0, S0, sde)
    rec.update(nit=np.int64(nit), x=x, fx=np.float64(fx), Fx=stat["Fx"], beta=stat["beta"],
               gradx=stat["gradx"], f_eval=np.int64(stat["f_eval"]),
               g_eval=np.int64(stat["g_eval"]), MaxIt=np.int64(stat["MaxIt"]))
    np.savez_compressed(os.path.join(outdir, "scg_ou.npz"), **rec)
    print("wrote scg_ou: fx", fx, "f_eval", stat["f_eval"], "MaxIt", stat["MaxIt"])

    # A short n-D run (analytic Lorenz-63, N=41) from the fixture's own start.
    D, N = 3, 41
    Ab, m0, S0, sde = build_nd("l63", D, N, 23)
    mParam = dict(m0=m0, S0=S0, fun=ref["l63"].energy_mode, sde_struct=sde)
    nit = 25
    opts = dict(nit=nit, xtol=1e-6, ftol=1e-8, disp=False, lmin=True)
    with contextlib.redirect_stdout(io.StringIO()):
        x, fx, mParam, stat = ref["optimize"].optim_SCG(
            ref["varFreeEnergy"].varFreeEnergy, Ab, ref["gradL_Ab"].gradL_Ab, opts, mParam)
    rec = _flatten_inputs(Ab, m0, S0, sde)
    rec.update(nit=np.int64(nit), x=x, fx=np.float64(fx), Fx=stat["Fx"], beta=stat["beta"],
               gradx=stat["gradx"], f_eval=np.int64(stat["f_eval"]),
               g_eval=np.int64(stat["g_eval"]), MaxIt=np.int64(stat["MaxIt"]))
    np.savez_compressed(os.path.join(outdir, "scg_l63.npz"), **rec)
    print("wrote scg_l63: fx", fx, "f_eval", stat["f_eval"], "MaxIt", stat["MaxIt"])


def golden_prep(ref, outdir):
    """Input preparation: seeded sample paths of every system, observations collected from
    them, and the starting values A0, b0.  The fixture holds the seed and the arguments;
    the consumer re-seeds ``numpy.random`` and must obtain the same arrays."""
    import contextlib
    import io
    from auxiliary import collect_obs as ref_collect, initialize_Ab0 as ref_init
    rec = {}
    quiet = contextlib.redirect_stdout(io.StringIO())
    T = np.arange(0.0, 0.6, 0.01)
    with quiet:
        np.random.seed(101)
        rec["ou_T"], rec["ou_sig"], rec["ou_theta"] = T, np.float64(0.8), np.float64(2.0)
        rec["ou_X"] = ref["ou"].system_path(T, 0.8, 2.0)
        np.random.seed(102)
        rec["dw_T"], rec["dw_sig"], rec["dw_theta"] = T, np.float64(0.8), np.float64(1.0)
        rec["dw_X"] = ref["dw"].system_path(T, 0.8, 1.0)
        th63 = np.array([10.0, 28.0, 8.0 / 3.0])
        sig63 = np.diag([6.0, 7.0, 8.0])
        np.random.seed(103)
        rec["l63_T"], rec["l63_sig"], rec["l63_theta"] = T, sig63, th63
        rec["l63_X"] = ref["l63"].system_path(T, sig63, th63)
        np.random.seed(104)
        rec["l63_ut_X"] = ref["l63_ut"].system_path(T, sig63, th63)
        D = 40
        sig96 = 4.0 * np.eye(D)
        np.random.seed(105)
        rec["l96_T"], rec["l96_sig"], rec["l96_theta"] = T, sig96, np.float64(8.0)
        rec["l96_X"] = ref["l96_ut"].system_path(T, D, sig96, 8.0)
        np.random.seed(106)
        rec["l96s_X"] = ref["l96_ut"].system_path(T, 6, 2.5, 8.0)         # scalar noise variance, D = 6
    # observations: scalar R (one random number per time), vector R with a mask
    Tl = np.arange(0.0, 2.0, 0.01)
    with quiet:
        np.random.seed(107)
        path = ref["l63_ut"].system_path(Tl, sig63, th63)
    rec["obs_T"], rec["obs_path"] = Tl, path
    np.random.seed(108)
    oX, oY = ref_collect.collect_obs(path, Tl, 4, 2.0)
    rec["obs1_X"], rec["obs1_Y"] = np.asarray(oX, dtype=np.int64), oY
    np.random.seed(109)
    oX2, oY2 = ref_collect.collect_obs(path, Tl, 2, np.array([1.0, 2.0]), [0, 2])
    rec["obs2_X"], rec["obs2_Y"] = np.asarray(oX2, dtype=np.int64), oY2
    # starting values, D = 3 and D = 1
    S0 = np.diag([0.25, 0.5, 0.75])
    sde = dict(Sig=sig63, N=len(Tl), D=3, Tw=Tl, obsX=oX, obsY=oY, dt=0.01)
    rec["ab0_S0"], rec["ab0_vec"] = S0, ref_init.initialize_Ab0(S0, sde)
    sde1 = dict(Sig=0.8, N=len(Tl), D=1, Tw=Tl, obsX=oX, obsY=oY[:, :1], dt=0.01)
    rec["ab0_1d_S0"], rec["ab0_1d_vec"] = np.float64(0.3), ref_init.initialize_Ab0(0.3, sde1)
    np.savez_compressed(os.path.join(outdir, "prep.npz"), **rec)
    print("wrote prep:", {k: v.shape for k, v in rec.items() if hasattr(v, "shape") and v.ndim})


def golden_callers(ref, outdir):
    """Two callers of the path, recorded from the reference:

    * three updates of the drift-parameter estimation loop of
      lorenz-63-parameter-estimation.py:189-277 (smoothing with the reference's
      ``smoothing.smoothing``, dEsde/dtheta from ``energy_mode`` at the optimum, theta
      <- theta - rate * gradient), on the N = 41 unscented Lorenz-63 fixture.  The
      script's ``run_smoothing`` draws a random m0 and builds spline starting values;
      here both are the fixture's fixed arrays so that the run can be repeated.
    * ``finiteDiff(varFreeEnergy, x, mParam)`` (auxiliary/numerical.py:89-150) on the
      N = 41 analytic Lorenz-63 fixture: 2 n evaluations of the free energy.
    """
    import contextlib
    import io
    from core import smoothing
    quiet = contextlib.redirect_stdout(io.StringIO())

    Ab0, m0, S0, sde = build_nd("l63_ut", 3, 41, 14)          # the inputs of sys_l63_ut.npz
    dyn = ref["l63_ut"]
    theta = np.array(sde["theta"]) * np.array([0.9, 1.05, 1.1])
    rate, nit_outer, scg_nit = 1.0e-2, 3, 8
    thetas, costs, grads = [theta.copy()], [], []

    def run_smoothing(theta_now):
        s = deepcopy(sde)
        s["theta"] = theta_now.copy()
        with quiet:
            Fmin, mParam = smoothing.smoothing(dyn.energy_mode, Ab0.copy(), m0, S0, s, scg_nit)
        return Fmin, mParam, s

    Fmin, mParam, s = run_smoothing(theta)
    costs.append(Fmin)
    for _ in range(nit_outer):
        dth = dyn.energy_mode(mParam["At"], mParam["bt"], mParam["mt"], mParam["St"], s)[5]
        grads.append(np.asarray(dth, dtype=np.float64).reshape(-1).copy())
        theta = theta - rate * grads[-1]
        thetas.append(theta.copy())
        Fmin, mParam, s = run_smoothing(theta)
        costs.append(Fmin)
    rec = _flatten_inputs(Ab0, m0, S0, sde)
    rec.update(rate=np.float64(rate), scg_nit=np.int64(scg_nit), thetas=np.array(thetas),
               costs=np.array(costs, dtype=np.float64), grads=np.array(grads))
    np.savez_compressed(os.path.join(outdir, "estimate_l63_ut.npz"), **rec)
    print("wrote estimate_l63_ut: costs", costs, "theta", thetas[-1])

    Ab, m0, S0, sde = build_nd("l63", 3, 41, 13)              # the inputs of sys_l63.npz
    mParam = dict(m0=m0, S0=S0, fun=ref["l63"].energy_mode, sde_struct=deepcopy(sde))
    with quiet:
        gnum = ref["numerical"].finiteDiff(ref["varFreeEnergy"].varFreeEnergy, Ab.copy(), mParam)
    rec = _flatten_inputs(Ab, m0, S0, sde)
    rec.update(finite_diff=np.asarray(gnum, dtype=np.float64))
    np.savez_compressed(os.path.join(outdir, "finite_diff_l63.npz"), **rec)
    print("wrote finite_diff_l63:", np.asarray(gnum).shape, float(np.abs(gnum).max()))


def golden_extras(ref, outdir):
    """Lorenz-96 at D = 16 (the dimension of the workgroup-per-step energy kernel) and the
    observation terms with a dense H and a non-diagonal R."""
    Ab, m0, S0, sde = build_nd("l96_ut", 16, 9, 41, dt=0.005)
    rec = _flatten_inputs(Ab, m0, S0, sde)
    for method in METHODS:
        res = run_full(ref, "l96_ut", Ab, m0, S0, sde, method, full=(method == "EULER"))
        for k, v in res.items():
            rec[f"{method}.{k}"] = v
    np.savez_compressed(os.path.join(outdir, "sys_l96_ut_d16.npz"), **rec)
    print("wrote l96_ut_d16", {m: rec[f"{m}.F"] for m in METHODS})

    rng = np.random.default_rng(77)
    D, N = 7, 30
    X = rng.standard_normal((D, D))
    R = 0.5 * np.eye(D) + 0.1 * X.dot(X.T)
    H = np.eye(D) + 0.2 * rng.standard_normal((D, D))
    obsX = np.array([2, 3, 11, 19, 28])
    obsY = rng.standard_normal((5, D))
    m = rng.standard_normal((N, D))
    S = 0.3 * np.eye(D)[None] + 0.05 * rng.standard_normal((N, D, D))
    sde = dict(D=D, N=N, dt=0.01, Rig=R, H=H, obsX=obsX, obsY=obsY)
    Eobs, dm, dS, dR = ref["likelihood"].gauss(m, S, sde)
    rec = dict(D=np.int64(D), N=np.int64(N), Rig=R, H=H, obsX=obsX, obsY=obsY, m=m, S=S,
               Eobs=np.asarray(Eobs, dtype=np.float64), dEobs_dm=np.asarray(dm), dEobs_dS=np.asarray(dS),
               dEobs_dR=np.asarray(dR, dtype=np.float64))
    np.savez_compressed(os.path.join(outdir, "likelihood_dense.npz"), **rec)
    print("wrote likelihood_dense: Eobs", float(np.asarray(Eobs).reshape(-1)[0]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    ap.add_argument("--only", default="", help="'prep', 'callers' or 'extras': only that group of fixtures")
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference tree not present: fixtures can only be regenerated in the build container")
    os.makedirs(args.out, exist_ok=True)
    ref = _import_reference()
    only = dict(prep=golden_prep, callers=golden_callers, extras=golden_extras)
    if args.only:
        only[args.only](ref, args.out)
        return
    golden_systems(ref, args.out)
    golden_units(ref, args.out)
    golden_nonpd_energy(ref, args.out)
    golden_scg(ref, args.out)
    golden_prep(ref, args.out)
    golden_callers(ref, args.out)
    golden_extras(ref, args.out)


if __name__ == "__main__":
    main()
