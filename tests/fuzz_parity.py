"""Randomised parity sweep of the batched HIP evaluation against the NumPy oracle.

Draws configurations the fixed tests do not enumerate -- system, integrator, state dimension,
number of time rows and of problems (both sides of every kernel-selection threshold), observation
patterns with adjacent indices and indices at the ends, dense or diagonal noise / observation
operators, per-problem drift parameters -- evaluates F, the gradient and every intermediate with
engine.free_energy_and_gradient (and the two-call path engine.free_energy + engine.gradient), and
compares a few problems of each batch with oracle.evaluate at 1e-9 max-norm relative.

  python tests/fuzz_parity.py [seconds] [first_seed] [long]

Prints one line per configuration and a summary; exit status 1 when any comparison failed.
tests/test_gpu_fuzz.py runs a fixed number of seeds of it through ``sweep``.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vgpa_amd import engine, _lib  # noqa: E402
from oracle import vgpa_oracle as orc  # noqa: E402

TOL = 1e-9
SYS = {"ou": _lib.SYS_OU, "dw": _lib.SYS_DW, "l63": _lib.SYS_L63, "l63_ut": _lib.SYS_L63_UT, "l96_ut": _lib.SYS_L96_UT}
METHODS = ("EULER", "HEUN", "RK2", "RK4")
L96_DIMS = (4, 5, 6, 8, 12, 16, 17, 24, 33, 40, 40, 40, 41, 48, 57, 60)
P_CHOICES = (1, 2, 3, 5, 17, 64, 161, 170)


def relerr(x, y):
    x, y = np.asarray(x, dtype=np.float64).reshape(-1), np.asarray(y, dtype=np.float64).reshape(-1)
    assert x.size == y.size, (x.size, y.size)
    den = np.max(np.abs(y)) if y.size else 0.0
    num = np.max(np.abs(x - y)) if y.size else 0.0
    return num / den if den > 0 else num


def spd(rng, D, lo, hi, dense):
    """A diagonal matrix with entries in [lo, hi], or that plus a symmetric perturbation that keeps
    it positive definite."""
    M = np.diag(lo + (hi - lo) * rng.random(D))
    if dense:
        Q = 0.1 * lo * rng.standard_normal((D, D)) / np.sqrt(D)
        M = M + Q + Q.T
    return M


def draw(rng):
    system = rng.choice(list(SYS))
    method = str(rng.choice(METHODS))
    D = 1 if system in ("ou", "dw") else 3 if system.startswith("l63") else int(rng.choice(L96_DIMS))
    P = int(rng.choice(P_CHOICES))
    big = D >= 33
    N = int(rng.choice((1, 2, 3, 4, 5, 9, 16, 17, 31, 64) if big else (1, 2, 3, 7, 8, 9, 16, 17, 25, 64, 130, 300)))
    if D >= 16 and P > 64 and N > 17:
        N = 17                                   # keep the oracle side to seconds
    dt = float(rng.choice((0.005, 0.01)))
    n_obs = int(rng.integers(0, min(N, 6) + 1)) if N > 1 else int(rng.integers(0, 2))
    if D > 1:
        n_obs = max(1, n_obs)                    # the reference's n-D likelihood needs one observation at least
    obsX = np.sort(rng.choice(N, size=n_obs, replace=False)) if n_obs else np.zeros(0, dtype=int)
    if n_obs >= 2 and rng.random() < 0.5 and obsX[0] + 1 < N:
        obsX[1] = obsX[0] + 1                    # adjacent indices: a zero-length trapezoid segment
        obsX = np.unique(obsX)
    if n_obs and rng.random() < 0.3:
        obsX[-1] = N - 1
        obsX = np.unique(obsX)
    if n_obs and rng.random() < 0.3:
        obsX[0] = 0
        obsX = np.unique(obsX)
    M = obsX.size
    if D == 1:
        Sig, S0v, tau0 = float(0.5 + rng.random()), float(0.2 + 0.2 * rng.random()), float(0.3 + 0.4 * rng.random())
        sde = dict(Sig=Sig, theta=float(1.0 + rng.random()), Rig=float(0.04 + 0.1 * rng.random()), D=1, N=N, H=1.0,
                   obsX=obsX, obsY=rng.standard_normal((M, 1)), px0=dict(mu0=float(rng.standard_normal()), tau0=tau0),
                   dt=dt, ode_method=method)
        m0, S0 = np.array([float(rng.standard_normal())]), S0v
        a0 = 0.5 * Sig / S0v
        A = a0 + 0.1 * rng.standard_normal((P, N, 1))
        b = rng.standard_normal((P, N, 1))
        thetas = (sde["theta"] * (1.0 + 0.1 * rng.standard_normal((P, 1)))) if P > 1 else None
    else:
        dense_sig = rng.random() < 0.2
        Sig = spd(rng, D, 3.0, 5.0, dense_sig) if system == "l96_ut" else spd(rng, D, 8.0, 12.0, dense_sig)
        dense_obs = rng.random() < 0.25
        H = np.eye(D) + (0.1 * rng.standard_normal((D, D)) if dense_obs else 0.0)
        R = spd(rng, D, 1.0, 2.0, dense_obs)
        S0 = spd(rng, D, 0.2, 0.3, rng.random() < 0.5)
        mu0 = rng.standard_normal(D)
        theta = 8.0 if system == "l96_ut" else np.array([10.0, 28.0, 8.0 / 3.0])
        sde = dict(Sig=Sig, theta=theta, Rig=R, D=D, N=N, H=H, obsX=obsX, obsY=3.0 * rng.standard_normal((M, D)),
                   px0=dict(mu0=mu0, tau0=spd(rng, D, 0.4, 0.6, rng.random() < 0.3)), dt=dt, ode_method=method)
        m0 = mu0 + 0.1 * rng.standard_normal(D)
        a0 = 0.5 * float(np.mean(np.diag(Sig))) / 0.25
        A = a0 * np.eye(D)[None, None] + 0.1 * rng.standard_normal((P, N, D, D))
        b = 5.0 * rng.standard_normal((P, N, D))
        if P > 1:
            nth = 1 if system == "l96_ut" else 3
            thetas = np.atleast_1d(theta).reshape(1, nth) * (1.0 + 0.05 * rng.standard_normal((P, nth)))
        else:
            thetas = None
    return dict(system=str(system), method=method, D=D, P=P, N=N, sde=sde, m0=m0, S0=S0, A=A, b=b, thetas=thetas)


def draw_long(rng):
    """Lorenz-96 at the matrix-core dimensions with long time axes (the scans' prefetch rings and chunk
    boundaries), every integrator, both sides of the fused-kernel threshold."""
    cfg = None
    while cfg is None or cfg["system"] != "l96_ut":
        cfg = draw(rng)
    D = int(rng.choice((16, 40, 40, 40)))
    P = int(rng.choice((1, 2, 161, 163)))
    N = int(rng.choice((63, 64, 65, 127, 128, 129, 255, 257, 300, 511, 513, 1000)))
    if P > 100:
        N = min(N, 257)
    sde = dict(cfg["sde"], D=D, N=N)
    obsX = np.unique(np.concatenate((rng.choice(N, size=5, replace=False), [N - 1] if rng.random() < 0.5 else [N // 2])))
    Sig = spd(rng, D, 3.0, 5.0, False)
    mu0 = rng.standard_normal(D)
    sde.update(Sig=Sig, Rig=np.eye(D), H=np.eye(D), obsX=obsX, obsY=3.0 * rng.standard_normal((obsX.size, D)),
               px0=dict(mu0=mu0, tau0=0.5 * np.eye(D)), theta=8.0)
    a0 = 0.5 * float(np.mean(np.diag(Sig))) / 0.25
    A = a0 * np.eye(D)[None, None] + 0.1 * rng.standard_normal((P, N, D, D))
    b = 5.0 * rng.standard_normal((P, N, D))
    thetas = 8.0 * (1.0 + 0.05 * rng.standard_normal((P, 1))) if P > 1 else None
    return dict(system="l96_ut", method=cfg["method"], D=D, P=P, N=N, sde=sde, m0=mu0 + 0.1 * rng.standard_normal(D),
                S0=spd(rng, D, 0.2, 0.3, rng.random() < 0.5), A=A, b=b, thetas=thetas)


PACKED_SEEN = [0]       # configurations that also ran on packed symmetric matrices


def check(cfg, rng):
    system, D, P, N, sde = cfg["system"], cfg["D"], cfg["P"], cfg["N"], cfg["sde"]
    prob = engine.make_problem(SYS[system], sde, cfg["m0"], cfg["S0"], P=P, theta=cfg["thetas"])
    if D == 1:
        Ad, bd = engine.to_dev(cfg["A"].reshape(P, N)), engine.to_dev(cfg["b"].reshape(P, N))
    else:
        Ad, bd = engine.to_dev(cfg["A"]), engine.to_dev(cfg["b"])
    st, gLA, gLb = engine.free_energy_and_gradient(prob, Ad, bd, want_hyper=True)
    st2 = engine.free_energy(prob, Ad, bd, want_hyper=False, want_edf=True)
    gA2, gb2 = engine.gradient(prob, Ad, bd, st2)
    torch.cuda.synchronize()
    if int(st.status.max().item()) != 0:
        return None                              # a covariance lost positive definiteness: not a parity case
    worst, where = 0.0, ""
    picks = sorted(set([0, P - 1, int(rng.integers(0, P))]))
    for p in picks:
        sp = dict(sde)
        if cfg["thetas"] is not None:
            th = cfg["thetas"][p]
            sp["theta"] = float(th[0]) if th.size == 1 else np.array(th)
        Ab = np.concatenate((cfg["A"][p].ravel(), cfg["b"][p].ravel()))
        ref = orc.evaluate(system, Ab, cfg["m0"], cfg["S0"], sp)
        errs = {"F": abs(float(st.F[p].item()) - ref["F"]) / max(abs(ref["F"]), 1e-300),
                "F2": abs(float(st2.F[p].item()) - ref["F"]) / max(abs(ref["F"]), 1e-300),
                "grad": relerr(np.concatenate((gLA[p].reshape(-1).cpu().numpy(), gLb[p].reshape(-1).cpu().numpy())), ref["grad"]),
                "grad2": relerr(np.concatenate((gA2[p].reshape(-1).cpu().numpy(), gb2[p].reshape(-1).cpu().numpy())), ref["grad"]),
                "m": relerr(st.m[p].cpu().numpy(), ref["m"]), "S": relerr(st.S[p].cpu().numpy(), ref["S"]),
                "lam": relerr(st.lam[p].cpu().numpy(), ref["lam"]), "Psi": relerr(st.Psi[p].cpu().numpy(), ref["Psi"]),
                "Ef": relerr(st.Ef[p].cpu().numpy(), ref["Ef"]), "Edf2": relerr(st2.Edf[p].cpu().numpy(), ref["Edf"])}
        if st.dth is not None and ref.get("dEsde_dth") is not None:
            errs["dth"] = relerr(st.dth[p].cpu().numpy(), ref["dEsde_dth"])
            errs["dSig"] = relerr(st.dSig[p].cpu().numpy(), ref["dEsde_dSig"])
        for k, v in errs.items():
            if not np.isfinite(v):
                v = np.inf
            if v > worst:
                worst, where = v, f"{k}[p={p}]"
    if engine.packed_supported(prob):
        # the packed symmetric layout of the Lorenz-96 Euler kernels: the same bits as the dense arrays
        sp, pA, pb = engine.free_energy_and_gradient(prob, Ad, bd, want_hyper=True, packed=True)
        same = (torch.equal(pA, gLA) and torch.equal(pb, gLb) and torch.equal(sp.F, st.F) and torch.equal(sp.lam, st.lam)
                and torch.equal(engine.sym_unpack(sp.S), st.S) and torch.equal(engine.sym_unpack(sp.dEdS), st.dEdS)
                and torch.equal(engine.sym_unpack(sp.Psi), st.Psi) and torch.equal(sp.dSig, st.dSig))
        PACKED_SEEN[0] += 1
        if not same:
            worst, where = np.inf, "packed != dense"
    if P == 1:
        # the reference-signature entry points on host arrays (column or flat Ab, chosen at random)
        from vgpa_amd.core.varFreeEnergy import varFreeEnergy
        from vgpa_amd.core.gradL_Ab import gradL_Ab
        from vgpa_amd.dynamics import (sys_ornstein_uhlenbeck, sys_double_well, sys_lorenz_63, sys_lorenz_63_ut,
                                       sys_lorenz_96_ut)
        mod = dict(ou=sys_ornstein_uhlenbeck, dw=sys_double_well, l63=sys_lorenz_63, l63_ut=sys_lorenz_63_ut,
                   l96_ut=sys_lorenz_96_ut)[system]
        Ab = np.concatenate((cfg["A"][0].ravel(), cfg["b"][0].ravel()))
        if rng.random() < 0.5:
            Ab = Ab.reshape(-1, 1)
        mParam = dict(m0=cfg["m0"], S0=cfg["S0"], fun=mod.energy_mode, sde_struct=dict(sde, checkGradf=False))
        F = varFreeEnergy(Ab, mParam)
        grad = gradL_Ab(Ab, mParam, bool(rng.random() < 0.3))
        ref = orc.evaluate(system, Ab, cfg["m0"], cfg["S0"], sde)
        errs = {"facade.F": abs(F - ref["F"]) / max(abs(ref["F"]), 1e-300) if isinstance(F, float) else np.inf,
                "facade.grad": relerr(grad, ref["grad"]) if grad.shape == (Ab.size, 1) else np.inf,
                "facade.mt": relerr(mParam["mt"], ref["m"]), "facade.St": relerr(mParam["St"], ref["S"]),
                "facade.lamt": relerr(mParam["lamt"], ref["lam"]), "facade.Psit": relerr(mParam["Psit"], ref["Psi"])}
        for k, v in errs.items():
            v = v if np.isfinite(v) else np.inf
            if v > worst:
                worst, where = v, k
    return worst, where


def check_energy_steps(seed):
    """energy_mode alone on covariances a third of which have one negative eigenvalue (positive diagonal):
    those steps take the diagonal sigma points of auxiliary/numerical.py:201-202.  Unscented systems, every
    kernel family of the Lorenz-96 energy (generic D, D = 16, D = 40).  Returns (tag, worst, where)."""
    from vgpa_amd.dynamics import sys_lorenz_63_ut, sys_lorenz_96_ut
    rng = np.random.default_rng(seed)
    system = str(rng.choice(("l63_ut", "l96_ut", "l96_ut", "l96_ut")))
    D = 3 if system == "l63_ut" else int(rng.choice((4, 6, 8, 16, 16, 17, 24, 40, 40, 41, 60)))
    N = int(rng.choice((1, 2, 3, 5, 8, 13)))
    A = 8.0 * np.eye(D)[None] + 0.3 * rng.standard_normal((N, D, D))
    b = 3.0 * rng.standard_normal((N, D))
    m = 2.0 * rng.standard_normal((N, D))
    S = np.zeros((N, D, D))
    n_bad = 0
    for t in range(N):
        X = rng.standard_normal((D, D))
        S[t] = 0.3 * np.eye(D) + 0.1 * X.dot(X.T) / D
        if rng.random() < 0.35:
            S[t] -= (0.5 if D == 3 else 0.6) * np.ones((D, D)) / (1 if D == 3 else D)
            S[t][np.diag_indices(D)] = np.abs(S[t][np.diag_indices(D)]) + 0.05
            n_bad += int(np.linalg.eigvalsh(S[t]).min() < 0)
    obsX = np.unique(rng.choice(N, size=min(N, 2), replace=False))
    theta = 8.0 if system == "l96_ut" else np.array([10.0, 28.0, 8.0 / 3.0])
    sde = dict(N=N, D=D, dt=0.01, Sig=np.diag(2.0 + rng.random(D)), theta=theta, obsX=obsX)
    mod = sys_lorenz_96_ut if system == "l96_ut" else sys_lorenz_63_ut
    got = mod.energy_mode(A, b, m, S, sde)
    ref = orc.energy_mode(system, A, b, m, S, sde)
    names = ["Esde", "Ef", "Edf", "dEsde_dm", "dEsde_dS", "dEsde_dth", "dEsde_dSig"]
    worst, where = 0.0, ""
    for nme, g_, r_ in zip(names, got, ref):
        e = relerr(g_, r_)
        e = e if np.isfinite(e) else np.inf
        if e > worst:
            worst, where = e, nme
    return f"energy seed {seed}: {system:7s} D={D:2d} N={N:2d} non-PD steps {n_bad}", worst, where


def sweep_energy(first_seed, count, verbose=True):
    failures, worst_all = [], 0.0
    for seed in range(first_seed, first_seed + count):
        tag, worst, where = check_energy_steps(seed)
        worst_all = max(worst_all, worst)
        line = f"{tag}: worst {worst:.2e} at {where}"
        if worst > TOL:
            failures.append(line)
            line += "   <-- FAIL"
        if verbose:
            print(line, flush=True)
    return failures, worst_all


def sweep(first_seed, count=None, seconds=None, verbose=True, long_axes=False):
    """Configurations first_seed, first_seed + 1, ...: `count` of them, or as many as fit in `seconds`.
    ``long_axes``: draw_long instead of draw.
    Returns (configurations, skipped, failures, worst error); failures is a list of report lines."""
    t0, n, skipped, worst_all, failures = time.time(), 0, 0, 0.0, []
    seed = first_seed
    while (count is None or n < count) and (seconds is None or time.time() - t0 < seconds):
        rng = np.random.default_rng(seed)
        cfg = draw_long(rng) if long_axes else draw(rng)
        tag = (f"seed {seed}: {cfg['system']:7s} {cfg['method']:5s} D={cfg['D']:2d} N={cfg['N']:3d} P={cfg['P']:3d} "
               f"obs={[int(i) for i in cfg['sde']['obsX']]}")
        try:
            res = check(cfg, rng)
        except Exception as exc:   # noqa: BLE001 - report and count, the sweep goes on
            res = False
            failures.append(f"{tag}: EXCEPTION {type(exc).__name__}: {exc}")
            if verbose:
                print(failures[-1], flush=True)
        if res is None:
            skipped += 1
            if verbose:
                print(f"{tag}: skipped (status flag set)", flush=True)
        elif res:
            worst, where = res
            worst_all = max(worst_all, worst)
            line = f"{tag}: worst {worst:.2e} at {where}"
            if worst > TOL:
                failures.append(line)
                line += "   <-- FAIL"
            if verbose:
                print(line, flush=True)
        n += 1
        seed += 1
    return n, skipped, failures, worst_all


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    if len(sys.argv) > 3 and sys.argv[3] == "long":
        n, skipped, failures, worst_all = sweep(seed, seconds=budget, long_axes=True)
        print(f"{n} long-axis configurations, {skipped} skipped, {len(failures)} failed, worst error {worst_all:.2e} "
              f"(tolerance {TOL:g}); {PACKED_SEEN[0]} of them also on packed symmetric matrices (bitwise equal or failed)",
              flush=True)
        sys.exit(1 if failures else 0)
    e_fail, e_worst = sweep_energy(seed, 200)
    print(f"200 energy_mode configurations with non-positive-definite steps, {len(e_fail)} failed, worst error {e_worst:.2e}",
          flush=True)
    n, skipped, failures, worst_all = sweep(seed, seconds=budget)
    failures += e_fail
    print(f"{n} configurations, {skipped} skipped, {len(failures)} failed, worst error {worst_all:.2e} (tolerance {TOL:g}); "
          f"{PACKED_SEEN[0]} of them also on packed symmetric matrices (bitwise equal or failed)", flush=True)
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
