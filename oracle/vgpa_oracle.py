"""CPU oracle for the VGPA free-energy / gradient path.

TEST INFRASTRUCTURE ONLY -- never imported by the product package
(``vgpa_amd``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may use this module, and only as the
checker / timed baseline.

This is an independent NumPy restatement of the reference algorithm
(ruiixu23/VGPA, paths relative to the reference root are cited per function).
It deliberately follows the reference's *literal* formulas (per-sigma-point
solves, explicit inverse of S, segment-wise trapezoid) rather than the closed
forms used by the HIP kernels, so that agreement between the two is a real
cross-check.  It is vectorised over the time axis where the reference loops.

Pinned: ``tests/test_oracle_golden.py`` checks every function here against the
fixtures in ``tests/golden/`` that ``oracle/make_golden.py`` produced by running
the reference itself (tolerance 1e-12 max-norm relative).
"""
import numpy as np

LOG2PI = 1.8378770664093453  # core/likelihood.py@10

METHODS = ("EULER", "HEUN", "RK2", "RK4")


# --------------------------------------------------------------------------
# helpers: auxiliary/numerical.py
# --------------------------------------------------------------------------
def log_det(A):
    """auxiliary/numerical.py:51-86 -- 2*sum(log(diag(chol(A))))."""
    A = np.asarray(A, dtype=np.float64)
    if A.ndim == 1:
        A = np.diag(A)
    return 2.0 * np.sum(np.log(np.linalg.cholesky(A).diagonal()))


def chol_inv(A):
    """auxiliary/numerical.py:10-48.  Returns (Ri.dot(Ri.T), Ri) with
    Ri = inv(chol(A)); note the first is L^-1 L^-T, which equals A^-1 only for
    diagonal A -- reproduced as is."""
    if np.isscalar(A):
        return 1.0 / A, 1.0 / np.sqrt(A)
    A = np.asarray(A, dtype=np.float64)
    if A.ndim == 1:
        A = np.diag(A)
    Ri = np.linalg.inv(np.linalg.cholesky(A))
    return Ri.dot(Ri.T), Ri


def trapz_weights(N, idx=None):
    """Per-sample weights w_t such that mytrapz(f, dt, idx) == dt * sum_t w_t f_t
    (auxiliary/numerical.py:231-284).  Segments are [f, l) between consecutive
    observation indices; the sample at l and everything from idx[-1] on is
    dropped; a one-sample segment contributes nothing."""
    w = np.zeros(N, dtype=np.float64)
    if idx is None:
        segs = [(0, N)]
    else:
        segs, f = [], 0
        for l in idx:
            segs.append((f, int(l)))
            f = int(l)
    for f, l in segs:
        n = l - f
        if n >= 2:
            w[f:l] += 1.0
            w[f] -= 0.5
            w[l - 1] -= 0.5
    return w


def mytrapz(fxt, dt, idx=None):
    """auxiliary/numerical.py:231-284, literal segment loop (kept separate from
    ``trapz_weights`` so the two can be checked against each other)."""
    fxt = np.asarray(fxt, dtype=np.float64)

    def tz(y):
        if y.shape[0] < 2:
            return np.zeros(y.shape[1:]) if y.ndim > 1 else 0.0
        return dt * (0.5 * (y[0] + y[-1]) + y[1:-1].sum(axis=0))

    if idx is None:
        return tz(fxt)
    v, f = 0.0, 0
    for l in idx:
        v = v + tz(fxt[f:l])
        f = l
    return v


def mom_gauss(m, S, order=0, dmS=None):
    """auxiliary/momGauss.py:5-107 -- uncentred Gaussian moments 0..8."""
    one, zero = np.ones_like(m, dtype=np.float64), np.zeros_like(m, dtype=np.float64)
    tab = {
        (0, None): lambda: one, (0, "Dm"): lambda: one, (0, "DS"): lambda: one,
        (1, None): lambda: m, (1, "Dm"): lambda: one, (1, "DS"): lambda: zero,
        (2, None): lambda: m**2 + S, (2, "Dm"): lambda: 2 * m, (2, "DS"): lambda: one,
        (3, None): lambda: m**3 + 3 * m * S, (3, "Dm"): lambda: 3 * (m**2 + S), (3, "DS"): lambda: 3 * m,
        (4, None): lambda: m**4 + 6 * (m**2) * S + 3 * (S**2),
        (4, "Dm"): lambda: 4 * (m**3 + 3 * m * S), (4, "DS"): lambda: 6 * (m**2 + S),
        (5, None): lambda: m**5 + 10 * (m**3) * S + 15 * m * (S**2),
        (5, "Dm"): lambda: 5 * (m**4 + 6 * (m**2) * S + 3 * (S**2)),
        (5, "DS"): lambda: 10 * (m**3) + 30 * (m * S),
        (6, None): lambda: m**6 + 15 * (m**4) * S + 45 * (m**2) * (S**2) + 15 * (S**3),
        (6, "Dm"): lambda: 6 * (m**5 + 10 * (m**3) * S + 15 * m * (S**2)),
        (6, "DS"): lambda: 15 * (m**4) + 90 * (m**2) * S + 45 * (S**2),
        (7, None): lambda: m**7 + 21 * (m**5) * S + 105 * (m**3) * (S**2) + 105 * m * (S**3),
        (7, "Dm"): lambda: 7 * (m**6 + 15 * (m**4) * S + 45 * (m**2) * (S**2) + 15 * (S**3)),
        (7, "DS"): lambda: 21 * (m**5) + 210 * (m**3) * S + 315 * m * (S**2),
        (8, None): lambda: m**8 + 28 * (m**6) * S + 210 * (m**4) * (S**2) + 420 * (m**2) * (S**3) + 105 * (S**4),
        (8, "Dm"): lambda: 8 * (m**7 + 21 * (m**5) * S + 105 * (m**3) * (S**2) + 105 * m * (S**3)),
        (8, "DS"): lambda: 28 * (m**6) + 420 * (m**4) * S + 1260 * (m**2) * (S**2) + 420 * (S**3),
    }
    if not (0 <= order <= 8):
        raise ValueError("Unknown Higher Order Gaussian Moment!")
    return tab[(order, dmS)]()


# --------------------------------------------------------------------------
# forward / backward moment scans: core/fwd_ode.py, core/bwd_ode.py
# --------------------------------------------------------------------------
def fwd_scan(A, b, m0, S0, Sigma, dt, method="EULER"):
    """core/fwd_ode.py:48-559.  1-D path iff A.ndim==1 or A.shape[1]==1."""
    method = method.upper()
    if method not in METHODS:
        raise ValueError("Unknown method of integration")
    one_d = (A.ndim == 1) or (A.shape[1] == 1)
    N = b.shape[0]
    if one_d:
        m = np.zeros((N, 1))
        S = np.zeros((N, 1))
        m[0], S[0] = m0, S0
        fm = lambda mt, At, bt: -At * mt + bt            # noqa: E731  fwd_ode.py:173
        fs = lambda St, At: -2 * At * St + Sigma          # noqa: E731  fwd_ode.py:176
    else:
        D = b.shape[1]
        m = np.zeros((N, D))
        S = np.zeros((N, D, D))
        m[0], S[0] = m0, S0
        fm = lambda mt, At, bt: -mt.dot(At.T) + bt        # noqa: E731  fwd_ode.py:44
        fs = lambda St, At: -At.dot(St) - St.dot(At.T) + Sigma  # noqa: E731  fwd_ode.py:45
    h = 0.5 * dt
    for t in range(N - 1):
        At, bt, St, mt = A[t], b[t], S[t], m[t]
        if method == "EULER":                             # fwd_ode.py:121-131
            m[t + 1] = mt + fm(mt, At, bt) * dt
            S[t + 1] = St + fs(St, At) * dt
        elif method == "HEUN":                            # fwd_ode.py:218-243
            Ak, bk = A[t + 1], b[t + 1]
            ftp = fm(mt, At, bt)
            ftc = fm(mt + ftp * dt, Ak, bk)
            m[t + 1] = mt + h * (ftp + ftc)
            ftp = fs(St, At)
            ftc = fs(St + ftp * dt, Ak)
            S[t + 1] = St + h * (ftp + ftc)
        elif method == "RK2":                             # fwd_ode.py:349-363
            Ak = 0.5 * (At + A[t + 1])
            bk = 0.5 * (bt + b[t + 1])
            m[t + 1] = mt + fm(mt + h * fm(mt, At, bt), Ak, bk) * dt
            S[t + 1] = St + fs(St + h * fs(St, At), Ak) * dt
        else:                                             # fwd_ode.py:460-489
            Ap, bp = A[t + 1], b[t + 1]
            Ak, bk = 0.5 * (At + Ap), 0.5 * (bt + bp)
            K1 = fm(mt, At, bt) * dt
            K2 = fm(mt + 0.5 * K1, Ak, bk) * dt
            K3 = fm(mt + 0.5 * K2, Ak, bk) * dt
            K4 = fm(mt + K3, Ap, bp) * dt
            m[t + 1] = mt + (K1 + 2.0 * (K2 + K3) + K4) / 6.0
            L1 = fs(St, At) * dt
            L2 = fs(St + 0.5 * L1, Ak) * dt
            L3 = fs(St + 0.5 * L2, Ak) * dt
            L4 = fs(St + L3, Ap) * dt
            S[t + 1] = St + (L1 + 2.0 * (L2 + L3) + L4) / 6.0
    return m, S


def bwd_scan(A, dEsde_dm, dEsde_dS, dEobs_dm, dEobs_dS, dt, method="EULER", lam_last=None, Psi_last=None):
    """core/bwd_ode.py:50-563.  The reference starts from lam = Psi = 0 at the last
    row; ``lam_last`` / ``Psi_last`` replace that start so that tests can check a scan
    cut into time windows."""
    method = method.upper()
    if method not in METHODS:
        raise ValueError("Unknown method of integration")
    one_d = (A.ndim == 1) or (A.shape[1] == 1)
    N = dEsde_dm.shape[0]
    if one_d:
        lam = np.zeros((N, 1))
        Psi = np.zeros((N, 1))
        fl = lambda dEdm, At, lt: -dEdm + At * lt            # noqa: E731 bwd_ode.py:46
        fp = lambda dEdS, At, Pt: -dEdS + 2.0 * Pt * At      # noqa: E731 bwd_ode.py:47
    else:
        D = dEsde_dm.shape[1]
        lam = np.zeros((N, D))
        Psi = np.zeros((N, D, D))
        fl = lambda dEdm, At, lt: -dEdm + lt.dot(At.T)       # noqa: E731 bwd_ode.py:42
        fp = lambda dEdS, At, Pt: -dEdS + Pt.dot(At) + At.T.dot(Pt)  # noqa: E731 bwd_ode.py:43
    if lam_last is not None:
        lam[N - 1] = lam_last
        Psi[N - 1] = Psi_last
    h = 0.5 * dt
    for t in range(N - 1, 0, -1):
        At, lt, Pt = A[t], lam[t], Psi[t]
        if method == "EULER":                                # bwd_ode.py:117-127
            lam[t - 1] = lt - fl(dEsde_dm[t], At, lt) * dt + dEobs_dm[t - 1]
            Psi[t - 1] = Pt - fp(dEsde_dS[t], At, Pt) * dt + dEobs_dS[t - 1]
        elif method == "HEUN":                               # bwd_ode.py:211-237
            Ak = A[t - 1]
            ftp = fl(dEsde_dm[t], At, lt)
            ftc = fl(dEsde_dm[t - 1], Ak, lt - ftp * dt)
            lam[t - 1] = lt - h * (ftp + ftc) + dEobs_dm[t - 1]
            ftp = fp(dEsde_dS[t], At, Pt)
            ftc = fp(dEsde_dS[t - 1], Ak, Pt - ftp * dt)
            Psi[t - 1] = Pt - h * (ftp + ftc) + dEobs_dS[t - 1]
        elif method == "RK2":                                # bwd_ode.py:338-362
            Ak = 0.5 * (A[t - 1] + At)
            dEmk = 0.5 * (dEsde_dm[t - 1] + dEsde_dm[t])
            dESk = 0.5 * (dEsde_dS[t - 1] + dEsde_dS[t])
            lamk = lt - h * fl(dEsde_dm[t], At, lt)
            lam[t - 1] = lt - fl(dEmk, Ak, lamk) * dt + dEobs_dm[t - 1]
            Psik = Pt - h * fp(dEsde_dS[t], At, Pt)
            Psi[t - 1] = Pt - fp(dESk, Ak, Psik) * dt + dEobs_dS[t - 1]
        else:                                                # bwd_ode.py:458-493
            Ap = A[t - 1]
            Ak = 0.5 * (Ap + At)
            dEmk = 0.5 * (dEsde_dm[t - 1] + dEsde_dm[t])
            dESk = 0.5 * (dEsde_dS[t - 1] + dEsde_dS[t])
            K1 = fl(dEsde_dm[t], At, lt) * dt
            K2 = fl(dEmk, Ak, lt - 0.5 * K1) * dt
            K3 = fl(dEmk, Ak, lt - 0.5 * K2) * dt
            K4 = fl(dEsde_dm[t - 1], Ap, lt - K3) * dt
            lam[t - 1] = lt - (K1 + 2.0 * (K2 + K3) + K4) / 6.0 + dEobs_dm[t - 1]
            L1 = fp(dEsde_dS[t], At, Pt) * dt
            L2 = fp(dESk, Ak, Pt - 0.5 * L1) * dt
            L3 = fp(dESk, Ak, Pt - 0.5 * L2) * dt
            L4 = fp(dEsde_dS[t - 1], Ap, Pt - L3) * dt
            Psi[t - 1] = Pt - (L1 + 2.0 * (L2 + L3) + L4) / 6.0 + dEobs_dS[t - 1]
    return lam, Psi


# --------------------------------------------------------------------------
# observation energy and initial-state KL: core/likelihood.py, core/kl0.py
# --------------------------------------------------------------------------
def likelihood_gauss(m, S, sde):
    """core/likelihood.py@13-179.  Returns (Eobs, dEobs_dm, dEobs_dS)."""
    Y, idx, N, H, R = sde["obsY"], sde["obsX"], sde["N"], sde["H"], sde["Rig"]
    if sde["D"] == 1:                                        # likelihood.py@52-105
        M = Y.shape[0]
        W = Y - H * np.take(m, idx, axis=0)
        Ri = 1.0 / R
        dm = np.zeros((N, 1))
        dS = np.zeros((N, 1))
        Ex2 = m**2 + S
        Eobs = 0.5 * Ri * np.sum(Y**2 - 2 * Y * m[idx] + Ex2[idx]) + 0.5 * M * (LOG2PI + np.log(R))
        dm[idx] = -Ri * W
        dS[idx] = 0.5 * Ri
        return Eobs, dm, dS
    D = sde["D"]                                             # likelihood.py@108-179
    M, d = Y.shape
    W = Y - np.take(m, idx, axis=0).dot(H)
    Ri, Qi = chol_inv(R)
    dRi = Ri.diagonal()
    Z = W.dot(Qi.T)
    Eobs = 0.0
    dm = np.zeros((N, D))
    dS = np.zeros((N, D, D))
    for n in range(M):
        tn = idx[n]
        Eobs += Z[n].dot(Z[n]) + dRi.dot(S[tn].diagonal())
        dm[tn] = -H.T.dot(Ri).dot(W[n])
        dS[tn] = 0.5 * H.T.dot(Ri).dot(H)
    Eobs = 0.5 * (Eobs + M * (d * LOG2PI + log_det(R)))
    return Eobs, dm, dS


def kl0_gauss(m0, S0, sde):
    """core/kl0.py:9-99 (energy only).  1-D uses -log(S0) (kl0.py:60); n-D with
    an (D,) mean broadcasts the scalar |z0|^2 into the matrix (kl0.py:90)."""
    mu0, tau0 = sde["px0"]["mu0"], sde["px0"]["tau0"]
    z0 = m0 - mu0
    if sde["D"] == 1:
        return -np.log(S0) - 0.5 * (1.0 - np.log(tau0)) + 0.5 / tau0 * (z0**2 + S0)
    tau0i = np.linalg.inv(tau0)
    S0i = np.linalg.inv(S0)
    return 0.5 * (log_det(tau0.dot(S0i)) + np.sum(np.diag(tau0i.dot(z0.T.dot(z0) + S0 - tau0))))


# --------------------------------------------------------------------------
# drift functions
# --------------------------------------------------------------------------
def lorenz63(x, u):
    """dynamics/sys_lorenz_63_ut.py:17-46 on (..., 3) arrays."""
    s, r, b = u
    dx = np.empty_like(x)
    dx[..., 0] = -s * (x[..., 0] - x[..., 1])
    dx[..., 1] = (r - x[..., 2]) * x[..., 0] - x[..., 1]
    dx[..., 2] = x[..., 0] * x[..., 1] - b * x[..., 2]
    return dx


def lorenz96(x, f):
    """dynamics/sys_lorenz_96_ut.py:17-40 on (..., D) arrays:
    dx_i = (x_{i+1} - x_{i-2}) x_{i-1} - x_i + f."""
    return (np.roll(x, -1, -1) - np.roll(x, 2, -1)) * np.roll(x, 1, -1) - x + f


# --------------------------------------------------------------------------
# unscented transform + literal gradient kernel
# --------------------------------------------------------------------------
def ut_sigma_points(m, S):
    """auxiliary/numerical.py:186-213, batched over leading axes of m (...,D),
    S (...,D,D).  Returns chi (...,2D+1,D), weights (2D+1,), fallback flags."""
    D = m.shape[-1]
    k = 1.05 * D
    lead = m.shape[:-1]
    mf = m.reshape(-1, D)
    Sf = S.reshape(-1, D, D)
    U = np.empty_like(Sf)
    flag = np.zeros(mf.shape[0], dtype=np.int32)
    for i in range(mf.shape[0]):
        try:
            U[i] = np.linalg.cholesky((D + k) * Sf[i]).T          # numerical.py:200
        except np.linalg.LinAlgError:
            U[i] = np.linalg.cholesky(Sf[i] * np.eye(D)).T       # numerical.py:202
            flag[i] = 1
    chi = np.concatenate((mf[:, None, :], mf[:, None, :] + U, mf[:, None, :] - U), axis=1)
    w = np.full(2 * D + 1, 1.0 / (2.0 * (D + k)))
    w[0] = k / (D + k)
    return chi.reshape(lead + (2 * D + 1, D)), w, flag.reshape(lead)


def ut_approx(f, xbar, Pxx, *args):
    """auxiliary/numerical.py:153-228 for one mean / covariance: returns
    (ybar (1,C), Pyy (C,C)).  Q is built exactly as there (:222-223); it is not the
    usual weighted covariance because the last factor carries the weights a second
    time."""
    chi, w, _ = ut_sigma_points(np.asarray(xbar).reshape(-1), np.asarray(Pxx))
    Y = f(chi, *args)                                       # :216
    M = w.size
    wM = np.eye(M) - np.tile(w[None, :], (M, 1))            # :222
    Q = wM.dot(np.diag(w)).dot(wM.T)                        # :223
    return w[None, :].dot(Y), Y.T.dot(Q).dot(Y)             # :219, :226


def grad_esde_ms(x, Fx, mt, St, At, bt, diagSigI):
    """core/gradEsde_mS.py:8-68, batched: x (...,K,D) samples, Fx (...,K,D) the
    drift at the samples.  Returns (dmt (...,K,D), dSt (...,K,D,D)) per sample."""
    r2 = (Fx + np.einsum("...kj,...ij->...ki", x, At) - bt[..., None, :]) ** 2   # :44
    v = np.einsum("...kd,d->...k", r2, diagSigI)                                  # :45
    dmt = np.linalg.solve(St[..., None, :, :], (v[..., None] * x)[..., None])[..., 0]   # :48
    Sti = np.linalg.inv(St)                                                        # :51
    z = x - mt[..., None, :]                                                       # :56
    u = np.linalg.solve(St[..., None, :, :], z[..., None])[..., 0]                # solve(St, Zk) = u z'
    zs = np.einsum("...kj,...ji->...ki", z, Sti)                                   # z' Sti
    dSt = v[..., None, None] * u[..., :, None] * zs[..., None, :]                  # :60
    return 0.5 * dmt, 0.5 * dSt


def _ut_energy(drift, A, b, m, S, diagSigI):
    """Shared body of the two unscented energy_mode functions
    (dynamics/sys_lorenz_96_ut.py:217-247, dynamics/sys_lorenz_63_ut.py:218-255),
    vectorised over t.  Returns Esde_t (N,), mbar (N,D), dEsde_dm, dEsde_dS."""
    chi, w, flag = ut_sigma_points(m, S)
    Fx = drift(chi)
    r2 = (Fx + np.einsum("nkj,nij->nki", chi, A) - b[:, None, :]) ** 2
    mbar = np.einsum("k,nkd->nd", w, r2)                   # ut_approx ybar, numerical.py:219
    Esde_t = 0.5 * mbar.dot(diagSigI)
    dmt, dSt = grad_esde_ms(chi, Fx, m, S, A, b, diagSigI)
    dm_bar = np.einsum("k,nkd->nd", w, dmt)
    dS_bar = np.einsum("k,nkij->nij", w, dSt)
    Sim = np.linalg.solve(S, m[..., None])[..., 0]
    dEdm = dm_bar - Esde_t[:, None] * Sim
    dEdS = 0.5 * (dS_bar - Esde_t[:, None, None] * np.linalg.inv(S))
    return Esde_t, mbar, dEdm, dEdS, flag


def _l63_moments(m, S):
    mx, my, mz = m[:, 0], m[:, 1], m[:, 2]
    Sxx, Sxy, Sxz = S[:, 0, 0], S[:, 0, 1], S[:, 0, 2]     # upper triangle only
    Syy, Syz, Szz = S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]
    return mx, my, mz, Sxx, Sxy, Sxz, Syy, Syz, Szz


def l63_drift_theta(A, b, m, S, theta):
    """dynamics/sys_lorenz_63.py:491-549 (identical to sys_lorenz_63_ut.py:269-327)."""
    vS, vR, vB = theta
    mx, my, mz, Sxx, Sxy, Sxz, Syy, Syz, Szz = _l63_moments(m, S)
    A11, A12, A13 = A[:, 0, 0], A[:, 0, 1], A[:, 0, 2]
    A21, A22, A23 = A[:, 1, 0], A[:, 1, 1], A[:, 1, 2]
    A31, A32, A33 = A[:, 2, 0], A[:, 2, 1], A[:, 2, 2]
    b1, b2, b3 = b[:, 0], b[:, 1], b[:, 2]
    Exx, Exy, Eyy = Sxx + mx**2, Sxy + mx * my, Syy + my**2
    Exz, Ezz, Eyz = Sxz + mx * mz, Szz + mz**2, Syz + my * mz
    Exxz = Sxx * mz + 2 * Sxz * mx + (mx**2) * mz
    Exyz = Sxy * mz + Sxz * my + Syz * mx + mx * my * mz
    V1 = Eyy * (vS + A12) + Exx * (vS - A11) + Exy * (A11 - 2 * vS - A12) + A13 * (Eyz - Exz) + b1 * (mx - my)
    V2 = vR * Exx - Exy - Exxz + A21 * Exx + A22 * Exy + A23 * Exz - b2 * mx
    V3 = -Exyz + vB * Ezz - A31 * Exz - A32 * Eyz - A33 * Ezz + b3 * mz
    return np.stack((V1, V2, V3), axis=1)


def _l63_ef_edf(m, S, theta):
    """<f> and <df/dx> for Lorenz-63: sys_lorenz_63.py:222-227."""
    vS, vR, vB = theta
    N = m.shape[0]
    Ef = np.stack((vS * (m[:, 1] - m[:, 0]),
                   vR * m[:, 0] - m[:, 1] - S[:, 2, 0] - m[:, 0] * m[:, 2],
                   S[:, 1, 0] + m[:, 0] * m[:, 1] - vB * m[:, 2]), axis=1)
    Edf = np.zeros((N, 3, 3))
    Edf[:, 0, 0], Edf[:, 0, 1] = -vS, vS
    Edf[:, 1, 0], Edf[:, 1, 1], Edf[:, 1, 2] = vR - m[:, 2], -1.0, -m[:, 0]
    Edf[:, 2, 0], Edf[:, 2, 1], Edf[:, 2, 2] = m[:, 1], m[:, 0], -vB
    return Ef, Edf


def l63_energy_dm_dS(A, b, m, S, diagSigI, theta):
    """dynamics/sys_lorenz_63.py:247-488, vectorised over t."""
    vS, vR, vB = theta
    mx, my, mz, Sxx, Sxy, Sxz, Syy, Syz, Szz = _l63_moments(m, S)
    A11, A12, A13 = A[:, 0, 0], A[:, 0, 1], A[:, 0, 2]
    A21, A22, A23 = A[:, 1, 0], A[:, 1, 1], A[:, 1, 2]
    A31, A32, A33 = A[:, 2, 0], A[:, 2, 1], A[:, 2, 2]
    b1, b2, b3 = b[:, 0], b[:, 1], b[:, 2]
    Exx, Exy, Exz = Sxx + mx**2, Sxy + mx * my, Sxz + mx * mz
    Eyy, Eyz, Ezz = Syy + my**2, Syz + my * mz, Szz + mz**2
    Exxy = Sxx * my + 2 * Sxy * mx + (mx**2) * my
    Exxz = Sxx * mz + 2 * Sxz * mx + (mx**2) * mz
    Exyy = Syy * mx + 2 * Sxy * my + (my**2) * mx
    Exzz = Szz * mx + 2 * Sxz * mz + (mz**2) * mx
    Exyz = Sxy * mz + Sxz * my + Syz * mx + mx * my * mz
    Exxyy = Sxx * (my**2 + Syy) + Syy * (mx**2) + 4 * Sxy * mx * my + (mx * my)**2 + 2 * (Sxy**2)
    Exxzz = Sxx * (mz**2 + Szz) + Szz * (mx**2) + 4 * Sxz * mx * mz + (mx * mz)**2 + 2 * (Sxz**2)
    EX = ((vS**2) * (Eyy + Exx - 2 * Exy) + (A11**2) * Exx + (A12**2) * Eyy + (A13**2) * Ezz + b1**2 +
          2 * (A11 * A12 * Exy + A11 * A13 * Exz - b1 * A11 * mx + A12 * A13 * Eyz - b1 * A12 * my - b1 * A13 * mz +
               vS * (A11 * Exy + A12 * Eyy + A13 * Eyz - b1 * my - A11 * Exx - A12 * Exy - A13 * Exz + b1 * mx)))
    EY = ((vR**2) * Exx + Eyy + Exxzz + (A21**2) * Exx + (A22**2) * Eyy + (A23**2) * Ezz + b2**2 +
          2 * (Exyz - A21 * Exy - A22 * Eyy - A23 * Eyz - A21 * Exxz - A22 * Exyz - A23 * Exzz + A21 * A22 * Exy +
               A21 * A23 * Exz + A22 * A23 * Eyz - vR * (Exy + Exxz - A21 * Exx - A22 * Exy - A23 * Exz) -
               b2 * (vR * mx - my - Exz + A21 * mx + A22 * my + A23 * mz)))
    EZ = (Exxyy + (vB**2) * Ezz + (A31**2) * Exx + (A32**2) * Eyy + (A33**2) * Ezz + b3**2 +
          2 * (A31 * Exxy + A32 * Exyy + A33 * Exyz + A31 * A32 * Exy + A31 * A33 * Exz + A32 * A33 * Eyz -
               vB * (Exyz + A31 * Exz + A32 * Eyz + A33 * Ezz) - b3 * (Exy - vB * mz + A31 * mx + A32 * my + A33 * mz)))
    Efg = np.stack((EX, EY, EZ), axis=1)
    # first derivatives of the moments (sys_lorenz_63.py:337-400)
    dExx_dmx, dExy_dmx, dExz_dmx = 2 * mx, my, mz
    dEyy_dmy, dExy_dmy, dEyz_dmy = 2 * my, mx, mz
    dEzz_dmz, dExz_dmz, dEyz_dmz = 2 * mz, mx, my
    dExxy_dmx, dExxz_dmx, dExyy_dmx, dExzz_dmx, dExyz_dmx = 2 * Exy, 2 * Exz, Eyy, Ezz, Eyz
    dExxy_dmy, dExyy_dmy, dExyz_dmy = Exx, 2 * Exy, Exz
    dExxz_dmz, dExzz_dmz, dExyz_dmz = Exx, 2 * Exz, Exy
    dExxy_dSxx, dExxz_dSxx = my, mz
    dExxy_dSxy, dExyy_dSxy, dExyz_dSxy = 2 * mx, 2 * my, mz
    dExzz_dSxz, dExyz_dSxz, dExxz_dSxz = 2 * mz, my, 2 * mx
    dExyy_dSyy, dExyz_dSyz, dExzz_dSzz = mx, mx, mx
    dExxyy_dmx, dExxzz_dmx, dExxyy_dmy, dExxzz_dmz = 2 * Exyy, 2 * Exzz, 2 * Exxy, 2 * Exxz
    dExxyy_dSxx, dExxzz_dSxx = Eyy, Ezz
    dExxyy_dSxy, dExxzz_dSxz = 4 * Exy, 4 * Exz
    dExxyy_dSyy, dExxzz_dSzz = Exx, Exx
    dmx1 = (dExx_dmx * (vS**2 + A11**2) +
            2 * (dExy_dmx * (-vS**2 + vS * A11 - vS * A12 + A11 * A12) + dExz_dmx * (A11 - vS) * A13 -
                 vS * A11 * dExx_dmx + b1 * (vS - A11)))
    dmx2 = (dExxzz_dmx + dExx_dmx * (vR**2 + A21**2) +
            2 * (dExy_dmx * (-vR + vR * A22 - A21 + A21 * A22) + dExz_dmx * (vR * A23 + b2 + A21 * A23) +
                 dExyz_dmx * (1 - A22) - vR * dExxz_dmx + vR * A21 * dExx_dmx - A21 * dExxz_dmx -
                 A23 * dExzz_dmx - b2 * (vR + A21)))
    dmx3 = (dExxyy_dmx + (A31**2) * dExx_dmx +
            2 * (dExy_dmx * (A31 * A32 - b3) + dExz_dmx * (A33 - vB) * A31 + dExyz_dmx * (A33 - vB) +
                 A31 * dExxy_dmx + A32 * dExyy_dmx - A31 * b3))
    dmy1 = (dEyy_dmy * (vS**2 + A12**2) +
            2 * (dExy_dmy * (-(vS**2) + vS * A11 - vS * A12 + A11 * A12) + dEyz_dmy * (vS + A12) * A13 +
                 vS * A12 * dEyy_dmy - b1 * (vS + A12)))
    dmy2 = (dEyy_dmy * (1 + A22**2) +
            2 * (dExy_dmy * (-vR + vR * A22 - A21 + A21 * A22) + dExyz_dmy * (1 - A22) - A22 * dEyy_dmy +
                 dEyz_dmy * (A22 * A23 - A23) + b2 * (1 - A22)))
    dmy3 = (dExxyy_dmy + (A32**2) * dEyy_dmy +
            2 * (dExyz_dmy * (A33 - vB) + A31 * dExxy_dmy + A32 * dExyy_dmy + dExy_dmy * (A31 * A32 - b3) +
                 dEyz_dmy * (A33 - vB) * A32 - A32 * b3))
    dmz1 = (A13**2) * dEzz_dmz + 2 * (dEyz_dmz * (vS + A12) + dExz_dmz * (A11 - vS) - b1) * A13
    dmz2 = (dExxzz_dmz + (A23**2) * dEzz_dmz +
            2 * (dExxz_dmz * (-vR - A21) + dExz_dmz * (vR * A23 + b2 + A21 * A23) + dExyz_dmz * (1 - A22) +
                 dEyz_dmz * (A22 * A23 - A23) - A23 * (dExzz_dmz + b2)))
    dmz3 = (dEzz_dmz * (vB**2 + A33**2) +
            2 * ((A33 - vB) * (dExyz_dmz + dExz_dmz * A31 + dEyz_dmz * A32 - b3) - vB * A33 * dEzz_dmz))
    iSx, iSy, iSz = diagSigI
    dEdm = 0.5 * np.stack((dmx1 * iSx + dmx2 * iSy + dmx3 * iSz,
                           dmy1 * iSx + dmy2 * iSy + dmy3 * iSz,
                           dmz1 * iSx + dmz2 * iSy + dmz3 * iSz), axis=1)
    dSxx = (iSx * ((vS - A11)**2) +
            iSy * (dExxzz_dSxx + ((vR + A21)**2) - 2 * dExxz_dSxx * (vR + A21)) +
            iSz * (dExxyy_dSxx + (A31**2) + 2 * A31 * dExxy_dSxx))
    dSxy = (iSx * 2 * (vS * A11 - vS**2 - vS * A12 + A11 * A12) +
            iSy * 2 * ((vR * A22 - vR - A21 + A21 * A22) + dExyz_dSxy * (1 - A22)) +
            iSz * (dExxyy_dSxy + 2 * (dExyz_dSxy * (A33 - vB) + A31 * dExxy_dSxy + A32 * dExyy_dSxy + (A31 * A32 - b3))))
    dSxz = (iSx * 2 * (A11 - vS) * A13 +
            iSy * (dExxzz_dSxz + 2 * ((vR * A23 + b2 + A21 * A23) + dExyz_dSxz * (1 - A22) -
                                      dExxz_dSxz * (vR + A21) - A23 * dExzz_dSxz)) +
            iSz * 2 * ((A33 - vB) * A31 + dExyz_dSxz * (A33 - vB)))
    dSyy = (iSx * ((vS + A12)**2) + iSy * ((1 - A22)**2) +
            iSz * (dExxyy_dSyy + (A32**2) + 2 * A32 * dExyy_dSyy))
    dSyz = (iSx * 2 * (vS + A12) * A13 +
            iSy * 2 * (dExyz_dSyz * (1 - A22) + (A22 - 1) * A23) +
            iSz * 2 * (dExyz_dSyz * (A33 - vB) + (A33 - vB) * A32))
    dSzz = (iSx * (A13**2) + iSy * (dExxzz_dSzz + (A23**2) - 2 * A23 * dExzz_dSzz) +
            iSz * ((vB - A33)**2))
    N = m.shape[0]
    dEdS = np.empty((N, 3, 3))
    dEdS[:, 0, 0], dEdS[:, 0, 1], dEdS[:, 0, 2] = dSxx, dSxy, dSxz
    dEdS[:, 1, 0], dEdS[:, 1, 1], dEdS[:, 1, 2] = dSxy, dSyy, dSyz
    dEdS[:, 2, 0], dEdS[:, 2, 1], dEdS[:, 2, 2] = dSxz, dSyz, dSzz
    return Efg, dEdm, 0.5 * dEdS


def l96_ef_edf(m, S, theta):
    """dynamics/sys_lorenz_96_ut.py:260-330, vectorised over t.  The Jacobian
    rows are filled in the reference's assignment order (later wins when indices
    collide for D < 4)."""
    N, D = m.shape
    i = np.arange(D)
    f1, b1, b2 = (i + 1) % D, (i - 1) % D, (i - 2) % D
    Cxx = S[:, f1, b1] - S[:, b2, b1]
    Ef = Cxx + (m[:, f1] - m[:, b2]) * m[:, b1] - m + theta
    Edf = np.zeros((N, D, D))
    Edf[:, i, i] = -1.0
    Edf[:, i, f1] = m[:, b1]
    Edf[:, i, b2] = -m[:, b1]
    Edf[:, i, b1] = m[:, f1] - m[:, b2]
    return Ef, Edf


# --------------------------------------------------------------------------
# energy_mode for the five systems
# --------------------------------------------------------------------------
def energy_mode(system, A, b, m, S, sde):
    """The ``energy_mode(A, b, m, S, sDyn)`` contract of dynamics/sys_*.py:
    returns (Esde, Ef, Edf, dEsde_dm, dEsde_dS, dEsde_dth, dEsde_dSig) plus a dict
    of per-step integrands (``extra``)."""
    dt, idx, theta = sde["dt"], sde["obsX"], sde["theta"]
    if system == "ou":                                      # sys_ornstein_uhlenbeck.py:120-159
        SigInv = 1.0 / sde["Sig"]
        Ex2 = mom_gauss(m, S, 2)
        Q1, Q2 = (theta - A)**2, A * b
        varQ = Ex2 * Q1 + 2 * m * (theta - A) * b + b**2
        Esde = 0.5 * SigInv * mytrapz(varQ, dt, idx)
        Ef = -theta * m
        Edf = -theta * np.ones(m.shape)
        dEdm = SigInv * (m * (theta - A)**2 + theta * b - Q2)
        dEdS = 0.5 * SigInv * Q1
        dth_t = Ex2 * (theta - A) + m * b
        dth = SigInv * mytrapz(dth_t, dt, idx)
        dSig = -SigInv * Esde
        return Esde, Ef, Edf, dEdm, dEdS, dth, dSig, dict(Esde_t=0.5 * SigInv * varQ, dth_t=dth_t, dSig_t=None)
    if system == "dw":                                      # sys_double_well.py:129-187
        SigInv = 1.0 / sde["Sig"]
        c = 4.0 * theta + A
        c2 = c**2
        Ex2, Ex3, Ex4, Ex6 = (mom_gauss(m, S, o) for o in (2, 3, 4, 6))
        varQ = 16.0 * Ex6 - 8.0 * c * Ex4 + 8.0 * b * Ex3 + c2 * Ex2 - 2.0 * b * c * m + b**2
        Esde = 0.5 * SigInv * mytrapz(varQ, dt, idx)
        Ef = 4.0 * (theta * m - Ex3)
        Edf = 4.0 * (theta - 3 * Ex2)
        Dm = {o: mom_gauss(m, S, o, "Dm") for o in (2, 3, 4, 6)}
        DS = {o: mom_gauss(m, S, o, "DS") for o in (2, 3, 4, 6)}
        dEdm = 0.5 * SigInv * (16.0 * Dm[6] - 8.0 * c * Dm[4] + 8.0 * b * Dm[3] + c2 * Dm[2] - 2.0 * b * c)
        dEdS = 0.5 * SigInv * (16.0 * DS[6] - 8.0 * c * DS[4] + 8.0 * b * DS[3] + c2 * DS[2])
        dth_t = c * Ex2 - 4.0 * Ex4 - b * m
        dth = 4.0 * SigInv * mytrapz(dth_t, dt, idx)
        dSig = -Esde * SigInv
        return Esde, Ef, Edf, dEdm, dEdS, dth, dSig, dict(Esde_t=0.5 * SigInv * varQ, dth_t=dth_t, dSig_t=None)

    SigInv = np.linalg.inv(sde["Sig"])
    dSI = np.diag(SigInv)
    flag = None
    if system == "l63":                                     # sys_lorenz_63.py:125-244
        Efg, dEdm, dEdS = l63_energy_dm_dS(A, b, m, S, dSI, theta)
        Esde_t = 0.5 * Efg.dot(dSI)
        Ef, Edf = _l63_ef_edf(m, S, theta)
        dth_t = l63_drift_theta(A, b, m, S, theta)
        dSig_t = Efg
    elif system == "l63_ut":                                # sys_lorenz_63_ut.py:132-266
        Esde_t, mbar, dEdm, dEdS, flag = _ut_energy(lambda x: lorenz63(x, theta), A, b, m, S, dSI)
        Ef, Edf = _l63_ef_edf(m, S, theta)
        dth_t = l63_drift_theta(A, b, m, S, theta)
        dSig_t = mbar
    elif system == "l96_ut":                                # sys_lorenz_96_ut.py:131-258
        Esde_t, mbar, dEdm, dEdS, flag = _ut_energy(lambda x: lorenz96(x, theta), A, b, m, S, dSI)
        Ef, Edf = l96_ef_edf(m, S, theta)
        dth_t = Ef + np.einsum("nj,nij->ni", m, A) - b       # :244
        dSig_t = mbar
    else:
        raise ValueError(f"unknown system {system!r}")
    Esde = mytrapz(Esde_t[:, None], dt, idx)                 # shape (1,)
    dth = dSI * mytrapz(dth_t, dt, idx)
    dSig = -0.5 * SigInv.dot(np.diag(mytrapz(dSig_t, dt, idx))).dot(SigInv)
    return Esde, Ef, Edf, dEdm, dEdS, dth, dSig, dict(Esde_t=Esde_t, dth_t=dth_t, dSig_t=dSig_t, flag=flag)


# --------------------------------------------------------------------------
# gradient assembly: core/gradL_Ab.py
# --------------------------------------------------------------------------
def grad_L_Ab(A, b, m, S, lam, Psi, Ef, Edf, Sigma, dt, outer=False):
    """core/gradL_Ab.py:50-180.  n-D: the two ``x.T.dot(mt)`` terms are inner
    products of 1-D arrays, i.e. scalars subtracted from every element
    (gradL_Ab.py:167,170); ``outer=True`` gives the mathematically intended outer
    products instead (not the reference's behaviour)."""
    if A.ndim == 1 or A.shape[1] == 1:                      # gradL_Ab.py:85-109
        SigInv = 1.0 / Sigma
        db = SigInv * (-Ef - A * m + b)
        dA = SigInv * (Edf + A) * S - db * m
        gLA = dA - lam * m - 2.0 * Psi * S
        gLb = db + lam
        return np.concatenate((dt * gLA, dt * gLb))
    SigInv = np.linalg.inv(Sigma)
    db = (-Ef - np.einsum("nj,nij->ni", m, A) + b).dot(SigInv.T)          # :164
    core = np.einsum("ij,njk,nkl->nil", SigInv, Edf + A, S)               # :167
    if outer:
        t1 = db[:, :, None] * m[:, None, :]
        t2 = lam[:, :, None] * m[:, None, :]
    else:
        t1 = np.einsum("ni,ni->n", db, m)[:, None, None]
        t2 = np.einsum("ni,ni->n", lam, m)[:, None, None]
    gLA = core - t1 - t2 - 2.0 * np.einsum("nij,njk->nik", Psi, S)        # :170
    gLb = db + lam                                                        # :173
    return np.concatenate(((dt * gLA).ravel()[:, None], (dt * gLb).ravel()[:, None]))


# --------------------------------------------------------------------------
# the two entry points: core/varFreeEnergy.py, core/gradL_Ab.py
# --------------------------------------------------------------------------
def unpack_Ab(Ab, N, D):
    """core/varFreeEnergy.py:64-69."""
    Ab = np.asarray(Ab, dtype=np.float64)
    if D == 1:
        Ab = Ab.reshape(-1, 1)
        return Ab[:N], Ab[N:]
    Ab = Ab.reshape(-1)
    K = N * D * D
    return Ab[:K].reshape(N, D, D), Ab[K:].reshape(N, D)


def evaluate(system, Ab, m0, S0, sde, with_grad=True, outer=False):
    """varFreeEnergy (core/varFreeEnergy.py:9-113) followed by gradL_Ab
    (core/gradL_Ab.py:11-180).  Returns a dict with F, grad and every
    intermediate."""
    N, D, dt, method = sde["N"], sde["D"], sde["dt"], sde["ode_method"]
    A, b = unpack_Ab(Ab, N, D)
    m, S = fwd_scan(A, b, m0, S0, sde["Sig"], dt, method)
    Eobs, dEobs_dm, dEobs_dS = likelihood_gauss(m, S, sde)
    Esde, Ef, Edf, dEdm, dEdS, dth, dSig, extra = energy_mode(system, A, b, m, S, sde)
    lam, Psi = bwd_scan(A, dEdm, dEdS, dEobs_dm, dEobs_dS, dt, method)
    KL0 = kl0_gauss(m0, S0, sde)
    F = float(np.asarray(KL0 + Eobs + Esde).reshape(-1)[0])
    out = dict(F=F, KL0=KL0, Eobs=Eobs, Esde=Esde, m=m, S=S, lam=lam, Psi=Psi, Ef=Ef, Edf=Edf,
               dEsde_dm=dEdm, dEsde_dS=dEdS, dEsde_dth=dth, dEsde_dSig=dSig,
               dEobs_dm=dEobs_dm, dEobs_dS=dEobs_dS, extra=extra)
    if with_grad:
        out["grad"] = grad_L_Ab(A, b, m, S, lam, Psi, Ef, Edf, sde["Sig"], dt, outer=outer)
    return out


def finite_diff(fun, x, h=1.0e-6):
    """auxiliary/numerical.py:89-150: central differences of ``fun`` (a function of the
    column vector alone here) in every coordinate direction, step h."""
    x = np.asarray(x, dtype=np.float64).reshape(-1, 1)
    grad = np.zeros_like(x)
    e = np.zeros_like(x)
    for i in range(x.size):
        e[i] = 1.0
        grad[i] = 0.5 * (fun(x + h * e) - fun(x - h * e)) / h      # :137-141
        e[i] = 0.0
    return grad


# ---------------------------------------------------------------------------
# input preparation: sample paths and starting values (test infrastructure)
# ---------------------------------------------------------------------------
def path_drift(system, x, theta):
    """Drift of the stochastic systems at state x (D,): dynamics/sys_ornstein_uhlenbeck.py:58,
    sys_double_well.py:72, sys_lorenz_63.py:32-34, sys_lorenz_96_ut.py:36-38, in the
    reference's order of operations."""
    if system == "ou":
        return theta * (0 - x)
    if system == "dw":
        return 4 * x * (theta - x ** 2)
    if system in ("l63", "l63_ut"):
        s, r, b = theta
        return np.array([-s * (x[0] - x[1]), (r - x[2]) * x[0] - x[1], x[0] * x[1] - b * x[2]])
    return (np.roll(x, -1) - np.roll(x, +2)) * np.roll(x, +1) - x + theta


def sde_path(system, x0, theta, kdiag, ek, dt, burn=0, dtau=0.0, add0=None):
    """The recursion of ``system_path`` (sys_lorenz_96_ut.py:82-106 and its siblings):
    ``burn`` deterministic Euler steps of size dtau from x0, X[0] = x (+ add0), then
    X[t] = X[t-1] + f(X[t-1]) dt + kdiag * ek[t]."""
    x = np.array(x0, dtype=np.float64).reshape(-1)
    for _ in range(burn):
        x = x + path_drift(system, x, theta) * dtau
    N, D = ek.shape
    X = np.zeros((N, D))
    X[0] = x if add0 is None else x + add0
    for t in range(1, N):
        X[t] = X[t - 1] + path_drift(system, X[t - 1], theta) * dt + ek[t] * kdiag
    return X


def init_ab0(mt0, sig_diag, s0_diag, dt):
    """auxiliary/initialize_Ab0.py:80-93 for given spline means mt0 (N,D): A0 (N,D,D), b0 (N,D)."""
    N, D = mt0.shape
    aii = 0.5 * (sig_diag / s0_diag)
    A0 = np.zeros((N, D, D))
    A0[:, np.arange(D), np.arange(D)] = aii
    b0 = np.zeros((N, D))
    b0[:-1] = np.diff(mt0, axis=0) / dt + aii * mt0[:-1]
    b0[-1] = aii * mt0[-1]
    return A0, b0
