"""The packed symmetric layout (1152 doubles per 40 x 40 matrix) of the Lorenz-96 Euler kernels
against the dense path, the golden vectors recorded from the reference and the oracle.  The
layout has no counterpart in the reference: what is checked is that every kernel produces, in
the kept entries, what its dense form produces (core/fwd_ode.py:87-132,
dynamics/sys_lorenz_96_ut.py:131, core/bwd_ode.py:86-129, core/gradL_Ab.py:156-177)."""
import numpy as np
import pytest

from conftest import load_golden, relerr, sde_from_golden, initial_moments
from oracle import vgpa_oracle as orc
from test_gpu_fused import _l96_problem, FUSED_P

pytestmark = pytest.mark.gpu

TOL = 1e-9
SIZE = 1152


def _packed_index():
    """(row, column) of every kept entry and its position, from the definition of the layout."""
    pos = {}
    off = 0
    for r in range(40):
        width = 16 * (r // 16 + 1)
        for c in range(width):
            if c < 40:
                pos[(r, c)] = off + c
        off += width
    assert off == SIZE
    return pos


def test_pack_and_unpack_follow_the_documented_layout():
    import torch
    from vgpa_amd import engine, _lib
    assert _lib.lib.vgpa_sym_packed_size(40) == SIZE and _lib.lib.vgpa_sym_packed_size(16) == -1
    rng = np.random.default_rng(1)
    X = rng.standard_normal((5, 3, 40, 40))
    X = X + np.swapaxes(X, -1, -2)
    packed = torch.full((5, 3, SIZE), -7.0, dtype=torch.float64, device="cuda")
    out = engine.sym_pack(engine.to_dev(X))
    packed.copy_(out)
    got = packed.cpu().numpy()
    pos = _packed_index()
    for (r, c), q in pos.items():
        assert np.array_equal(got[..., q], X[..., r, c]), (r, c)
    assert len(pos) == 16 * 16 + 16 * 32 + 8 * 40
    back = engine.sym_unpack(out)
    assert np.array_equal(back.cpu().numpy(), X)
    # the padding (columns 40 .. 47 of rows 32 .. 39) is not written
    fresh = torch.full((1, SIZE), -7.0, dtype=torch.float64, device="cuda")
    _lib.call("vgpa_sym_pack", 1, 40, engine._p(engine.to_dev(X[0, 0])), engine._p(fresh), None)
    pad = sorted(set(range(SIZE)) - set(pos.values()))
    assert len(pad) == 64 and bool((fresh[0, pad] == -7.0).all().item())
    with pytest.raises(_lib.VgpaError, match="D == 40"):
        _lib.call("vgpa_sym_pack", 1, 16, engine._p(fresh), engine._p(fresh), None)


@pytest.mark.parametrize("N", [1, 2, 3, 7, 64, 201])
def test_packed_evaluation_equals_dense(N):
    """Forward scan, observation terms, energy kernel and the fused backward scan + gradient on
    packed matrices: the same arithmetic on the same numbers, so the same bits."""
    import torch
    from vgpa_amd import engine
    prob, A, b = _l96_problem(N, FUSED_P, 700 + N)
    assert engine.packed_supported(prob)
    st, gLA, gLb = engine.free_energy_and_gradient(prob, A, b, want_hyper=True)
    sp, pA, pb = engine.free_energy_and_gradient(prob, A, b, want_hyper=True, packed=True)
    assert int(st.status.max().item()) == 0 and int(sp.status.max().item()) == 0
    assert sp.S.shape == (FUSED_P, N, SIZE) and sp.dEdS.shape == sp.S.shape and sp.Psi.shape == sp.S.shape
    assert sp.Edf is None
    assert torch.equal(sp.m, st.m) and torch.equal(engine.sym_unpack(sp.S), st.S)
    assert torch.equal(sp.Eobs, st.Eobs) and torch.equal(sp.dEobs_dm, st.dEobs_dm)
    assert torch.equal(sp.Esde, st.Esde) and torch.equal(sp.Ef, st.Ef) and torch.equal(sp.dEdm, st.dEdm)
    assert torch.equal(engine.sym_unpack(sp.dEdS), st.dEdS)
    assert torch.equal(sp.dth, st.dth) and torch.equal(sp.dSig, st.dSig)
    assert torch.equal(sp.F, st.F)
    assert torch.equal(sp.lam, st.lam) and torch.equal(engine.sym_unpack(sp.Psi), st.Psi)
    assert torch.equal(pA, gLA) and torch.equal(pb, gLb)
    # dEsde/dS alone packed: the intermediate between the energy kernel and the backward scan; S, Psi dense
    sm, mA, mb = engine.free_energy_and_gradient(prob, A, b, want_hyper=True, packed="dEdS")
    assert sm.S.shape == st.S.shape and sm.Psi.shape == st.Psi.shape and sm.dEdS.shape == (FUSED_P, N, SIZE)
    assert torch.equal(sm.S, st.S) and torch.equal(sm.Psi, st.Psi) and torch.equal(sm.lam, st.lam) and torch.equal(sm.F, st.F)
    assert torch.equal(engine.sym_unpack(sm.dEdS), st.dEdS) and torch.equal(sm.dSig, st.dSig)
    assert torch.equal(mA, gLA) and torch.equal(mb, gLb)
    # without the costate rows
    s0, qA, qb = engine.free_energy_and_gradient(prob, A, b, packed=True, costate=False)
    assert s0.Psi.shape == (FUSED_P, 1, SIZE)
    assert torch.equal(engine.sym_unpack(s0.Psi)[:, 0], st.Psi[:, 0])
    assert torch.equal(qA, gLA) and torch.equal(qb, gLb) and torch.equal(s0.F, st.F)


def test_packed_evaluation_vs_golden():
    from vgpa_amd import engine, _lib
    g = load_golden("sys_l96_ut_d40.npz")
    method = "EULER"
    sde = sde_from_golden(g, method)
    m0, S0 = initial_moments(g)
    N, D = int(g["N"]), 40
    An, bn = orc.unpack_Ab(g["Ab"], N, D)
    P = FUSED_P
    prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P)
    rep = lambda x: engine.to_dev(np.broadcast_to(x, (P,) + x.shape))   # noqa: E731
    st, gLA, gLb = engine.free_energy_and_gradient(prob, rep(An), rep(bn), packed=True)
    Psi = engine.sym_unpack(st.Psi)
    for p in (0, P - 1):
        got = np.concatenate((gLA[p].reshape(-1).cpu().numpy(), gLb[p].reshape(-1).cpu().numpy()))
        assert relerr(got, g[method + ".grad"].reshape(-1)) <= TOL
        assert abs(float(st.F[p].item()) - float(g[method + ".F"])) <= TOL * abs(float(g[method + ".F"]))
        assert relerr(Psi[p].cpu().numpy(), g[method + ".Psi"]) <= TOL
        assert relerr(st.lam[p].cpu().numpy(), g[method + ".lam"]) <= TOL


@pytest.mark.parametrize("N", [1, 2, 5, 33])
def test_packed_fused_kernel_with_terminal_state_and_flagged_jumps_vs_oracle(N):
    """The entry point itself on random symmetric inputs, with a dense terminal state and jumps
    at flagged rows; oracle: bwd_scan + grad_L_Ab."""
    import torch
    from vgpa_amd import engine
    rng = np.random.default_rng(900 + N)
    D, dt, P = 40, 0.01, FUSED_P
    prob, _, _ = _l96_problem(max(N, 2), P, 7)
    prob.N = N
    A = 4.0 * np.eye(D)[None, None] + 0.5 * rng.standard_normal((P, N, D, D))
    b, m, Ef = (rng.standard_normal((P, N, D)) for _ in range(3))
    sym = lambda X: 0.5 * (X + np.swapaxes(X, -1, -2))   # noqa: E731
    S = 0.3 * np.eye(D)[None, None] + 0.1 * sym(rng.standard_normal((P, N, D, D)))
    gm, gS = rng.standard_normal((P, N, D)), sym(rng.standard_normal((P, N, D, D)))
    jm = rng.standard_normal((P, N, D))
    flag = (rng.random(N) < 0.5).astype(np.int32)
    jump = sym(rng.standard_normal((D, D)))
    lamT, PsiT = rng.standard_normal((P, D)), sym(rng.standard_normal((P, D, D)))
    d = engine.to_dev
    prob.dt = dt
    prob.obs_flag = torch.from_numpy(flag).to(prob.m0.device)
    lam, Psi, gLA, gLb = engine.bwd_scan_grad(prob, d(A), d(b), d(m), engine.sym_pack(d(S)), d(gm), engine.sym_pack(d(gS)),
                                              d(jm), d(Ef), None, jumpS=d(jump), symmetric=True, state_symmetric=True,
                                              lam_last=d(lamT), Psi_last=d(PsiT), packed=True)
    assert Psi.shape == (P, N, SIZE)
    Psi = engine.sym_unpack(Psi)
    assert torch.equal(Psi, Psi.transpose(-1, -2))
    Sig = np.linalg.inv(prob.SigInv.cpu().numpy())
    jS = flag[:, None, None] * jump[None]
    for p in (0, 1, P - 1):
        lr, Pr = orc.bwd_scan(A[p], gm[p], gS[p], jm[p], jS, dt, "EULER", lam_last=lamT[p], Psi_last=PsiT[p])
        assert relerr(Psi[p].cpu().numpy(), Pr) <= TOL and relerr(lam[p].cpu().numpy(), lr) <= TOL
        _, Edf = orc.l96_ef_edf(m[p], S[p], 8.0)
        gr = orc.grad_L_Ab(A[p], b[p], m[p], S[p], lr, Pr, Ef[p], Edf, Sig, dt).reshape(-1)
        got = np.concatenate((gLA[p].reshape(-1).cpu().numpy(), gLb[p].reshape(-1).cpu().numpy()))
        assert relerr(got, gr) <= TOL


def test_packed_energy_kernel_cholesky_fallback_vs_golden():
    """Every third S(t) of the fixture has a negative eigenvalue: the wavefront kernel flags the
    step and the fallback pass redoes it from the packed S (auxiliary/numerical.py:201-202)."""
    import torch
    from vgpa_amd import engine, _lib
    g = load_golden("l96_ut_nonpd_d40.npz")
    sde = sde_from_golden(g)
    N, D = sde["N"], 40
    A, b = orc.unpack_Ab(g["Ab"], N, D)
    d = engine.to_dev
    sig_inv = d(1.0 / np.diag(sde["Sig"]))
    theta = d(np.array([float(np.ravel(sde["theta"])[0])]))
    S = 0.5 * (g["S"] + np.swapaxes(g["S"], -1, -2))
    assert np.array_equal(S, g["S"])          # the fixture is exactly symmetric
    args = (d(A[None]), d(b[None]), d(g["m"][None]))
    dense = engine.esde(_lib.SYS_L96_UT, *args, d(S[None]), sig_inv, theta, want_edf=False)
    packed = engine.esde_l96_packed(*args, engine.sym_pack(d(S[None])), sig_inv, theta)
    mixed = engine.esde_l96_packed(*args, d(S[None]), sig_inv, theta, packed_S=False)      # dense S in, packed dEdS out
    assert torch.equal(mixed["status"], packed["status"]) and torch.equal(mixed["Esde_t"], packed["Esde_t"])
    assert torch.equal(engine.sym_unpack(mixed["dEdS"]), engine.sym_unpack(packed["dEdS"]))
    assert int((packed["status"] & _lib.ST_CHOL_FALLBACK).sum().item()) >= N // 3 - 1
    assert torch.equal(packed["status"], dense["status"])
    for k in ("Esde_t", "Ef", "dEdm", "dth_t", "dSig_t"):
        assert torch.equal(packed[k], dense[k]), k
    assert torch.equal(engine.sym_unpack(packed["dEdS"]), dense["dEdS"])
    assert relerr(engine.sym_unpack(packed["dEdS"])[0].cpu().numpy(), g["dEsde_dS"]) <= TOL


def test_packed_evaluation_at_the_benchmark_size():
    import torch
    from vgpa_amd import engine
    prob, A, b = _l96_problem(2002, FUSED_P, 9)
    st, gLA, gLb = engine.free_energy_and_gradient(prob, A, b)
    sp, pA, pb = engine.free_energy_and_gradient(prob, A, b, packed=True)
    assert int(st.status.max().item()) == 0
    assert torch.equal(sp.F, st.F) and torch.equal(pA, gLA) and torch.equal(pb, gLb)
    assert torch.equal(engine.sym_unpack(sp.Psi), st.Psi) and torch.equal(engine.sym_unpack(sp.S), st.S)


def test_packed_layout_is_refused_where_no_kernel_takes_it():
    from vgpa_amd import engine, _lib
    few, A, b = _l96_problem(4, 3, 11)                       # too few problems for the one-kernel path
    assert not engine.packed_supported(few)
    with pytest.raises(ValueError, match="packed=True"):
        engine.free_energy_and_gradient(few, A, b, packed=True)
    heun, A, b = _l96_problem(4, FUSED_P, 11, method="HEUN")
    assert not engine.packed_supported(heun)
    with pytest.raises(_lib.VgpaError, match="VGPA_SCAN_PACKED"):
        engine.fwd_scan(heun.method, A, b, heun.m0, heun.S0, heun.Sigma, heun.dt, symmetric=True, packed=True)
    euler, A, b = _l96_problem(4, FUSED_P, 11)
    with pytest.raises(_lib.VgpaError, match="VGPA_SCAN_PACKED"):
        engine.fwd_scan(euler.method, A, b, euler.m0, euler.S0, euler.Sigma, euler.dt, symmetric=False, packed=True)


def test_batched_optimiser_takes_the_packed_path_and_follows_the_dense_trajectory():
    """scg_batched keeps the symmetric intermediates packed where the kernels take them; its iterates
    are those of a run whose evaluations go through the dense kernels (auxiliary/optimize.py:132-238)."""
    import torch
    from vgpa_amd import engine
    from vgpa_amd.auxiliary.optimize import scg_batched
    N = 12
    prob, A, b = _l96_problem(N, FUSED_P, 41)
    assert engine.packed_supported(prob)
    calls = []
    real = engine.free_energy_and_gradient

    def spy(*a, **kw):
        calls.append(bool(kw.get("packed")))
        return real(*a, **kw)

    engine.free_energy_and_gradient = spy
    try:
        A1, b1, F1, stat1 = scg_batched(prob, A, b, nit=4)
    finally:
        engine.free_energy_and_gradient = real
    assert calls and all(calls)

    def dense(Ax, bx, need_grad):
        st, gA, gb = real(prob, Ax, bx)
        return st.F, torch.cat((gA.reshape(FUSED_P, -1), gb.reshape(FUSED_P, -1)), dim=1)

    A2, b2, F2, stat2 = scg_batched(prob, A, b, nit=4, evaluate=dense)
    assert torch.equal(F1, F2) and torch.equal(A1, A2) and torch.equal(b1, b2)
    assert torch.equal(stat1["Fx"], stat2["Fx"])
