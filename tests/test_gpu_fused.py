"""The backward scan fused with the gradient assembly (vgpa_bwd_scan_grad) against the two
separate kernels, the oracle and the golden vectors recorded from the reference
(core/bwd_ode.py:117-127 + core/gradL_Ab.py:156-177)."""
import numpy as np
import pytest

from conftest import load_golden, relerr, sde_from_golden, initial_moments
from oracle import vgpa_oracle as orc

pytestmark = pytest.mark.gpu

TOL = 1e-9


def _sym(X):
    return 0.5 * (X + np.swapaxes(X, -1, -2))


FUSED_P = 161      # the library fuses from this many problems on


def _l96_problem(N, P, seed, dt=0.005, method="EULER"):
    """A Lorenz-96 D = 40 problem with observations at a few rows and random variational
    parameters; returns (prob, A, b) on the device."""
    import torch
    from vgpa_amd import engine, _lib
    rng = np.random.default_rng(seed)
    D = 40
    dev = torch.device("cuda", 0)
    gen = torch.Generator(device=dev).manual_seed(seed)
    A = 0.2 * torch.randn((P, N, D, D), generator=gen, device=dev, dtype=torch.float64)
    A += 8.0 * torch.eye(D, device=dev, dtype=torch.float64)
    b = 5.0 * torch.randn((P, N, D), generator=gen, device=dev, dtype=torch.float64)
    obsX = np.unique(rng.integers(0, N, size=max(1, N // 3)))
    mu0 = rng.standard_normal(D)
    sde = dict(Sig=np.diag(3.0 + rng.random(D)), theta=8.0, Rig=np.eye(D), D=D, N=N, H=np.eye(D), obsX=obsX,
               obsY=3.0 * rng.standard_normal((obsX.size, D)), px0=dict(mu0=mu0, tau0=0.5 * np.eye(D)), dt=dt,
               ode_method=method)
    prob = engine.make_problem(_lib.SYS_L96_UT, sde, mu0 + 0.1, 0.25 * np.eye(D), P=P,
                               theta=np.linspace(7.0, 9.0, P).reshape(P, 1))
    return prob, A, b


@pytest.mark.parametrize("method", ["EULER", "HEUN", "RK2"])
@pytest.mark.parametrize("N", [1, 2, 3, 4, 7, 64])
def test_fused_evaluation_equals_scan_then_assembly(N, method):
    import torch
    from vgpa_amd import engine
    prob, A, b = _l96_problem(N, FUSED_P, 100 + N, method=method)
    assert prob.sym_fwd and prob.sym_jump
    st = engine.free_energy(prob, A, b, want_edf=False)
    gLA, gLb = engine.gradient(prob, A, b, st)
    st2, fA, fb = engine.free_energy_and_gradient(prob, A, b)
    assert int(st.status.max().item()) == 0
    assert torch.equal(st2.F, st.F)
    if method == "EULER":
        # the recurrence is the same arithmetic in both kernels
        assert torch.equal(st2.Psi, st.Psi) and torch.equal(st2.lam, st.lam)
        assert torch.equal(fb, gLb)
    else:
        # two-stage kernels: the fused Heun kernel carries Psi - dt/2 k1 instead of k1, and the
        # compiler contracts the vector arithmetic of the two code paths differently: equal to
        # rounding, Psi still exactly symmetric
        assert relerr(st2.Psi.cpu().numpy(), st.Psi.cpu().numpy()) <= 1e-13
        assert relerr(st2.lam.cpu().numpy(), st.lam.cpu().numpy()) <= 1e-13
        assert torch.equal(st2.Psi, st2.Psi.transpose(-1, -2))
        assert relerr(fb.cpu().numpy(), gLb.cpu().numpy()) <= 1e-12
    # the D^3 product runs on the matrix cores in the fused kernel: another summation order
    assert relerr(fA.cpu().numpy(), gLA.cpu().numpy()) <= 1e-12


@pytest.mark.parametrize("method", ["EULER", "HEUN", "RK2"])
def test_fused_evaluation_vs_golden_and_oracle(method):
    from vgpa_amd import engine, _lib
    g = load_golden("sys_l96_ut_d40.npz")
    sde = sde_from_golden(g, method)
    m0, S0 = initial_moments(g)
    N, D = int(g["N"]), 40
    An, bn = orc.unpack_Ab(g["Ab"], N, D)
    P = FUSED_P        # copies of the recorded problem: the first and the last are compared
    prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P)
    rep = lambda x: engine.to_dev(np.broadcast_to(x, (P,) + x.shape))   # noqa: E731
    st, gLA, gLb = engine.free_energy_and_gradient(prob, rep(An), rep(bn))
    for p in (0, P - 1):
        got = np.concatenate((gLA[p].reshape(-1).cpu().numpy(), gLb[p].reshape(-1).cpu().numpy()))
        assert relerr(got, g[method + ".grad"].reshape(-1)) <= TOL
        assert abs(float(st.F[p].item()) - float(g[method + ".F"])) <= TOL * abs(float(g[method + ".F"]))
        if method + ".Psi" in g:       # the intermediates were recorded for the Euler run only
            assert relerr(st.Psi[p].cpu().numpy(), g[method + ".Psi"]) <= TOL
            assert relerr(st.lam[p].cpu().numpy(), g[method + ".lam"]) <= TOL


@pytest.mark.parametrize("method", ["EULER", "HEUN", "RK2"])
@pytest.mark.parametrize("N", [1, 2, 5, 33])
def test_fused_kernel_with_terminal_state_and_flagged_jumps_vs_oracle(N, method):
    """The entry point itself on random symmetric inputs, with a terminal state (time windows
    hand one in) and jumps at flagged rows; oracle: bwd_scan + grad_L_Ab."""
    import torch
    from vgpa_amd import engine, _lib
    rng = np.random.default_rng(500 + N)
    D, dt, P = 40, 0.01, FUSED_P
    prob, _, _ = _l96_problem(max(N, 2), P, 7, method=method)
    prob.N = N
    A = 4.0 * np.eye(D)[None, None] + 0.5 * rng.standard_normal((P, N, D, D))
    b, m, Ef = (rng.standard_normal((P, N, D)) for _ in range(3))
    S = 0.3 * np.eye(D)[None, None] + 0.1 * _sym(rng.standard_normal((P, N, D, D)))
    gm, gS = rng.standard_normal((P, N, D)), _sym(rng.standard_normal((P, N, D, D)))
    jm = rng.standard_normal((P, N, D))
    flag = (rng.random(N) < 0.5).astype(np.int32)
    jump = _sym(rng.standard_normal((D, D)))
    lamT, PsiT = rng.standard_normal((P, D)), _sym(rng.standard_normal((P, D, D)))
    d = engine.to_dev
    prob.dt = dt
    prob.obs_flag = torch.from_numpy(flag).to(prob.m0.device)
    lam, Psi, gLA, gLb = engine.bwd_scan_grad(prob, d(A), d(b), d(m), d(S), d(gm), d(gS), d(jm), d(Ef), None,
                                              jumpS=d(jump), symmetric=True, state_symmetric=True,
                                              lam_last=d(lamT), Psi_last=d(PsiT))
    assert torch.equal(Psi, Psi.transpose(-1, -2))
    Sig = np.linalg.inv(prob.SigInv.cpu().numpy())
    jS = flag[:, None, None] * jump[None]
    for p in (0, 1, P - 1):
        lr, Pr = orc.bwd_scan(A[p], gm[p], gS[p], jm[p], jS, dt, method, lam_last=lamT[p], Psi_last=PsiT[p])
        assert relerr(Psi[p].cpu().numpy(), Pr) <= TOL and relerr(lam[p].cpu().numpy(), lr) <= TOL
        _, Edf = orc.l96_ef_edf(m[p], S[p], 8.0)
        gr = orc.grad_L_Ab(A[p], b[p], m[p], S[p], lr, Pr, Ef[p], Edf, Sig, dt).reshape(-1)
        got = np.concatenate((gLA[p].reshape(-1).cpu().numpy(), gLb[p].reshape(-1).cpu().numpy()))
        assert relerr(got, gr) <= TOL


@pytest.mark.parametrize("system,fname,method", [("l96_ut", "sys_l96_ut_d6.npz", "HEUN"), ("l63", "sys_l63.npz", "EULER"),
                                                 ("ou", "sys_ou.npz", "RK4"), ("l96_ut", "sys_l96_ut_d40.npz", "RK2")])
def test_entry_point_runs_the_two_kernels_for_every_other_configuration(system, fname, method):
    import torch
    from vgpa_amd import engine, _lib
    g = load_golden(fname)
    sde = sde_from_golden(g, method)
    m0, S0 = initial_moments(g)
    N, D = int(g["N"]), int(g["D"])
    An, bn = orc.unpack_Ab(g["Ab"], N, D)
    A = engine.to_dev(An.reshape((1, N) if D == 1 else (1, N, D, D)))
    b = engine.to_dev(bn.reshape((1, N) if D == 1 else (1, N, D)))
    prob = engine.make_problem(_lib.SYSTEMS[system], sde, m0, S0, P=1)
    st, gLA, gLb = engine.free_energy_and_gradient(prob, A, b)
    got = np.concatenate((gLA.reshape(-1).cpu().numpy(), gLb.reshape(-1).cpu().numpy()))
    assert relerr(got, g[f"{method}.grad"].reshape(-1)) <= TOL
    st1 = engine.free_energy(prob, A, b, want_edf=(D == 1))
    g1 = engine.gradient(prob, A, b, st1)
    assert torch.equal(g1[0], gLA) and torch.equal(g1[1], gLb) and torch.equal(st1.Psi, st.Psi)


def test_fused_evaluation_at_the_benchmark_size():
    """N = 2002 rows: the fused kernel against the two separate ones."""
    import torch
    from vgpa_amd import engine
    prob, A, b = _l96_problem(2002, FUSED_P, 9)
    st = engine.free_energy(prob, A, b, want_edf=False)
    gLA, gLb = engine.gradient(prob, A, b, st)
    st2, fA, fb = engine.free_energy_and_gradient(prob, A, b)
    assert int(st.status.max().item()) == 0
    assert torch.equal(st2.Psi, st.Psi) and torch.equal(st2.lam, st.lam) and torch.equal(fb, gLb)
    assert relerr(fA.cpu().numpy(), gLA.cpu().numpy()) <= 1e-12


@pytest.mark.parametrize("method", ["EULER", "HEUN", "RK2"])
@pytest.mark.parametrize("N", [1, 2, 9])
def test_fused_evaluation_without_costate_rows(N, method):
    """costate=False: the one-kernel path stores Psi(0) only; F, lam and the gradient are the same
    bits as with every row stored.  Where the library runs two kernels the argument changes nothing."""
    import torch
    from vgpa_amd import engine, _lib
    prob, A, b = _l96_problem(N, FUSED_P, 300 + N, method=method)
    st, gLA, gLb = engine.free_energy_and_gradient(prob, A, b)
    st2, fA, fb = engine.free_energy_and_gradient(prob, A, b, costate=False)
    assert st2.Psi.shape == (FUSED_P, 1, 40, 40)
    assert torch.equal(st2.Psi[:, 0], st.Psi[:, 0]) and torch.equal(st2.lam, st.lam) and torch.equal(st2.F, st.F)
    assert torch.equal(fA, gLA) and torch.equal(fb, gLb)
    # fewer problems than the one-kernel path takes: every row is there
    small, As, bs = _l96_problem(N, 3, 300 + N, method=method)
    st3, _, _ = engine.free_energy_and_gradient(small, As, bs, costate=False)
    assert st3.Psi.shape == (3, N, 40, 40)
    sf = _lib.SCAN_SYMMETRIC | _lib.SCAN_STATE_SYMMETRIC
    assert _lib.lib.vgpa_bwd_scan_grad_is_fused(small.method, FUSED_P, 40, sf, _lib.GRAD_SIGINV_DIAG, 0, 0, _lib.SYS_L96_UT) == 1
    assert _lib.lib.vgpa_bwd_scan_grad_is_fused(small.method, 3, 40, sf, _lib.GRAD_SIGINV_DIAG, 0, 0, _lib.SYS_L96_UT) == 0
    # the flag itself is refused where the assembly kernel needs every row
    en = dict(dEdm=st3.dEdm, dEdS=st3.dEdS)
    lam, Psi = torch.empty_like(st3.dEdm), torch.empty((3, 1, 40, 40), dtype=torch.float64, device=As.device)
    gA, gb = torch.empty_like(As), torch.empty_like(bs)
    p_ = engine._p
    with pytest.raises(_lib.VgpaError, match="VGPA_SCAN_PSI_ROW0"):
        _lib.call("vgpa_bwd_scan_grad", small.method, 3, N, 40, p_(As), p_(bs), p_(st3.m), p_(st3.S), p_(en["dEdm"]),
                  p_(en["dEdS"]), p_(st3.dEobs_dm), None, p_(small.obs_flag), p_(st3.jumpS), 0, p_(st3.Ef), None,
                  small.system, p_(small.theta), 1, p_(small.SigInv), 0, float(small.dt), sf | _lib.SCAN_PSI_ROW0,
                  _lib.GRAD_SIGINV_DIAG, None, None, p_(lam), p_(Psi), p_(gA), p_(gb), None)


@pytest.mark.parametrize("packed", [False, True])
def test_parameter_arrays_aligned_to_eight_bytes_only(packed):
    """The D = 40 kernels move 16-byte pieces; a view that starts on an odd double (8-byte aligned, as any
    slice of a float64 buffer may be) must give the same bits, for inputs and for the gradient outputs."""
    import torch
    from vgpa_amd import engine
    prob, A, b = _l96_problem(33, FUSED_P, 5)

    def shifted(t):
        flat = torch.empty(t.numel() + 1, dtype=t.dtype, device=t.device)
        view = flat[1:].view(t.shape)
        view.copy_(t)
        return view

    A2, b2 = shifted(A), shifted(b)
    assert A2.data_ptr() % 16 == 8 and b2.data_ptr() % 16 == 8
    st, gA, gb = engine.free_energy_and_gradient(prob, A, b, packed=packed)
    out = (shifted(torch.empty_like(A)), shifted(torch.empty_like(b)))
    st2, gA2, gb2 = engine.free_energy_and_gradient(prob, A2, b2, out=out, packed=packed)
    assert gA2.data_ptr() == out[0].data_ptr()
    full = engine.sym_unpack if packed else (lambda x: x)     # the padding of a packed row is never written
    assert torch.equal(st.F, st2.F) and torch.equal(full(st.Psi), full(st2.Psi)) and torch.equal(full(st.S), full(st2.S))
    assert torch.equal(gA, gA2) and torch.equal(gb, gb2)
