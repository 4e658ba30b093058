import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
from vgpa_amd import engine
from test_gpu_fused import _l96_problem
P, N = 512, 2002
prob, A, b = _l96_problem(N, P, 3)
out = (torch.empty_like(A), torch.empty_like(b))
def timed(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ref = engine.free_energy_and_gradient(prob, A, b)
for packed in (False, "dEdS", True):
    t = timed(lambda: engine.free_energy_and_gradient(prob, A, b, out=out, packed=packed))
    st, gA, gb = engine.free_energy_and_gradient(prob, A, b, packed=packed)
    same = torch.equal(gA, ref[1]) and torch.equal(gb, ref[2]) and torch.equal(st.F, ref[0].F)
    m, S = engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True, packed=(packed is True))
    if packed:
        te = timed(lambda: engine.esde_l96_packed(A, b, m, S, prob.sig_inv_diag, prob.theta, want_hyper=False, packed_S=(packed is True)))
        en = engine.esde_l96_packed(A, b, m, S, prob.sig_inv_diag, prob.theta, want_hyper=False, packed_S=(packed is True))
    else:
        te = timed(lambda: engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False))
        en = engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False)
    Eobs, dm, jS = engine.obs_terms(prob, m, S, packed=(packed is True))
    tb = timed(lambda: engine.bwd_scan_grad(prob, A, b, m, S, en["dEdm"], en["dEdS"], dm, en["Ef"], None, jumpS=jS, symmetric=True, state_symmetric=True, out=out, packed=packed))
    print(f"packed={packed!s:5}: evaluation {t:7.3f} ms  energy {te:6.3f}  backward+gradient {tb:6.3f}  same bits {same}", flush=True)
