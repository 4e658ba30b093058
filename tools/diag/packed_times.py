"""Dense against packed symmetric matrices at the benchmark size: ms per evaluation and per kernel.
Usage: python tools/diag/packed_times.py [P] [N] [reps]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from vgpa_amd import engine
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
from test_gpu_fused import _l96_problem

P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2002
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
prob, A, b = _l96_problem(N, P, 3)
out = (torch.empty_like(A), torch.empty_like(b))


def timed(fn, n=reps):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for packed in (False, True):
    for costate in (True, False):
        t = timed(lambda: engine.free_energy_and_gradient(prob, A, b, out=out, costate=costate, packed=packed))
        print(f"packed={packed!s:5} costate={costate!s:5}  {t:7.3f} ms/evaluation   {P * N / t / 1e3:6.2f} M units/s", flush=True)
    m, S = engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True, packed=packed)
    tf = timed(lambda: engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True, packed=packed))
    if packed:
        te = timed(lambda: engine.esde_l96_packed(A, b, m, S, prob.sig_inv_diag, prob.theta, want_hyper=False))
        en = engine.esde_l96_packed(A, b, m, S, prob.sig_inv_diag, prob.theta, want_hyper=False)
    else:
        te = timed(lambda: engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False))
        en = engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False)
    Eobs, dm, jS = engine.obs_terms(prob, m, S, packed=packed)
    tb = {}
    for costate in (True, False):
        tb[costate] = timed(lambda: engine.bwd_scan_grad(prob, A, b, m, S, en["dEdm"], en["dEdS"], dm, en["Ef"], None, jumpS=jS,
                                                         symmetric=True, state_symmetric=True, out=out, costate=costate,
                                                         packed=packed))
    print(f"   kernels: forward {tf:6.3f}  energy {te:6.3f}  backward+gradient {tb[True]:6.3f} (Psi(0) only: {tb[False]:6.3f})", flush=True)
