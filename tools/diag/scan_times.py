"""Forward scan and fused backward scan + gradient of the Lorenz-96 D = 40 Euler case alone, for P = 1, 128
and 512 problems: median of eight launches (A/B runs of library builds, VGPA_HIP_LIBRARY)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from vgpa_amd import engine, _lib
dev = torch.device("cuda", 0)
D, N, dt = 40, 2002, 0.005


def med(fn):
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts[2:])[4]


for P in (1, 128, 512):
    A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
    prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)
    m, S = engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True)
    Eobs, dEobs_dm, jumpS = engine.obs_terms(prob, m, S)
    en = engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False)
    tf = med(lambda: engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True))
    tb = med(lambda: engine.bwd_scan_grad(prob, A, b, m, S, en["dEdm"], en["dEdS"], dEobs_dm, en["Ef"], en["Edf"],
                                          jumpS=jumpS, symmetric=True, state_symmetric=True))
    print(f"P={P:4d}: forward scan {tf:.3f} ms, backward scan + gradient {tb:.3f} ms", flush=True)
    del A, b, m, S, en
