import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from vgpa_amd import engine, _lib
dev = torch.device("cuda", 0)
D, N, dt, P = 40, 2002, 0.005, 256
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)
m, S = engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True)
for hyper in (False, True):
    for edf in (False, True):
        ts = []
        for _ in range(4):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            en = engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, edf, hyper)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"hyper={hyper} edf={edf}: {min(ts):.2f} ms")
