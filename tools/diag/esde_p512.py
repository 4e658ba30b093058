"""Energy kernel alone at the benchmark size (512 problems): minimum and median of ten launches."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
import bench
from vgpa_amd import engine, _lib
dev = torch.device("cuda", 0)
D, N, dt, P = 40, 2002, 0.005, 512
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)
m, S = engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True)
ts = []
for _ in range(12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    en = engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts = sorted(ts[2:])
print(f"energy kernel P=512: min {ts[0]:.3f} ms, median {ts[len(ts) // 2]:.3f} ms, max {ts[-1]:.3f} ms")
