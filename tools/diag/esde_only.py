"""Run only the Lorenz-96 energy kernel a few times (for rocprofv3 counter passes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vgpa_amd import engine, _lib
import bench
dev = torch.device("cuda", 0)
D, N, dt, P = 40, 2002, 0.005, int(sys.argv[1]) if len(sys.argv) > 1 else 256
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)
m, S = engine.fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=True)
for _ in range(3):
    en = engine.esde(prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False)
torch.cuda.synchronize()
print("done", float(en["Esde_t"].sum()))
