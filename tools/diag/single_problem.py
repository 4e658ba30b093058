"""One Lorenz-96 problem (D = 40, N = 2002) evaluated ten times: for `rocprofv3 --kernel-trace --stats`,
which then lists where the time of a single-problem evaluation goes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vgpa_amd import engine, _lib
import bench
dev = torch.device("cuda", 0)
D, N, dt = 40, 2002, 0.005
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, 1, 1, dev)
prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=1, theta=thetas)
for _ in range(10):
    out = engine.free_energy_and_gradient(prob, A, b)
torch.cuda.synchronize()
print("F", float(out[0].F[0]))
