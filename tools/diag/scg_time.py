"""Time per iteration of the batched SCG against the bare evaluations (Lorenz-96, D=40)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vgpa_amd import engine, _lib
from vgpa_amd.auxiliary.optimize import scg_batched
import bench
dev = torch.device("cuda", 0)
D, N, dt = 40, 2002, 0.005
P = int(sys.argv[1]) if len(sys.argv) > 1 else 128
nit = int(sys.argv[2]) if len(sys.argv) > 2 else 6
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)
def ev():
    return engine.free_energy_and_gradient(prob, A, b)
ev(); torch.cuda.synchronize()
t0 = time.perf_counter(); ev(); torch.cuda.synchronize(); t_eval = time.perf_counter() - t0
scg_batched(prob, A, b, nit=2); torch.cuda.synchronize()
t0 = time.perf_counter()
Ao, bo, F, stat = scg_batched(prob, A, b, nit=nit)
torch.cuda.synchronize(); t_scg = time.perf_counter() - t0
print(f"P={P}: one F+grad evaluation {t_eval*1e3:.1f} ms; SCG {nit} iterations {t_scg*1e3:.1f} ms = {t_scg/nit*1e3:.1f} ms per iteration "
      f"({t_scg/nit/t_eval:.2f} evaluations' worth); F[0] {float(stat['Fx'][0,0]):.3f} -> {float(F[0]):.3f}; peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
