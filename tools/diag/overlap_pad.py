"""Overlap experiment with an energy kernel that leaves room on the CUs: the batch in chunks on separate HIP
streams, chunk i started i * delay late, the energy kernel's LDS request padded (VGPA_ESDE_LDS_PAD, a library
built with -DVGPA_EXPERIMENT_LDS_PAD) so that a scan workgroup of another chunk fits beside its workgroups.

  VGPA_HIP_LIBRARY=... VGPA_ESDE_LDS_PAD=13552 python tools/diag/overlap_pad.py [P] [chunks]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from vgpa_amd import engine, _lib
import bench
dev = torch.device("cuda", 0)
D, N, dt = 40, 2002, 0.005
P = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nchunk = int(sys.argv[2]) if len(sys.argv) > 2 else 2
A, b, sde, m0, S0, thetas = bench.synth_problem(D, N, dt, P, 1, dev)
Pc = P // nchunk
probs = [engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=Pc, theta=thetas[i * Pc:(i + 1) * Pc]) for i in range(nchunk)]
As = [A[i * Pc:(i + 1) * Pc] for i in range(nchunk)]
bs = [b[i * Pc:(i + 1) * Pc] for i in range(nchunk)]
streams = [torch.cuda.Stream() for _ in range(nchunk)]
whole = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)
REPS = 4


def run(delay_ms):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        for i in range(nchunk):
            with torch.cuda.stream(streams[i]):
                if delay_ms and i:
                    torch.cuda._sleep(int(i * delay_ms * 2.1e6))
                engine.free_energy_and_gradient(probs[i], As[i], bs[i])
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e3


def run_whole():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(REPS):
        engine.free_energy_and_gradient(whole, A, b)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / REPS * 1e3


run(0.0); run_whole()
print(f"pad {os.environ.get('VGPA_ESDE_LDS_PAD', '0')} B, P={P}: one call {run_whole():.2f} ms; {nchunk} chunks on {nchunk} streams: "
      + ", ".join(f"delay {d} ms: {run(d):.2f}" for d in (0.0, 1.5, 3.0, 4.5, 6.0, 8.0)), flush=True)
