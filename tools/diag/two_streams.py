"""Does running two halves of the batch on two HIP streams overlap the latency-bound
energy kernel of one half with the bandwidth-bound kernels of the other?"""
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

def one(i):
    return engine.free_energy_and_gradient(probs[i], As[i], bs[i])

def run(parallel, delay_ms=0.0):
    """delay_ms: chunk i starts i * delay_ms late on its stream (a spinning one-lane kernel), so that
    the kernels of different chunks that meet on the chip are of different kinds."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        outs = []
        for i in range(nchunk):
            if parallel:
                with torch.cuda.stream(streams[i]):
                    if delay_ms and i:
                        torch.cuda._sleep(int(i * delay_ms * 2.4e6))   # cycles at about 2.4 GHz
                    outs.append(one(i))
            else:
                outs.append(one(i))
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3

def run_whole():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = engine.free_energy_and_gradient(whole, A, b)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 3 * 1e3

run(False); run(True); run_whole()
print(f"P={P}: one call {run_whole():.2f} ms; in {nchunk} chunks: one after the other {run(False):.2f} ms, on {nchunk} streams {run(True):.2f} ms, "
      + ", ".join(f"chunk i delayed by {d} i ms {run(True, d):.2f} ms" for d in (2.0, 4.0, 6.0)))
