#!/usr/bin/env python3
"""CPU rate of the reference and of the NumPy oracle, one process per core.

BUILD-CONTAINER ONLY: imports the reference from /root/reference (read-only, with the
three NumPy-2 compatibility patches of oracle/make_golden.py) and times

    varFreeEnergy(Ab, mParam) + gradL_Ab(Ab, mParam)        core/varFreeEnergy.py:9, core/gradL_Ab.py:11

on the Lorenz-96 (D = 40, dt = 0.005, Euler) benchmark inputs, then the oracle
(oracle/vgpa_oracle.py) on the same inputs.  Every worker process pins BLAS to one thread
and evaluates its own independent problem; the rate is the sum over the workers, in
(problem x time step) evaluations per second.  The reference does not exist on the GPU
box, so the numbers this prints are kept as a tracked file under profiles/; bench.py times the oracle
there and the ratio printed here converts that into a rate of the reference.

Usage:  python tools/time_reference.py [--steps N] [--procs P] [--repeat R]
"""
import argparse
import json
import math
import multiprocessing as mp
import os
import sys
import time

os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("OMP_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def bench_inputs(D, N, dt, seed):
    """The synthetic Lorenz-96 inputs of bench.py, one problem."""
    rng = np.random.default_rng(seed)
    A = 8.0 * np.eye(D)[None] + 0.1 * rng.standard_normal((N, D, D))
    b = 5.0 * rng.standard_normal((N, D))
    obsX = np.arange(4, N, 8)
    mu0 = rng.standard_normal(D)
    sde = dict(Sig=4.0 * np.eye(D), theta=8.0, Rig=np.eye(D), D=D, N=N, H=np.eye(D), obsX=obsX,
               obsY=3.0 * rng.standard_normal((obsX.size, D)), px0=dict(mu0=mu0, tau0=0.5 * np.eye(D)),
               dt=dt, ode_method="EULER", checkGradf=False, Tw=np.arange(N) * dt)
    Ab = np.concatenate((A.ravel(), b.ravel()))[:, None]
    return Ab, mu0 + 0.1, 0.25 * np.eye(D), sde


def _worker(args):
    kind, D, N, dt, seed, repeat = args
    import warnings
    warnings.simplefilter("ignore")
    Ab, m0, S0, sde = bench_inputs(D, N, dt, seed)
    if kind == "reference":
        np.asscalar = lambda a: np.asarray(a).item()  # noqa: E731
        np.int = int
        np.math = math
        import matplotlib
        matplotlib.use("Agg")
        sys.dont_write_bytecode = True
        sys.path.insert(0, REF)
        from core.varFreeEnergy import varFreeEnergy
        from core.gradL_Ab import gradL_Ab
        from dynamics import sys_lorenz_96_ut as dyn

        def evaluate():
            mParam = dict(m0=m0, S0=S0, fun=dyn.energy_mode, sde_struct=sde)
            F = varFreeEnergy(Ab, mParam)
            g = gradL_Ab(Ab, mParam)
            return F, g
    else:
        sys.path.insert(0, ROOT)
        from oracle import vgpa_oracle as orc

        def evaluate():
            out = orc.evaluate("l96_ut", Ab, m0, S0, sde)
            return out["F"], out["grad"]
    t0 = time.perf_counter()
    for _ in range(repeat):
        F, _ = evaluate()
    return time.perf_counter() - t0, float(F)


def run(kind, procs, D, N, dt, repeat):
    jobs = [(kind, D, N, dt, 1000 + p, repeat) for p in range(procs)]
    t0 = time.perf_counter()
    if procs == 1:
        res = [_worker(jobs[0])]
    else:
        with mp.get_context("fork").Pool(procs) as pool:
            res = pool.map(_worker, jobs)
    wall = time.perf_counter() - t0
    per_proc = [repeat * N / el for el, _ in res]
    return dict(kind=kind, procs=procs, steps=N, repeat=repeat, wall_s=wall,
                rate_sum=float(sum(per_proc)), rate_per_process=float(np.mean(per_proc)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=101, help="time points per problem")
    ap.add_argument("--procs", type=int, default=os.cpu_count())
    ap.add_argument("--repeat", type=int, default=1)
    args = ap.parse_args()
    if not os.path.isdir(REF):
        sys.exit("reference tree not present: this script runs in the build container only")
    D, dt = 40, 0.005
    out = dict(host_cores=os.cpu_count(), config=f"Lorenz-96 UT D={D}, dt={dt}, Euler, N={args.steps}, F + gradient")
    for kind in ("reference", "port"):
        for procs in (1, args.procs):
            r = run(kind, procs, D, args.steps, dt, args.repeat)
            out[f"{kind}_{procs}p"] = r
            print(f"{kind:9s} {procs:2d} process(es): {r['rate_sum']:9.1f} problem*timesteps/s in all, "
                  f"{r['rate_per_process']:8.1f} per process ({r['wall_s']:.1f} s)", flush=True)
    n = args.procs
    out["port_over_reference_1p"] = out["port_1p"]["rate_sum"] / out["reference_1p"]["rate_sum"]
    out["port_over_reference_all"] = out[f"port_{n}p"]["rate_sum"] / out[f"reference_{n}p"]["rate_sum"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
