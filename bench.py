#!/usr/bin/env python3
"""Throughput of the free-energy + gradient evaluation on MI355X.

Workload (BASELINE.json metric): Lorenz-96, D = 40, T = 10, dt = 0.005 (N = 2002
time points), unscented-transform expectations, Euler integrators; a batch of P
independent problems per GPU (parameter-grid points), synthetic inputs resident
in HBM.  One "step" = varFreeEnergy + gradL_Ab for the whole batch:
forward scan, observation terms, per-step energy kernel, time reductions,
backward scan, gradient assembly.  value = P_total * N * steps / time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--problems P] [--no-cpu]

For N > 1 launch under torch.distributed.run (one rank per GPU); problems are
sharded across ranks with no data-path collective (weak scaling: P per GPU fixed).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak


def synth_problem(D, N, dt, P, seed, dev):
    """Synthetic inputs of the benchmark shape, generated on the device."""
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    a0 = 0.5 * 4.0 / 0.25
    A = 0.1 * torch.randn((P, N, D, D), generator=gen, device=dev, dtype=torch.float64)
    A += a0 * torch.eye(D, device=dev, dtype=torch.float64)
    b = 5.0 * torch.randn((P, N, D), generator=gen, device=dev, dtype=torch.float64)
    rng = np.random.default_rng(seed)
    n_obs = 8
    M = int(np.floor((N - 1) * dt * n_obs))
    obsX = np.linspace(0, N, M + 2, dtype=int)[1:-1]
    mu0 = rng.standard_normal(D)
    sde = dict(Sig=4.0 * np.eye(D), theta=8.0, Rig=np.eye(D), D=D, N=N, H=np.eye(D), obsX=obsX,
               obsY=3.0 * rng.standard_normal((M, D)), px0=dict(mu0=mu0, tau0=0.5 * np.eye(D)),
               dt=dt, ode_method="EULER")
    thetas = np.linspace(6.0, 10.0, max(P, 2))[:P].reshape(P, 1)
    return A, b, sde, mu0 + 0.1, 0.25 * np.eye(D), thetas


def cpu_baseline(D, dt, seconds_target=12.0):
    """The NumPy oracle (a port of the reference algorithm, one process, BLAS
    threads as configured on the host) on a bounded sample of the same workload:
    one problem, as many time steps as fit the time budget."""
    from oracle import vgpa_oracle as orc
    rng = np.random.default_rng(1)
    threads = 1
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=1)
    except Exception:   # pragma: no cover
        limiter = None
    n = 64
    best = None
    while True:
        A = 8.0 * np.eye(D)[None] + 0.1 * rng.standard_normal((n, D, D))
        b = 5.0 * rng.standard_normal((n, D))
        obsX = np.arange(4, n, 8)
        mu0 = rng.standard_normal(D)
        sde = dict(Sig=4.0 * np.eye(D), theta=8.0, Rig=np.eye(D), D=D, N=n, H=np.eye(D), obsX=obsX,
                   obsY=3.0 * rng.standard_normal((obsX.size, D)), px0=dict(mu0=mu0, tau0=0.5 * np.eye(D)),
                   dt=dt, ode_method="EULER")
        Ab = np.concatenate((A.ravel(), b.ravel()))
        t0 = time.perf_counter()
        orc.evaluate("l96_ut", Ab, mu0 + 0.1, 0.25 * np.eye(D), sde)
        el = time.perf_counter() - t0
        best = dict(value=n / el, unit="problem*timesteps/s", cores=threads, kind="port",
                    sample=f"1 problem x {n} time steps of the same Lorenz-96 D={D} evaluation, "
                           f"NumPy oracle, 1 BLAS thread, {el:.1f} s")
        if el * 4 > seconds_target or n >= 2002:
            break
        n = min(2002, n * 4)
    if limiter is not None:
        limiter.restore_original_limits() if hasattr(limiter, "restore_original_limits") else None
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--problems", type=int, default=512, help="problems per GPU")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)

    from vgpa_amd import engine, _lib

    D, N, dt, P = 40, 2002, 0.005, args.problems
    A, b, sde, m0, S0, thetas = synth_problem(D, N, dt, P, 20240 + rank, dev)
    prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=P, theta=thetas)

    events = {k: [] for k in ("fwd_scan", "esde", "bwd_scan", "grad_assemble")}
    sym_bwd = prob.sym_jump and engine.esde_output_is_symmetric(prob.system, D)

    def timed_call(name, timed, fn):
        """Run fn between two HIP events recorded on the stream the kernels are launched on
        (torch's current stream is the one handed to the library)."""
        if not timed:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        events[name].append((e0, e1))
        return out

    def step(timed):
        m, S = timed_call("fwd_scan", timed, lambda: engine.fwd_scan(
            prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=prob.sym_fwd))
        Eobs, dEobs_dm, jumpS = engine.obs_terms(prob, m, S)
        en = timed_call("esde", timed, lambda: engine.esde(
            prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False))
        Esde = engine.reduce_energy(en["Esde_t"], prob.w, prob.dt)
        lam, Psi = timed_call("bwd_scan", timed, lambda: engine.bwd_scan(
            prob.method, A, en["dEdm"], en["dEdS"], dEobs_dm, prob.dt,
            obs_flag=prob.obs_flag, jumpS=jumpS, symmetric=sym_bwd))
        F = engine.kl0(prob) + Eobs + Esde
        gLA, gLb = timed_call("grad_assemble", timed, lambda: engine.grad_assemble(
            prob, A, b, m, S, lam, Psi, en["Ef"], en["Edf"]))
        return F, gLA, gLb, en["status"]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step(True)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    F, gLA, gLb, status = out
    ok = bool(torch.isfinite(F).all().item()) and int(status.max().item()) == 0
    kernel_ms = {k: float(np.mean([a.elapsed_time(b_) for a, b_ in v])) for k, v in events.items()}
    esde_ms = kernel_ms["esde"]

    if rank == 0:
        total_units = world * P * N * args.steps
        bytes_per_unit = 8 * (3 * D * D + 4 * D + 1)   # reads A, S, b, m; writes dEsde_dS, dEsde_dm, Ef, Esde_t
        achieved = P * N * bytes_per_unit / (esde_ms * 1e-3) / 1e9
        # Algorithmic bytes per (problem, time step) of the other three kernels: forward scan reads
        # A, b and writes m, S; backward scan reads A, dEsde_dS, dEsde_dm, jumps and writes lam, Psi;
        # gradient assembly reads A, S, Psi and the vectors and writes dL/dA, dL/db.
        per_unit = {"fwd_scan": 8 * (2 * D * D + 2 * D), "esde": bytes_per_unit,
                    "bwd_scan": 8 * (3 * D * D + 3 * D), "grad_assemble": 8 * (4 * D * D + 6 * D)}
        kernels = {k: {"avg_ms": kernel_ms[k], "bytes_per_unit": per_unit[k],
                       "achieved_GBps": P * N * per_unit[k] / (kernel_ms[k] * 1e-3) / 1e9,
                       "frac": P * N * per_unit[k] / (kernel_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                   for k in kernel_ms}
        # HBM bytes per unit measured with rocprofv3 PMC counters (FETCH_SIZE doubled for the
        # 16-byte-per-lane reads, WRITE_SIZE as reported) in separate passes at 256 problems:
        # profiles/r1_pmc_hbm_traffic.txt.  Scaled by the units of this launch.
        measured_traffic_per_unit = 39873
        line = {
            "metric": "free-energy+grad evals/sec (problems x timesteps), L96 D=40",
            "value": total_units / elapsed,
            "unit": "problem*timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Lorenz-96 UT, D=40, T=10, dt=0.005 (N=2002), Euler, "
                                   f"{P} problems per GPU (theta grid), F + dL/dA, dL/db",
                       "problems_per_gpu": P, "N": N, "D": D, "sharding": f"problems x{world}"},
            "valid": ok,
            "roofline": {"kernel": "esde_l96_ut_wave40_kernel<false>", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": measured_traffic_per_unit * P * N, "bytes_per_unit": bytes_per_unit,
                         "avg_ms": esde_ms},
            "kernels": kernels,
        }
        if not args.no_cpu and world == 1:
            line["cpu_baseline"] = cpu_baseline(D, dt)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
