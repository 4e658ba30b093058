#!/usr/bin/env python3
"""Throughput of the free-energy + gradient evaluation on MI355X.

Workload (BASELINE.json metric): Lorenz-96, D = 40, T = 10, dt = 0.005 (N = 2002
time points), unscented-transform expectations, Euler integrators; a batch of P
independent problems (parameter-grid points), synthetic inputs resident in HBM.
One "step" = varFreeEnergy + gradL_Ab for the whole batch: forward scan,
observation terms, per-step energy kernel, time reductions, backward scan fused
with the gradient assembly.  value = P_total * N * steps / time.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--mode problems|windows]
                  [--problems P] [--micro M] [--backend nccl|gloo] [--no-cpu] [--no-parity]

--mode problems (default): P problems PER GPU, sharded by problem across the ranks
  (vgpa_amd.batch.evaluate_problem_shard): no data-path collective, one all_gather of
  the P_total free energies per step.  Weak scaling.
--mode windows: P problems IN ALL, every one cut into --gpus contiguous time windows
  (vgpa_amd.batch.free_energy_time_sharded): scan boundary rows handed between
  neighbouring ranks, one all_reduce of the energy and hyper-gradient partial sums per
  step, M micro-batches of problems to pipeline the hand-offs.  Strong scaling.

For N > 1 either launch it under torch.distributed.run (one rank per GPU, RCCL) or just run
`python bench.py --gpus N`: outside a launcher it starts its own N ranks (launch_ranks).
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak
FP64_PEAK_TFLOPS = 78.6  # MI355X FP64, vector and matrix alike
ESDE_KERNEL = "esde_l96_ut_wave40_kernel<false, false, false>"
KERNEL_NAMES = {"fwd_scan": "fwd_euler_sym_mfma40_kernel<false>", "esde": ESDE_KERNEL,
                "bwd_scan_grad": "bwd_grad_euler_sym_mfma40_kernel<false, false, false>"}
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "pmc_hbm_traffic.txt")
STREAM_FILE = os.path.join(ROOT, "profiles", "r3_hbm_stream_probe.txt")
REFERENCE_FILE = os.path.join(ROOT, "profiles", "r3_time_reference.txt")
# read / write streams of each kernel, as tools/diag/hbm_stream_probe names the mixes
STREAM_MIX = {"fwd_scan": "1R1W", "esde": "2R1W", "bwd_scan_grad": "3R2W"}


def synth_problem(D, N, dt, P, seed, dev):
    """Synthetic inputs of the benchmark shape, generated on the device."""
    import torch
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


# ---------------------------------------------------------------------------
# CPU baseline: the NumPy oracle, one process per core of this job's CPU share
# ---------------------------------------------------------------------------
def _cpu_worker(args):
    """One problem of the benchmark shape with n time steps, `repeat` evaluations."""
    D, dt, n, seed, repeat = args
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(limits=1)
    except Exception:   # pragma: no cover
        pass
    from oracle import vgpa_oracle as orc
    rng = np.random.default_rng(seed)
    A = 8.0 * np.eye(D)[None] + 0.1 * rng.standard_normal((n, D, D))
    b = 5.0 * rng.standard_normal((n, D))
    obsX = np.arange(4, n, 8)
    mu0 = rng.standard_normal(D)
    sde = dict(Sig=4.0 * np.eye(D), theta=8.0, Rig=np.eye(D), D=D, N=n, H=np.eye(D), obsX=obsX,
               obsY=3.0 * rng.standard_normal((obsX.size, D)), px0=dict(mu0=mu0, tau0=0.5 * np.eye(D)),
               dt=dt, ode_method="EULER")
    Ab = np.concatenate((A.ravel(), b.ravel()))
    t0 = time.perf_counter()
    for _ in range(repeat):
        orc.evaluate("l96_ut", Ab, mu0 + 0.1, 0.25 * np.eye(D), sde)
    return time.perf_counter() - t0


def cpu_share():
    """Cores this job may use: the thread budget the environment sets for the box
    (os.cpu_count() shows the whole machine there), else what the scheduler allows."""
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:   # pragma: no cover
        avail = os.cpu_count() or 1
    budget = os.environ.get("VGPA_BENCH_CORES") or os.environ.get("OMP_NUM_THREADS")
    n = int(budget) if budget and budget.isdigit() and int(budget) > 0 else avail
    return max(1, min(n, avail))


def reference_over_port():
    """Rate of the reference itself over the rate of the NumPy oracle, all cores busy, as measured where the
    reference can be imported (tools/time_reference.py at N = 2002; the tracked output is read here).  The
    oracle vectorises the reference's per-step Python loops, so it is the faster of the two."""
    try:
        with open(REFERENCE_FILE) as f:
            rec = json.loads([ln for ln in f if ln.startswith("{")][-1])
        return 1.0 / float(rec["port_over_reference_all"])
    except (OSError, ValueError, KeyError, IndexError):
        return None


def cpu_baseline(D, dt, seconds_target=12.0):
    """The NumPy oracle (a port of the reference algorithm) on a bounded sample of the same
    workload: one process per core, each with one BLAS thread and its own problem of n time
    steps, n sized so that the sample takes about `seconds_target` seconds.  Started before
    anything touches the GPU (worker processes are spawned)."""
    import multiprocessing as mp
    os.environ["OPENBLAS_NUM_THREADS"] = "1"        # inherited by the workers
    cores = cpu_share()
    t_cal = _cpu_worker((D, dt, 64, 1, 1))          # calibration in this process
    n = int(min(2002, max(64, 64 * seconds_target / max(t_cal, 1e-3))))
    repeat = max(1, int(seconds_target / max(t_cal * n / 64.0, 1e-3)))
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(cores) as pool:
        times = pool.map(_cpu_worker, [(D, dt, n, 100 + c, repeat) for c in range(cores)])
    wall = time.perf_counter() - t0
    rate = float(sum(repeat * n / t for t in times))
    rop = reference_over_port()
    return dict(value=rate, unit="problem*timesteps/s", cores=cores, kind="port",
                per_core=rate / cores,
                # what the reference itself would do on these cores (ratio measured in the build container)
                reference_over_port=rop, reference_equivalent=None if rop is None else rate * rop,
                reference_over_port_source=os.path.relpath(REFERENCE_FILE, ROOT),
                sample=f"{cores} processes x 1 problem x {n} time steps x {repeat} evaluation(s) of the same "
                       f"Lorenz-96 D={D} free energy + gradient, NumPy oracle, 1 BLAS thread each, "
                       f"{max(times):.1f} s of work per process ({wall:.1f} s wall incl. start-up)")


def stream_rate(mix):
    """GB/s of the best plain streaming kernel with this mix of read and write streams on the same
    part (tracked output of tools/diag/hbm_stream_probe), or None."""
    best = None
    try:
        with open(STREAM_FILE) as f:
            for ln in f:
                w = ln.split()
                if len(w) >= 5 and w[0] == mix and w[-1] == "GB/s":
                    best = max(best or 0.0, float(w[-2]))
    except (OSError, ValueError):
        return None
    return best


def measured_traffic_per_unit(kernel):
    """HBM bytes per (problem, time step) of `kernel` from the tracked rocprofv3 counter
    summary (separate FETCH_SIZE / WRITE_SIZE passes, FETCH_SIZE doubled for 16-byte-per-lane
    reads): the last line of profiles/pmc_hbm_traffic.txt whose first field is the kernel
    name.  None when the file has no line for this kernel."""
    try:
        lines = open(TRAFFIC_FILE).read().splitlines()
    except OSError:
        return None
    found = None
    for ln in lines:
        if ln.startswith("#") or "," not in ln:
            continue
        m = re.match(r"^(.*),([0-9.e+]+),([0-9.e+]+),([0-9.e+]+),([0-9.e+]+),([0-9.e+]+),([0-9.e+]+),([0-9.e+]+)$", ln)
        if m and m.group(1).strip() == kernel:
            found = float(m.group(7))
    return found


def parity_on_fixture(dev, copies=192):
    """Max-norm relative error of F and of the gradient column against the vectors recorded from
    the reference for Lorenz-96 D = 40 (tests/golden/sys_l96_ut_d40.npz, Euler), evaluated by the
    kernels the timed region runs: the recorded problem is replicated `copies` times so that
    the library takes the fused backward scan + gradient kernel.  None without the fixture."""
    import torch
    from vgpa_amd import engine, _lib
    path = os.path.join(ROOT, "tests", "golden", "sys_l96_ut_d40.npz")
    try:
        with np.load(path, allow_pickle=False) as z:
            g = {k: z[k] for k in z.files}
    except OSError:
        return None
    D, N = int(g["D"]), int(g["N"])
    sde = dict(Sig=np.array(g["Sig"]), theta=float(g["theta"]) if g["theta"].ndim == 0 else np.array(g["theta"]),
               Rig=np.array(g["Rig"]), D=D, N=N, H=np.array(g["H"]), obsX=np.array(g["obsX"]),
               obsY=np.array(g["obsY"]), px0=dict(mu0=np.array(g["mu0"]), tau0=np.array(g["tau0"])),
               dt=float(g["dt"]), ode_method="EULER")
    Ab = np.asarray(g["Ab"], dtype=np.float64).reshape(-1)
    A1, b1 = Ab[: N * D * D].reshape(N, D, D), Ab[N * D * D:].reshape(N, D)
    prob = engine.make_problem(_lib.SYS_L96_UT, sde, np.array(g["m0"]), np.array(g["S0"]), P=copies)
    A = torch.from_numpy(A1).to(dev).expand(copies, N, D, D).contiguous()
    b = torch.from_numpy(b1).to(dev).expand(copies, N, D).contiguous()
    st, gLA, gLb = engine.free_energy_and_gradient(prob, A, b)
    F_ref, g_ref = float(g["EULER.F"]), np.asarray(g["EULER.grad"], dtype=np.float64).reshape(-1)
    F_err = float((st.F - F_ref).abs().max().item()) / abs(F_ref)
    got = torch.cat((gLA.reshape(copies, -1), gLb.reshape(copies, -1)), dim=1).cpu().numpy()
    g_err = float(np.max(np.abs(got - g_ref[None, :])) / np.max(np.abs(g_ref)))
    sflags = (_lib.SCAN_SYMMETRIC if prob.sym_jump and engine.esde_output_is_symmetric(prob.system, D) else 0) | \
             (_lib.SCAN_STATE_SYMMETRIC if prob.sym_fwd else 0)
    fused = bool(_lib.lib.vgpa_bwd_scan_grad_is_fused(prob.method, copies, D, sflags,
                                                      _lib.GRAD_SIGINV_DIAG if prob.sig_is_diag else 0, 0, 0,
                                                      prob.system))
    return {"fixture": os.path.relpath(path, ROOT), "recorded_from": "the reference's varFreeEnergy + gradL_Ab",
            "problems": copies, "N": N, "D": D, "fused_backward_kernel": fused,
            "F_rel_err": F_err, "grad_rel_err": g_err, "tol": 1e-9, "ok": bool(F_err <= 1e-9 and g_err <= 1e-9)}


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside a launcher: time the CPU baseline here, then start the N
    ranks as children of this process under torch.distributed.run (rendezvous on 127.0.0.1, a
    free port), pass every argument through, print rank 0's JSON line with the CPU baseline
    merged in and return the launcher's exit code.  This process never touches the GPU."""
    import socket
    import subprocess
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cpu = None if args.no_cpu else cpu_baseline(40, 0.005)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    if "--no-cpu" not in argv:
        cmd.append("--no-cpu")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        s_ = ln.strip()
        if s_.startswith("{") and '"metric"' in s_:
            try:
                line = json.loads(s_)
                continue
            except ValueError:
                pass
        sys.stderr.write(ln)       # anything else the ranks print goes to stderr: stdout carries the one JSON line
    rc = proc.wait()
    if line is not None:
        if cpu is not None:
            line["cpu_baseline"] = cpu
        line["launcher"] = "bench.py started its own ranks (torch.distributed.run)"
        print(json.dumps(line), flush=True)
    elif rc == 0:
        rc = 1
    return rc


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("problems", "windows"), default="problems")
    ap.add_argument("--packed-intermediate", action="store_true",
                    help="pass dEsde/dS, the array between the energy kernel and the backward scan (not one the reference "
                         "publishes), through memory in the packed symmetric layout; every published array stays dense")
    ap.add_argument("--no-optimiser-path", action="store_true",
                    help="skip the extra timing of the packed, gradient-only evaluation reported as `optimiser_path`")
    ap.add_argument("--problems", type=int, default=512,
                    help="problems per GPU (mode problems) or in all (mode windows)")
    ap.add_argument("--micro", type=int, default=None,
                    help="micro-batches of problems (mode windows); default 4, or 1 when there is one window")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend for N > 1: nccl (RCCL, one GPU per rank) or gloo with the messages "
                         "staged through the host (ranks may then share a GPU: a rehearsal on a one-GPU box)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-parity", action="store_true",
                    help="skip the fixture check after the timed region (profiling runs: its small launches "
                         "of the same kernels would enter the per-kernel averages)")
    args = ap.parse_args(argv)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))      # not under a launcher: start the ranks ourselves
    if args.gpus > 1 and world != args.gpus:
        sys.exit(f"--gpus {args.gpus} does not match the launcher's WORLD_SIZE={world}")
    D, N, dt, P = 40, 2002, 0.005, args.problems
    # the host driver of this pool shares device memory between processes through dmabuf only (RCCL's
    # hipIpcGetMemHandle fails otherwise); normally exported already, set before anything initialises HIP
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    cpu = None
    if not args.no_cpu and world == 1 and rank == 0:
        cpu = cpu_baseline(D, dt)          # before the GPU is initialised

    import torch
    if args.backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # VGPA_BENCH_FORCE_DIST=1 runs a single rank through the RCCL communicator too (a rehearsal of the
    # collectives on a one-GPU box)
    force_dist = world == 1 and os.environ.get("VGPA_BENCH_FORCE_DIST") == "1"
    if world > 1 or force_dist:
        import torch.distributed as dist
        if args.backend == "gloo":
            dist.init_process_group("gloo")
        elif force_dist and "MASTER_ADDR" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        else:
            dist.init_process_group("nccl", device_id=dev)

    from vgpa_amd import batch, engine, _lib
    if dist is None:
        comm = batch.SingleComm()
    else:
        comm = batch.HostStagedComm() if args.backend == "gloo" else batch.DistComm()

    events = {k: [] for k in ("fwd_scan", "esde", "bwd_scan_grad")}

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

    if args.mode == "problems":
        P_total = world * P
        lo, hi = batch.shard_bounds(P_total, rank, world)
        A, b, sde, m0, S0, thetas = synth_problem(D, N, dt, hi - lo, 20240 + rank, dev)
        prob = engine.make_problem(_lib.SYS_L96_UT, sde, m0, S0, P=hi - lo, theta=thetas)
        sym_bwd = prob.sym_jump and engine.esde_output_is_symmetric(prob.system, D)
        units_per_launch = (hi - lo) * N
        packed_mid = engine.packed_supported(prob) and args.packed_intermediate

        def evaluate_block(timed):
            m, S = timed_call("fwd_scan", timed, lambda: engine.fwd_scan(
                prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=prob.sym_fwd))
            Eobs, dEobs_dm, jumpS = engine.obs_terms(prob, m, S)
            # dEsde/dS, an intermediate between the two kernels that the reference does not publish, goes through
            # memory in the packed symmetric layout where the kernels take it; m, S, lam, Psi and the gradient
            # are the dense arrays of the contract
            if packed_mid:
                en = timed_call("esde", timed, lambda: engine.esde_l96_packed(
                    A, b, m, S, prob.sig_inv_diag, prob.theta, want_hyper=False, packed_S=False))
            else:
                en = timed_call("esde", timed, lambda: engine.esde(
                    prob.system, A, b, m, S, prob.sig_inv_diag, prob.theta, False, False))
            Esde = engine.reduce_energy(en["Esde_t"], prob.w, prob.dt)
            # backward scan and gradient assembly: one kernel for this configuration
            lam, Psi, gLA, gLb = timed_call("bwd_scan_grad", timed, lambda: engine.bwd_scan_grad(
                prob, A, b, m, S, en["dEdm"], en["dEdS"], dEobs_dm, en["Ef"], en["Edf"], jumpS=jumpS,
                symmetric=sym_bwd, state_symmetric=prob.sym_fwd, packed="dEdS" if packed_mid else False))
            F = engine.kl0(prob) + Eobs + Esde
            return F, (gLA, gLb, en["status"])

        def step(timed):
            # gradients stay on the owning GPU; the free energies of the whole grid are gathered
            F_all, extras, _ = batch.evaluate_problem_shard(lambda lo_, hi_: evaluate_block(timed), P_total, comm)
            return F_all, extras[2]

        def optimiser_step():
            """What an optimiser consumes -- F, dL/dA, dL/db -- with the symmetric matrices of every row kept
            packed (1152 doubles instead of 1600) and Psi(t) not stored beyond t = 0.  Reported beside the
            headline, never as it: the headline writes every array of the reference's contract, dense."""
            st, gLA, gLb = engine.free_energy_and_gradient(prob, A, b, costate=False, packed=True)
            return st.F, st.status
    else:
        P_total = P
        win = batch.time_window(N, rank, world)
        # micro-batches overlap the hand-offs between windows; with one window there is nothing to overlap
        micro = max(1, min(args.micro if args.micro else (4 if world > 1 else 1), P))
        sde = None
        ops = []
        for c in range(micro):
            plo, phi = batch.shard_bounds(P, c, micro)
            # every rank draws the same problems (same seed) and keeps its rows
            A, b, sde, m0, S0, thetas = synth_problem(D, N, dt, phi - plo, 20240 + c, dev)
            sl = win.local_slice()
            ops.append(batch.DeviceWindowOps(_lib.SYS_L96_UT, sde, m0, S0, win, A[:, sl].contiguous(),
                                             b[:, sl].contiguous(), P=phi - plo, theta=thetas, want_hyper=True))
            del A, b
        units_per_launch = None   # several launches per kernel and step: per-kernel timing is not split out

        def step(timed):
            res = batch.free_energy_time_sharded(ops, comm)
            return torch.cat([r.F for r in res]), None

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            # RCCL picks the device of a barrier from the rank unless told: name it
            dist.barrier(device_ids=[local_rank]) if args.backend == "nccl" else dist.barrier()
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
        t = torch.tensor([elapsed], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    F, status = out
    ok = bool(torch.isfinite(F).all().item()) and int(F.numel()) == P_total
    if status is not None:
        ok = ok and int(status.max().item()) == 0

    fast = None
    if args.mode == "problems" and not args.no_optimiser_path and engine.packed_supported(prob):
        Ff, sf = optimiser_step()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            Ff, sf = optimiser_step()
        barrier()
        fast_elapsed = time.perf_counter() - t1
        lo_, hi_ = batch.shard_bounds(P_total, rank, world)
        same = bool(torch.equal(Ff, F[lo_:hi_].to(Ff.device))) and int(sf.max().item()) == 0
        fast = {"what": "F, dL/dA, dL/db only: symmetric matrices packed (1152 of 1600 doubles a row), Psi(t) kept for "
                        "t = 0 alone; this rank's rate, outside the timed region of `value`",
                "ms_per_step": 1e3 * fast_elapsed / args.steps,
                "units_per_s_per_gpu": (hi_ - lo_) * N * args.steps / fast_elapsed,
                "F_bitwise_equal_to_headline_path": same}

    if rank == 0:
        total_units = P_total * N * args.steps
        value = total_units / elapsed
        line = {
            "metric": "free-energy+grad evals/sec (problems x timesteps), L96 D=40",
            "value": value,
            "unit": "problem*timesteps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak" if args.mode == "problems" else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "valid": ok,
        }
        # whole evaluation against the contract minimum: inputs A, b once, every contract output
        # (m, S, lam, Psi, dL/dA, dL/db ...) once = 8 (5 D^2 + 5 D) bytes per (problem, time step)
        e2e_bytes = 8 * (5 * D * D + 5 * D)
        e2e = value / world * e2e_bytes / 1e9
        line["end_to_end"] = {"bytes_per_unit": e2e_bytes, "achieved_GBps_per_gpu": e2e, "frac": e2e / HBM_PEAK_GBS}
        if fast is not None:
            # against the same contract bytes as `end_to_end`, so that the two fractions compare
            fast["end_to_end_frac"] = fast["units_per_s_per_gpu"] * e2e_bytes / 1e9 / HBM_PEAK_GBS
            line["optimiser_path"] = fast
        if args.mode == "problems":
            line["config"] = {"workload": "Lorenz-96 UT, D=40, T=10, dt=0.005 (N=2002), Euler, "
                                          f"{P} problems per GPU (theta grid), F + dL/dA, dL/db",
                              "problems_per_gpu": P, "N": N, "D": D, "sharding": f"problems x{world}",
                              "collective": "all_gather of F" if world > 1 else "none (1 rank)",
                              "backend": args.backend if dist is not None else None}
            kernel_ms = {k: float(np.mean([a.elapsed_time(b_) for a, b_ in v])) for k, v in events.items()}
            esde_ms = kernel_ms["esde"]
            bytes_per_unit = 8 * (3 * D * D + 4 * D + 1)   # reads A, S, b, m; writes dEsde_dS, dEsde_dm, Ef, Esde_t
            # Algorithmic bytes per (problem, time step) of the other kernels: forward scan reads A, b
            # and writes m, S; the fused backward scan + gradient kernel reads A, dEsde_dS, S and five
            # vectors (dEsde_dm, jumps, m, <f>, b) and writes Psi, dL/dA, lam, dL/db.
            per_unit = {"fwd_scan": 8 * (2 * D * D + 2 * D), "esde": bytes_per_unit,
                        "bwd_scan_grad": 8 * (5 * D * D + 7 * D)}
            names = dict(KERNEL_NAMES)
            if packed_mid:
                # --packed-intermediate: dEsde/dS moves as the 1088 kept entries of a packed row instead of D^2
                per_unit["esde"] -= 8 * (D * D - 1088)
                per_unit["bwd_scan_grad"] -= 8 * (D * D - 1088)
                names["esde"] = "esde_l96_ut_wave40_kernel<false, false, true>"
                names["bwd_scan_grad"] = "bwd_grad_euler_sym_mfma40_kernel<false, true, false>"
            kernels = {k: {"avg_ms": kernel_ms[k], "bytes_per_unit": per_unit[k],
                           "achieved_GBps": units_per_launch * per_unit[k] / (kernel_ms[k] * 1e-3) / 1e9,
                           "frac": units_per_launch * per_unit[k] / (kernel_ms[k] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                       for k in kernel_ms}
            def roof(k):
                """The roofline object of kernel k: algorithmic bytes per launch over its measured
                average duration, and the HBM traffic the counters of the tracked profile give."""
                t = measured_traffic_per_unit(names[k])
                sr = stream_rate(STREAM_MIX[k])
                return {"kernel": names[k], "bound": "hbm", "achieved": kernels[k]["achieved_GBps"],
                        "stream_same_mix": None if sr is None else
                        {"mix": STREAM_MIX[k], "GBps": sr, "frac_of_it": kernels[k]["achieved_GBps"] / sr,
                         "source": os.path.relpath(STREAM_FILE, ROOT)},
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": kernels[k]["frac"],
                        "traffic": None if t is None else t * units_per_launch,
                        "traffic_source": os.path.relpath(TRAFFIC_FILE, ROOT) if t is not None else None,
                        "bytes_per_unit": per_unit[k], "avg_ms": kernels[k]["avg_ms"],
                        "share_of_step": kernels[k]["avg_ms"] / sum(v["avg_ms"] for v in kernels.values())}

            # `roofline`: the kernel with the largest share of the step (the fused backward scan +
            # gradient since the fusion); `roofline_energy_kernel`: the expected-energy kernel, the
            # second largest, also against the FP64 roof: about 170 kflop of arithmetic per unit
            # (factorisation D^3/3, A L' D^3, inverse D^3/3, W' diag(g) W D^3/3, 20 flop for each of the
            # D^2 sigma-point residual entries) against the 78.6 TFLOP/s FP64 peak.
            dom = max(kernels, key=lambda k: kernels[k]["avg_ms"])
            line["roofline"] = roof(dom)
            line["roofline_energy_kernel"] = roof("esde")
            flops_per_unit = 170e3
            tf = units_per_launch * flops_per_unit / (esde_ms * 1e-3) / 1e12
            line["roofline_energy_kernel"]["fp64"] = {"flops_per_unit": flops_per_unit, "achieved_TFLOPs": tf,
                                                      "peak": FP64_PEAK_TFLOPS, "frac": tf / FP64_PEAK_TFLOPS}
            line["kernels"] = kernels
        else:
            line["config"] = {"workload": "Lorenz-96 UT, D=40, T=10, dt=0.005 (N=2002), Euler, "
                                          f"a batch of {P} problem(s) (--problems; 1 = the single long problem), "
                                          f"every one cut into {world} time windows, F + dL/dA, dL/db "
                                          "+ dEsde/dtheta, dEsde/dSigma",
                              "problems": P, "micro_batches": len(ops), "N": N, "D": D,
                              "sharding": f"time windows x{world}",
                              "collective": "all_reduce of (2 + 2 D) partial sums per problem; "
                                            "point-to-point scan boundary rows",
                              "backend": args.backend if dist is not None else None}
        # the same kernels against the vectors recorded from the reference (outside the timed region)
        line["parity"] = None if args.no_parity else parity_on_fixture(dev)
        line["valid"] = bool(ok and (line["parity"] is None or line["parity"]["ok"]))
        if cpu is not None:
            line["cpu_baseline"] = cpu
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
