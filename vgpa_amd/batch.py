"""Spreading free-energy evaluations over several GPUs, one process per GPU.

Two decompositions, both on top of ``torch.distributed`` (backend ``nccl`` is RCCL
on ROCm; ``gloo`` serves the CPU tests):

* **Problem sharding.**  Independent problems (parameter-grid points, restarts,
  data sets) are dealt out in contiguous blocks.  Nothing is exchanged while the
  kernels run; one ``all_gather`` of the P free energies at the end lets every
  rank see the whole grid.

* **Time-window sharding** of one long problem (or one batch of them).  Rank g
  owns the time rows [t0, t1) of A, b and of every per-step array, plus a copy of
  row t1 (the "halo") when it is not the last rank.  The moment ODEs are a
  recurrence in time, so the forward scan is handed from window to window: rank g
  receives (m, S) of row t0, integrates to its halo row t1 and sends that row on.
  The per-step energy kernel, the observation terms and the gradient assembly are
  local.  The backward scan is handed the other way: rank g receives (lam, Psi) of
  its halo row, integrates down to row t0 and sends that row back.  The energies and
  the hyper-parameter gradients are sums of the windows' partial sums: one
  ``all_reduce`` of [Esde, Eobs, integral of the dEsde/dtheta integrand, integral of
  the dEsde/dSigma integrand] per problem (2 + 2 D doubles for the Lorenz systems;
  the trapezoid weights are those of the global segmentation, so the partial sums
  add exactly).  Splitting the batch of problems into micro-batches pipelines the
  hand-offs, so that rank g+1 scans micro-batch c while rank g already scans c+1.

The kernels themselves are reached through an *ops* object (``DeviceWindowOps``
here); the window logic only moves tensors, which is what lets the CPU tests drive
the same code with a different ops object over ``gloo``.

The recurrences being cut are core/fwd_ode.py:87-132 and core/bwd_ode.py:86-127 of
the reference (and their Heun / RK variants); the reference itself is single
process.
"""
import threading
from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import torch


# ---------------------------------------------------------------------------
# index arithmetic
# ---------------------------------------------------------------------------
def shard_bounds(n, rank, world):
    """Contiguous block [lo, hi) of ``n`` items for ``rank`` of ``world``; block
    sizes differ by at most one and the first ranks take the larger ones."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"rank {rank} is not in a world of {world}")
    base, rem = divmod(int(n), world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


@dataclass(frozen=True)
class TimeWindow:
    """Rows [t0, t1) of an N-row time grid, owned by ``rank``."""
    rank: int
    world: int
    N: int
    t0: int
    t1: int

    @property
    def halo(self):
        """1 when a copy of row t1 follows the owned rows."""
        return 0 if self.rank == self.world - 1 else 1

    @property
    def owned(self):
        return self.t1 - self.t0

    @property
    def rows(self):
        return self.owned + self.halo

    def local_slice(self):
        """Global row range held locally, halo included."""
        return slice(self.t0, self.t1 + self.halo)


def time_window(N, rank, world):
    if N < world:
        raise ValueError(f"cannot cut {N} time rows into {world} windows")
    t0, t1 = shard_bounds(N, rank, world)
    return TimeWindow(rank, world, int(N), t0, t1)


# ---------------------------------------------------------------------------
# communicators
# ---------------------------------------------------------------------------
class DistComm:
    """Point-to-point hand-offs and the two collectives over a torch.distributed
    process group.  Each hand-off travels as one flat float64 message."""

    def __init__(self, group=None):
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._dist, self._group = dist, group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self._pending = []

    def _global(self, r):
        return r if self._group is None else self._dist.get_global_rank(self._group, r)

    def send(self, dst, tensors):
        # completed sends release their staging buffers here, so that a long run of
        # micro-batches does not keep every message alive until flush()
        self._pending = [(w, b) for w, b in self._pending if not w.is_completed()]
        buf = torch.cat([t.reshape(-1) for t in tensors]).contiguous()
        self._pending.append((self._dist.isend(buf, self._global(dst), group=self._group), buf))

    def recv(self, src, shapes, device):
        sizes = [int(np.prod(s)) for s in shapes]
        buf = torch.empty(sum(sizes), dtype=torch.float64, device=device)
        self._dist.recv(buf, self._global(src), group=self._group)
        return [c.reshape(s) for c, s in zip(torch.split(buf, sizes), shapes)]

    def flush(self):
        for work, _ in self._pending:
            work.wait()
        self._pending.clear()

    def all_reduce_sum(self, t):
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self._group)
        return t

    def all_gather(self, t):
        out = [torch.empty_like(t) for _ in range(self.world)]
        self._dist.all_gather(out, t.contiguous(), group=self._group)
        return out


class HostStagedComm(DistComm):
    """DistComm for a process group whose backend moves host memory only (gloo): device tensors
    travel through host copies.  For ranks that share one GPU, or a node without a working RCCL:
    the messages are boundary rows and a few numbers per problem, so the staging cost is small."""

    def send(self, dst, tensors):
        self._pending = [(w, b) for w, b in self._pending if not w.is_completed()]
        buf = torch.cat([t.reshape(-1) for t in tensors]).cpu().contiguous()
        self._pending.append((self._dist.isend(buf, self._global(dst), group=self._group), buf))

    def recv(self, src, shapes, device):
        sizes = [int(np.prod(s)) for s in shapes]
        buf = torch.empty(sum(sizes), dtype=torch.float64)
        self._dist.recv(buf, self._global(src), group=self._group)
        buf = buf.to(device)
        return [c.reshape(s) for c, s in zip(torch.split(buf, sizes), shapes)]

    def all_reduce_sum(self, t):
        h = t.detach().cpu()
        self._dist.all_reduce(h, op=self._dist.ReduceOp.SUM, group=self._group)
        t.copy_(h)
        return t

    def all_gather(self, t):
        h = t.detach().cpu().contiguous()
        out = [torch.empty_like(h) for _ in range(self.world)]
        self._dist.all_gather(out, h, group=self._group)
        return [o.to(t.device) for o in out]


class SingleComm:
    """World of one: no messages, collectives are the identity."""
    rank, world = 0, 1

    def send(self, dst, tensors):          # pragma: no cover - never reached with one window
        raise RuntimeError("no peer to send to")

    def recv(self, src, shapes, device):   # pragma: no cover
        raise RuntimeError("no peer to receive from")

    def flush(self):
        pass

    def all_reduce_sum(self, t):
        return t

    def all_gather(self, t):
        return [t]


class ThreadComm:
    """Ranks as threads of one process that share one device: lets a single GPU
    rehearse the window hand-offs.  Build the set with ``ThreadComm.world_of(n)``."""

    def __init__(self, rank, shared):
        self.rank, self._sh = rank, shared
        self.world = shared["world"]

    @classmethod
    def world_of(cls, world):
        import queue
        shared = dict(world=world, q={(s, d): queue.Queue() for s in range(world) for d in range(world)},
                      slots=[None] * world, barrier=threading.Barrier(world))
        return [cls(r, shared) for r in range(world)]

    def send(self, dst, tensors):
        self._sh["q"][(self.rank, dst)].put([t.clone() for t in tensors])

    def recv(self, src, shapes, device):
        got = self._sh["q"][(src, self.rank)].get(timeout=120)
        return [t.reshape(s) for t, s in zip(got, shapes)]

    def flush(self):
        pass

    def _exchange(self, t):
        sh = self._sh
        sh["slots"][self.rank] = t
        sh["barrier"].wait(timeout=120)
        parts = list(sh["slots"])
        sh["barrier"].wait(timeout=120)
        return parts

    def all_reduce_sum(self, t):
        parts = self._exchange(t.clone())
        total = parts[0].clone()
        for x in parts[1:]:
            total = total + x
        t.copy_(total)
        return t

    def all_gather(self, t):
        return [x.clone() for x in self._exchange(t.clone())]


# ---------------------------------------------------------------------------
# problem sharding
# ---------------------------------------------------------------------------
def gather_problem_values(local, P_total, comm):
    """All ranks' per-problem values, in problem order.  ``local`` holds the values
    of this rank's block ``shard_bounds(P_total, rank, world)``; blocks are padded to
    a common length for the one ``all_gather``."""
    if comm.world == 1:
        return local
    longest = -(-int(P_total) // comm.world)
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = comm.all_gather(pad)
    keep = []
    for r, part in enumerate(parts):
        lo, hi = shard_bounds(P_total, r, comm.world)
        keep.append(part[: hi - lo])
    return torch.cat(keep)


def evaluate_problem_shard(evaluate, P_total, comm, gather=True):
    """Run ``evaluate(lo, hi)`` on this rank's block of problems.  ``evaluate``
    returns (F_local, extras); F of the whole grid is gathered unless ``gather`` is
    False (then nothing at all is exchanged)."""
    lo, hi = shard_bounds(P_total, comm.rank, comm.world)
    F_local, extras = evaluate(lo, hi)
    F_all = gather_problem_values(F_local, P_total, comm) if gather else F_local
    return F_all, extras, (lo, hi)


# ---------------------------------------------------------------------------
# time-window sharding
# ---------------------------------------------------------------------------
@dataclass
class WindowResult:
    """One rank's share of a time-sharded evaluation.  Per-step arrays hold the
    owned rows only; F, Esde, Eobs and the hyper-parameter gradients dth, dSig
    (dynamics/sys_lorenz_96_ut.py:250-256) are those of the whole time axis, on every
    rank (dth, dSig are None when the ops object was built without hyper-gradients)."""
    F: torch.Tensor
    m: torch.Tensor
    S: torch.Tensor
    lam: torch.Tensor
    Psi: torch.Tensor
    gLA: Optional[torch.Tensor]
    gLb: Optional[torch.Tensor]
    window: TimeWindow
    Esde: Optional[torch.Tensor] = None
    Eobs: Optional[torch.Tensor] = None
    dth: Optional[torch.Tensor] = None
    dSig: Optional[torch.Tensor] = None


def free_energy_time_sharded(ops_list, comm, want_grad=True) -> List[WindowResult]:
    """Evaluate F (and dL/dA, dL/db) of problems cut into time windows.

    ``ops_list`` holds one ops object per micro-batch of problems, all for this
    rank's window.  An ops object provides

        window                  the TimeWindow
        device                  where its tensors live
        state_shapes()          shapes of (m, S) of one row == shapes of (lam, Psi)
        initial_state()         (m0, S0): used by rank 0
        fwd(m_first, S_first)   -> m, S over the local rows (halo included)
        energy_and_obs(m, S)    -> opaque bundle; its partial energies over owned rows
        bwd(bundle, lam_last, Psi_last) -> lam, Psi over the local rows (None = zero start)
        partial_sums(bundle)    -> (P, K) sums over the owned rows: column 0 Esde, column 1
                                   Eobs, then the time integrals of the dEsde/dtheta and
                                   dEsde/dSigma integrands (absent without hyper-gradients)
        finish(sums)            -> dict(F, Esde, Eobs, dth, dSig) from the all-reduced sums
                                   (adds KL0, which every rank can evaluate, and scales the
                                   hyper-gradient integrals)
        grad(bundle, m, S, lam, Psi) -> gLA, gLb over the local rows
    """
    rank, world = comm.rank, comm.world
    first, last = rank == 0, rank == world - 1
    fwd_out, bundles = [], []
    # Forward hand-off, micro-batch by micro-batch; the local per-step work of a
    # micro-batch runs right after its row has been sent on.
    for ops in ops_list:
        if first:
            m_in, S_in = ops.initial_state()
        else:
            m_in, S_in = comm.recv(rank - 1, ops.state_shapes(), ops.device)
        m, S = ops.fwd(m_in, S_in)
        if not last:
            comm.send(rank + 1, [m[:, -1], S[:, -1]])
        fwd_out.append((m, S))
        bundles.append(ops.energy_and_obs(m, S))
    # Backward hand-off in the same micro-batch order.
    bwd_out = []
    for ops, bundle in zip(ops_list, bundles):
        if last:
            lam_T = Psi_T = None
        else:
            lam_T, Psi_T = comm.recv(rank + 1, ops.state_shapes(), ops.device)
        lam, Psi = ops.bwd(bundle, lam_T, Psi_T)
        if not first:
            comm.send(rank - 1, [lam[:, 0], Psi[:, 0]])
        bwd_out.append((lam, Psi))
    # One collective for all micro-batches.
    parts = [ops.partial_sums(bundle) for ops, bundle in zip(ops_list, bundles)]
    sizes = [int(p.numel()) for p in parts]
    total = comm.all_reduce_sum(torch.cat([p.reshape(-1) for p in parts]))
    sums = [t.reshape(p.shape) for t, p in zip(torch.split(total, sizes), parts)]
    results = []
    for ops, bundle, (m, S), (lam, Psi), sm in zip(ops_list, bundles, fwd_out, bwd_out, sums):
        gLA = gLb = None
        n = ops.window.owned
        if want_grad:
            gLA, gLb = ops.grad(bundle, m, S, lam, Psi)
            gLA, gLb = gLA[:, :n], gLb[:, :n]
        fin = ops.finish(sm)
        results.append(WindowResult(F=fin["F"], m=m[:, :n], S=S[:, :n], lam=lam[:, :n], Psi=Psi[:, :n],
                                    gLA=gLA, gLb=gLb, window=ops.window, Esde=fin["Esde"], Eobs=fin["Eobs"],
                                    dth=fin.get("dth"), dSig=fin.get("dSig")))
    comm.flush()
    return results


def local_window_sde(sde, win):
    """The reference's ``sde_struct`` restricted to a window: N becomes the local
    row count and only the observations on owned rows stay, re-indexed from t0.
    Returns (local_sde, w_local): the trapezoid weights of the *global* segmentation
    (auxiliary/numerical.py:231-284) for the local rows, zero on the halo."""
    from .engine import trapz_weights
    obsX = np.asarray(sde["obsX"]).reshape(-1).astype(np.int64)
    mine = (obsX >= win.t0) & (obsX < win.t1)
    obsY = np.asarray(sde["obsY"], dtype=np.float64)
    if int(sde["D"]) == 1:
        obsY = obsY.reshape(obsX.size, 1) if obsY.size == obsX.size else obsY.reshape(-1, obsX.size, 1)
    local = dict(sde)
    local["N"] = win.rows
    local["obsX"] = obsX[mine] - win.t0
    local["obsY"] = np.compress(mine, obsY, axis=-2)
    w = trapz_weights(win.N, obsX)[win.t0:win.t1]
    return local, np.concatenate([w, np.zeros(win.halo)])


class DeviceWindowOps:
    """The HIP kernels on one rank's rows.  A, b: device tensors over the local rows
    (``window.local_slice()`` of the full arrays), leading axis P."""

    def __init__(self, system, sde, m0, S0, window, A_rows, b_rows, P=1, theta=None, want_edf=False,
                 want_hyper=True):
        from . import engine
        self._e = engine
        self.window = window
        self.want_hyper = bool(want_hyper) and system >= 0
        local_sde, w = local_window_sde(sde, window)
        self.prob = engine.make_problem(system, local_sde, m0, S0, P=P, theta=theta)
        self.prob.w = engine.to_dev(w)
        if A_rows.shape[1] != window.rows:
            raise ValueError(f"A holds {A_rows.shape[1]} rows, the window needs {window.rows}")
        self.A, self.b = A_rows.contiguous(), b_rows.contiguous()
        self.device = self.A.device
        self.want_edf = want_edf or self.prob.D == 1

    def state_shapes(self):
        P, D = self.prob.P, self.prob.D
        return [(P,), (P,)] if D == 1 else [(P, D), (P, D, D)]

    def initial_state(self):
        return self.prob.m0, self.prob.S0

    def fwd(self, m_first, S_first):
        p = self.prob
        return self._e.fwd_scan(p.method, self.A, self.b, m_first.contiguous(), S_first.contiguous(), p.Sigma,
                                p.dt, symmetric=p.sym_fwd)

    def energy_and_obs(self, m, S):
        p, e = self.prob, self._e
        Eobs, dEobs_dm, jumpS = e.obs_terms(p, m, S)
        en = e.energy_terms(p.system, self.A, self.b, m, S, p.SigInv, p.sig_inv_diag, p.theta, p.w, p.dt,
                            want_hyper=self.want_hyper, want_edf=self.want_edf)
        en.update(Eobs=Eobs, dEobs_dm=dEobs_dm, jumpS=jumpS, m=m, S=S)
        return en

    def bwd(self, en, lam_last, Psi_last):
        """The backward scan over the local rows; the gradient of those rows is formed in the
        same library call (one kernel for the Lorenz-96 D = 40 Euler case) and kept for grad()."""
        p, e = self.prob, self._e
        sym = p.sym_jump and e.esde_output_is_symmetric(p.system, p.D)
        if lam_last is not None:
            lam_last, Psi_last = lam_last.contiguous(), Psi_last.contiguous()
        lam, Psi, gLA, gLb = e.bwd_scan_grad(p, self.A, self.b, en["m"], en["S"], en["dEdm"], en["dEdS"],
                                             en["dEobs_dm"], en["Ef"], en["Edf"], jumpS=en["jumpS"], symmetric=sym,
                                             state_symmetric=p.sym_fwd, lam_last=lam_last, Psi_last=Psi_last)
        en["grad"] = (gLA, gLb)
        return lam, Psi

    def partial_sums(self, en):
        P = self.prob.P
        cols = [en["Esde"].reshape(P, 1), en["Eobs"].reshape(P, 1)]
        if self.want_hyper:
            cols.append(en["tr_th"].reshape(P, -1))
            if en["tr_sig"] is not None:
                cols.append(en["tr_sig"].reshape(P, -1))
        return torch.cat(cols, dim=1)

    def finish(self, sums):
        p, e = self.prob, self._e
        Esde, Eobs = sums[:, 0].contiguous(), sums[:, 1].contiguous()
        out = dict(Esde=Esde, Eobs=Eobs, F=e.kl0(p) + Eobs + Esde)      # core/varFreeEnergy.py:111
        if self.want_hyper:
            if p.D == 1:
                tr_th, tr_sig = sums[:, 2], None
            else:
                n_th = (sums.shape[1] - 2) // 2
                tr_th, tr_sig = sums[:, 2:2 + n_th], sums[:, 2 + n_th:]
            out["dth"], out["dSig"] = e.hyper_gradients(p.system, p.D, p.SigInv, p.sig_inv_diag, Esde, tr_th, tr_sig)
        return out

    def grad(self, en, m, S, lam, Psi):
        return en["grad"]
