"""Optimisers that drive the free-energy path: the reference's call protocol
(auxiliary/optimize.py) re-implemented, plus a batched scaled-conjugate-gradient
that keeps a whole grid of problems on the device.

``optim_SCG`` / ``optim_CG`` / ``optim_SA`` take ``(f, x0, gf, options, mParam)`` and
return ``(x, fx, mParam, stat)`` exactly like the reference: same evaluation order of
``f`` and ``gf`` (they communicate through ``mParam``), same counters, same
termination rules, same ``stat`` arrays -- so that a run through the HIP path follows
the trajectory of a run through the reference (tests replay a recorded one).

``scg_batched`` runs the same SCG recursion for P independent problems at once:
the iterate is a (P, n) device tensor, every scalar of the recursion (mu, kappa,
theta, delta, alpha, beta, the comparison ratio, the success flags) is a (P,) tensor,
and one batched free-energy / gradient evaluation serves all problems.
"""
import math
from copy import deepcopy

import numpy as np

__all__ = ["optim_SCG", "optim_CG", "optim_SA", "backtrack", "scg_batched"]

_EPS = np.finfo(float).eps


def _opt(options, key, default):
    return options[key] if key in options else default


def _dot(a, b):
    """a' b of two columns as a float."""
    return float(np.dot(a.ravel(), b.ravel()))


# ---------------------------------------------------------------------------
# scaled conjugate gradient (Moller 1993, in Nabney's NETLAB form)
# ---------------------------------------------------------------------------
def optim_SCG(f, x0, gf, options, mParam):
    """auxiliary/optimize.py:9-247.

    options: 'nit' (150), 'xtol' (1e-6), 'ftol' (1e-8), 'disp' (False).
    stat: 'MaxIt', 'Fx' (nit,1), 'f_eval', 'g_eval', 'beta' (nit,1), 'gradx' (nit,1).
    """
    nit = _opt(options, "nit", 150)
    xtol = _opt(options, "xtol", 1.0e-6)
    ftol = _opt(options, "ftol", 1.0e-8)
    disp = _opt(options, "disp", False)
    print(" >> SCG optimization")

    n = x0.size
    stat = dict(MaxIt=nit, Fx=np.zeros((nit, 1)), f_eval=0, g_eval=0,
                beta=np.zeros((nit, 1)), gradx=np.zeros((nit, 1)))
    x = deepcopy(x0).reshape(n, 1)

    sigma0 = 1.0e-3
    beta, beta_lo, beta_hi = 1.0, 1.0e-15, 1.0e+100

    f_cur = f(x, mParam)                       # optimize.py:99-100
    g_new = gf(x, mParam)
    stat["f_eval"] += 1
    stat["g_eval"] += 1
    f_old, g_old = f_cur, g_new
    d = -g_new
    accepted = True                            # forces the curvature probe at j = 0
    n_accepted = 0
    mu = kappa = theta = 0.0

    for j in range(nit):
        if accepted:
            # slope and curvature along d (optimize.py:134-159)
            mu = _dot(d, g_new)
            if mu >= 0.0:
                d = -g_new
                mu = _dot(d, g_new)
            kappa = _dot(d, d)
            if kappa < _EPS:
                stat["MaxIt"] = j
                return x, f_cur, mParam, stat
            sigma = sigma0 / math.sqrt(kappa)
            g_probe = gf(x + sigma * d, mParam, True)
            stat["f_eval"] += 1
            stat["g_eval"] += 1
            theta = _dot(d, g_probe - g_new) / sigma

        # raise the effective curvature until it is positive (optimize.py:162-167)
        delta = theta + beta * kappa
        if delta <= 0:
            delta = beta * kappa
            beta = beta - theta / kappa
        alpha = -(mu / delta)

        x_try = x + alpha * d
        f_try = f(x_try, mParam)
        stat["f_eval"] += 1

        ratio = 2.0 * (f_try - f_old) / (alpha * mu)      # optimize.py:175
        if ratio >= 0:
            accepted = True
            n_accepted += 1
            x, f_cur, g_shown = x_try, f_try, g_new
        else:
            accepted = False
            f_cur, g_shown = f_old, g_old

        total = math.fsum(np.abs(g_shown).ravel())
        stat["Fx"][j] = f_cur
        stat["beta"][j] = beta
        stat["gradx"][j] = total
        if disp:
            print(" {0}:\tfx={1:.3f}\tsum(gx)={2:.3f}".format(j, f_cur, total))

        if accepted:
            if np.abs(alpha * d).max() <= xtol and abs(f_try - f_old) <= ftol:
                stat["MaxIt"] = j
                return x, f_try, mParam, stat
            f_old, g_old = f_try, g_new
            f_cur = f(x, mParam)                           # optimize.py:210-211
            g_new = gf(x, mParam)
            stat["f_eval"] += 1
            stat["g_eval"] += 1
            if _dot(g_new, g_new) == 0.0:
                stat["MaxIt"] = j
                return x, f_cur, mParam, stat

        if ratio < 0.25:
            beta = min(4.0 * beta, beta_hi)
        if ratio > 0.75:
            beta = max(0.5 * beta, beta_lo)

        # Polak-Ribiere direction, restarted after n accepted steps (optimize.py:232-238)
        if n_accepted == n:
            d = -g_new
            n_accepted = 0
        elif accepted:
            gamma = max(_dot(g_new, g_old - g_new) / mu, 0.0)
            d = gamma * d - g_new

    print(" Maximum number of iterations has been reached.")
    return x, f_old, mParam, stat


# ---------------------------------------------------------------------------
# conjugate gradient with back-tracking
# ---------------------------------------------------------------------------
def backtrack(f, x0, f0, df0, eta0, r, *args):
    """auxiliary/optimize.py:429-482: shrink the step eta0 by the factor r (at most 15
    times) until f(x0 + eta df0) is finite and not above f0.  Returns (eta, number of
    evaluations)."""
    eta, tries = eta0, 0
    fx = f(x0 + eta0 * df0, *args)
    while tries < 15 and (not np.isfinite(fx) or fx > f0):
        eta *= r
        fx = f(x0 + eta * df0, *args)
        tries += 1
    return max(eta, 0.0), tries + 1


def optim_CG(f, x0, gf, options, mParam):
    """auxiliary/optimize.py:250-426.  stat has 'eta' in place of 'beta'."""
    nit = _opt(options, "nit", 500)
    xtol = _opt(options, "xtol", 1.0e-6)
    ftol = _opt(options, "ftol", 1.0e-8)
    disp = _opt(options, "disp", False)
    print(" >> CG optimization (with backtracking)")

    n = x0.size
    stat = dict(MaxIt=nit, Fx=np.zeros((nit, 1)), f_eval=0, g_eval=0,
                eta=np.zeros((nit, 1)), gradx=np.zeros((nit, 1)))
    x = deepcopy(x0).reshape(n, 1)
    eta0, shrink = 0.15, 0.5

    f_new = f(x, mParam)
    g_new = gf(x, mParam)
    stat["f_eval"] += 1
    stat["g_eval"] += 1
    d = -g_new
    stat["Fx"][0] = f_new
    stat["eta"][0] = eta0
    stat["gradx"][0] = np.sum(np.abs(g_new))
    f_old = f_new

    for j in range(1, nit):
        x_old, f_old, g_old = x, f_new, g_new
        mu = _dot(g_old, g_old)
        if mu == 0.0:
            stat["MaxIt"] = j
            return x, f_old, mParam, stat
        if _dot(g_new, d) > 0.0:
            d = -d
        line = d / np.linalg.norm(d)
        # the line search works on a copy so that mParam keeps the state of x_old
        eta, cnt = backtrack(f, x_old, f_old, line, eta0, shrink, deepcopy(mParam))
        stat["f_eval"] += cnt
        if eta == 0.0:
            stat["MaxIt"] = j
            return x_old, f_old, mParam, stat
        x = x_old + eta * line
        f_new = f(x, mParam)
        g_new = gf(x, mParam)
        stat["f_eval"] += 1
        stat["g_eval"] += 1
        if np.abs(x - x_old).max() <= xtol and abs(f_new - f_old) <= ftol:
            stat["MaxIt"] = j
            return x, f_new, mParam, stat
        gamma = max(_dot(g_new, g_old - g_new) / mu, 0.0)
        d = gamma * d - g_new
        total = math.fsum(np.abs(g_new).ravel())
        stat["Fx"][j] = f_new
        stat["eta"][j] = eta
        stat["gradx"][j] = total
        if disp:
            print(" {0}:\tfx={1:.3f}\tsum(gx)={2:.3f}".format(j, f_new, total))

    print(" Maximum number of iterations has been reached.")
    return x, f_old, mParam, stat


# ---------------------------------------------------------------------------
# simulated annealing with gradient drift
# ---------------------------------------------------------------------------
def optim_SA(f, x0, gf, options, mParam):
    """auxiliary/optimize.py:485-674.  Draws from numpy's global generator in the same
    order as the reference (one ``randn(n,1)`` per proposal, one ``rand()`` only when
    the proposal is uphill)."""
    nrun = _opt(options, "nrun", 1)
    xtol = _opt(options, "xtol", 1.0e-6)
    ftol = _opt(options, "ftol", 1.0e-8)
    disp = _opt(options, "disp", False)
    print(" >> Simulated Annealing optimization (with SCG local search)")

    n = x0.size
    x = deepcopy(x0).reshape(n, 1)
    nit = _opt(options, "nit", 100 * n)
    stat = dict(MaxIt=nit, Fx=np.zeros((nit, 1)), f_eval=0, g_eval=0, Temp=np.zeros((nit, 1)))
    local_search = _opt(options, "lmin", False)

    fx = f(x, mParam)
    g = gf(x, mParam)
    stat["f_eval"] += 1
    stat["g_eval"] += 1

    every, T0, TF, stall = 10, 1.0, 1.0e-2, 1000
    Tk, n_acc = T0, 0
    x_best, f_best = x, fx
    for run in range(nrun):
        print(" Run: {0} ".format(run))
        for i in range(nit):
            if i % every == 0:
                Tk = T0 * (TF / T0) ** (i / nit)
            x_try = x + Tk * np.random.randn(n, 1) - (1.0 - Tk) * g
            f_try = f(x_try, mParam)
            g_try = gf(x_try, mParam)
            stat["f_eval"] += 1
            stat["g_eval"] += 1
            rise = f_try - fx
            if rise < 0 or np.exp(-rise / Tk) > np.random.rand():
                n_acc += 1
                x, fx, g = x_try, f_try, g_try
            stat["Fx"][i] = fx
            stat["Temp"][i] = Tk
            if disp:
                print(" {0}:\tfx={1:.3f}\tTemp={2:.3f}\tAcc.Ratio = {3:.2f}%".format(i, fx, Tk, 100 * n_acc / (i + 1)))
            if i >= stall and np.diff(stat["Fx"][i - stall:i], axis=0).sum() == 0.0:
                print(" Algorithm terminated in {0} iterations".format(i))
                stat["MaxIt"] = i
                break
        x_best, f_best = x, fx
        if local_search:
            print(" >> Searching locally using SCG ...")
            x_loc, f_loc, p_loc, _ = optim_SCG(f, x_best, gf, dict(nit=1000, xtol=xtol, ftol=ftol, disp=True),
                                               deepcopy(mParam))
            if f_loc < f_best:
                x_best, f_best, mParam = x_loc, f_loc, p_loc
    return x_best, f_best, mParam, stat


# ---------------------------------------------------------------------------
# batched SCG on the device
# ---------------------------------------------------------------------------
class _ParamVectors:
    """Storage convention of the batched SCG driver: a parameter array is one flat
    tensor [P x nA | P x nb]; its two blocks, viewed as (P,N,D,D) and (P,N,D), are the
    arrays the evaluation takes, so nothing is packed or unpacked between the optimiser
    and the kernels."""

    def __init__(self, A0, b0):
        self.P = A0.shape[0]
        self.shape_A, self.shape_b = tuple(A0.shape), tuple(b0.shape)
        self.nA, self.nb = A0[0].numel(), b0[0].numel()
        self.n = self.nA + self.nb
        self.dev = A0.device

    def new(self):
        import torch
        return torch.zeros((self.P * self.n,), dtype=torch.float64, device=self.dev)

    def parts(self, v):
        cut = self.P * self.nA
        return v[:cut].view(self.shape_A), v[cut:].view(self.shape_b)

    def rows(self, v):
        cut = self.P * self.nA
        return v[:cut].view(self.P, self.nA), v[cut:].view(self.P, self.nb)


class _DeviceParamOps(_ParamVectors):
    """The passes of one SCG iteration as libvgpa_hip.so kernels (vgpa_scg_*)."""

    def __init__(self, A0, b0):
        import torch
        super().__init__(A0, b0)
        self.nch = int(min(128, max(1, -(-max(self.nA, self.nb) // 16384))))
        self.work = torch.empty((self.P * 2 * self.nch * 4,), dtype=torch.float64, device=self.dev)

    def _reduce(self, name, *ptr_args, mode):
        import torch
        from .. import engine
        out = torch.empty((self.P, 4), dtype=torch.float64, device=self.dev)
        engine.call(name, self.P, self.nA, self.nb, *[engine._p(t) for t in ptr_args], engine._p(mode), self.nch,
                    engine._p(self.work), engine._p(out), engine._stream())
        return out

    def dots(self, d, g, mode=None):
        return self._reduce("vgpa_scg_dots", d, g, mode=mode)

    def step(self, x, d, g, coef, mode, y):
        from .. import engine
        engine.call("vgpa_scg_step", self.P, self.nA, self.nb, engine._p(x), engine._p(d), engine._p(g),
                    engine._p(coef.contiguous()), engine._p(mode), engine._p(y), engine._stream())

    def curvature(self, d, g_probe, g, mode):
        return self._reduce("vgpa_scg_curvature", d, g_probe, g, mode=mode)[:, 0]

    def trial_stats(self, d, g, g_try, alpha, mode):
        return self._reduce("vgpa_scg_trial_stats", d, g, g_try, alpha.contiguous(), mode=mode)

    def update(self, x, x_try, g, g_try, d, gamma, mode):
        return self._reduce("vgpa_scg_update", x, x_try, g, g_try, d, gamma.contiguous(), mode=mode)


class _TorchParamOps(_ParamVectors):
    """The same passes written with torch operations: the statement of what each
    vgpa_scg_* kernel computes.  scg_batched uses it only for host tensors, which only
    happens with an injected ``evaluate`` (tests of the recursion logic); device tensors
    always take the kernels."""

    def _dot(self, a, b):
        (aA, ab), (bA, bb) = self.rows(a), self.rows(b)
        return (aA * bA).sum(dim=1) + (ab * bb).sum(dim=1)

    def _out(self, cols, mode):
        import torch
        out = torch.zeros((self.P, 4), dtype=torch.float64, device=self.dev)
        for k, c in enumerate(cols):
            out[:, k] = c if mode is None else torch.where(mode != 0, c, torch.zeros_like(c))
        return out

    def dots(self, d, g, mode=None):
        return self._out([self._dot(d, g), self._dot(d, d), self._dot(g, g)], mode)

    def step(self, x, d, g, coef, mode, y):
        import torch
        for xs, ds, gs, ys in zip(self.rows(x), self.rows(d), self.rows(g), self.rows(y)):
            m = mode.unsqueeze(1)
            ds.copy_(torch.where(m == 2, -gs, ds))
            ys.copy_(torch.where(m != 0, xs + coef.unsqueeze(1) * ds, ys))

    def curvature(self, d, g_probe, g, mode):
        return self._out([self._dot(d, g_probe - g)], mode)[:, 0]

    def trial_stats(self, d, g, g_try, alpha, mode):
        step_max = sum_abs = None
        for ds, ts in zip(self.rows(d), self.rows(g_try)):
            sm, sa = (alpha.unsqueeze(1) * ds).abs().amax(dim=1), ts.abs().sum(dim=1)
            step_max = sm if step_max is None else step_max.maximum(sm)
            sum_abs = sa if sum_abs is None else sum_abs + sa
        return self._out([step_max, self._dot(g_try, g_try), self._dot(g_try, g - g_try), sum_abs], mode)

    def update(self, x, x_try, g, g_try, d, gamma, mode):
        import torch
        copy = ((mode & 9) != 0).unsqueeze(1)                   # bits 1, 8: x <- x_try, g <- g_try
        restart = ((mode & 2) != 0).unsqueeze(1)
        turn = ((mode & 5) != 0).unsqueeze(1)                    # bits 1, 4: new direction from the new g
        for xs, xt, gs, gt, ds in zip(self.rows(x), self.rows(x_try), self.rows(g), self.rows(g_try), self.rows(d)):
            xs.copy_(torch.where(copy, xt, xs))
            gs.copy_(torch.where(copy, gt, gs))
            ds.copy_(torch.where(restart, -gs, torch.where(turn, gamma.unsqueeze(1) * ds - gs, ds)))
        return self.dots(d, g, torch.where((mode & 8) != 0, torch.zeros_like(mode), mode))


def scg_batched(prob, A0, b0, nit=150, xtol=1.0e-6, ftol=1.0e-8, evaluate=None, disp=False):
    """SCG for the P problems of ``prob`` (an ``engine.Problem``) at once.

    A0 (P,N,D,D) / (P,N), b0 (P,N,D) / (P,N): device tensors with the starting
    parameters.  Every problem follows, independently, the recursion of ``optim_SCG``
    (optimize.py:132-238) -- its own accept / reject decisions, scale beta, restarts
    after n accepted steps and termination tests; finished problems are frozen while
    the others continue.  The work per iteration is one batched gradient evaluation at
    the probe points, one at the trial points (the reference evaluates F a third time
    at the accepted point, which is the trial point again: that value is reused), and
    five passes over the parameter arrays (vgpa_scg_step twice, vgpa_scg_curvature,
    vgpa_scg_trial_stats, vgpa_scg_update).  An accepted trial point and its gradient are not
    copied: the arrays (x, g) and (x_try, g_try) change roles, and only the rows of problems that
    did not move are copied back (15 array traversals per iteration instead of 18).  The gradient before the last accepted step
    is needed only through sum|g| (the 'gradx' column), which is kept as a scalar.

    ``evaluate(A, b, need_grad)`` -> (F (P,), grad (P,n) or None) replaces the device
    evaluation (used by tests); by default it is engine.free_energy_and_gradient.

    Returns (A, b, F, stat): final parameters, final free energies (P,), and stat with
    'Fx', 'beta', 'gradx' (nit,P) as the per-problem arrays of ``optim_SCG``, 'MaxIt'
    (P,), 'f_eval', 'g_eval' (P,) counted the way the reference counts them.
    """
    import torch
    from .. import engine

    ops = (_DeviceParamOps if A0.is_cuda else _TorchParamOps)(A0, b0)
    if not A0.is_cuda and evaluate is None:
        raise RuntimeError("scg_batched evaluates on the HIP device: A0, b0 must be device tensors")
    P, n, nA = ops.P, ops.n, ops.nA
    dev, f64, i32 = ops.dev, torch.float64, torch.int32

    # F and the gradient are all the recursion uses: the symmetric intermediates stay in their packed
    # form where the kernels take it (same bits, fewer bytes) and the rows of Psi are not stored
    packed = evaluate is None and engine.packed_supported(prob)

    def fg(v, g_out):
        """F at the parameter array v; the gradient goes into g_out."""
        A, b = ops.parts(v)
        if evaluate is None:
            st, _, _ = engine.free_energy_and_gradient(prob, A, b, out=ops.parts(g_out), costate=False, packed=packed)
            return st.F
        F, grad = evaluate(A, b, True)
        gA, gb = ops.rows(g_out)
        gA.copy_(grad[:, :nA])
        gb.copy_(grad[:, nA:])
        return F

    x, y, g, g_try, d = ops.new(), ops.new(), ops.new(), ops.new(), ops.new()
    for dst, src in zip(ops.parts(x) + ops.parts(y), (A0, b0, A0, b0)):
        dst.copy_(src)           # y always holds valid parameters, also in rows a pass skips
    stat = dict(Fx=torch.zeros((nit, P), dtype=f64, device=dev), beta=torch.zeros((nit, P), dtype=f64, device=dev),
                gradx=torch.zeros((nit, P), dtype=f64, device=dev),
                MaxIt=torch.full((P,), nit, dtype=torch.int64, device=dev),
                f_eval=torch.ones((P,), dtype=torch.int64, device=dev),
                g_eval=torch.ones((P,), dtype=torch.int64, device=dev))

    zeros = torch.zeros((P,), dtype=f64, device=dev)
    every = torch.ones((P,), dtype=i32, device=dev)
    f_cur = fg(x, g)
    f_old = f_cur.clone()
    absg_new = ops.trial_stats(g, g, g, zeros, every)[:, 3]
    absg_old = absg_new.clone()
    slopes = ops.update(x, x, g, g, d, zeros, 2 * every)          # d = -g; {d.g, d.d, g.g}
    accepted = torch.ones((P,), dtype=torch.bool, device=dev)
    active = torch.ones((P,), dtype=torch.bool, device=dev)
    n_acc = torch.zeros((P,), dtype=torch.int64, device=dev)
    beta = torch.ones((P,), dtype=f64, device=dev)
    mu = torch.zeros((P,), dtype=f64, device=dev)
    kappa = torch.ones((P,), dtype=f64, device=dev)
    theta = torch.zeros((P,), dtype=f64, device=dev)
    f_final = f_cur.clone()
    sigma0, beta_lo, beta_hi = 1.0e-3, 1.0e-15, 1.0e+100

    def finish(mask, j, value):
        """Freeze the problems in ``mask`` with iteration count j and final value."""
        nonlocal active
        mask = mask & active
        stat["MaxIt"] = torch.where(mask, torch.full_like(stat["MaxIt"], j), stat["MaxIt"])
        f_final.copy_(torch.where(mask, value, f_final))
        active = active & ~mask

    for j in range(nit):
        if not bool(active.any().item()):
            break
        live0 = active.clone()
        probe = accepted & active
        # slope / curvature for the problems whose last step was accepted (optimize.py:150-156)
        dg, dd, gg = slopes[:, 0], slopes[:, 1], slopes[:, 2]
        uphill = probe & (dg >= 0.0)
        mu = torch.where(probe, torch.where(uphill, -gg, dg), mu)
        kappa = torch.where(probe, torch.where(uphill, gg, dd), kappa)
        finish(probe & (kappa < _EPS), j, f_cur)
        probe = probe & active
        uphill = uphill & probe
        sigma = torch.where(probe, sigma0 / torch.sqrt(kappa), zeros)
        if bool(probe.any().item()):
            pmode = probe.to(i32) + uphill.to(i32)
            ops.step(x, d, g, sigma, pmode, y)                                   # optimize.py:158
            fg(y, g_try)
            theta = torch.where(probe, ops.curvature(d, g_try, g, pmode) / sigma, theta)   # :159
            stat["f_eval"] += probe.long()
            stat["g_eval"] += probe.long()

        delta = theta + beta * kappa                                             # optimize.py:162-167
        fix = (delta <= 0) & active
        delta = torch.where(fix, beta * kappa, delta)
        beta = torch.where(fix, beta - theta / kappa, beta)
        alpha = torch.where(active, -(mu / delta), zeros)

        ops.step(x, d, g, alpha, active.to(i32), y)                              # optimize.py:170
        f_try = fg(y, g_try)
        stat["f_eval"] += active.long()

        ratio = 2.0 * (f_try - f_old) / (alpha * mu)
        ok = (ratio >= 0) & active
        accepted = torch.where(active, ok, accepted)
        n_acc += ok.long()
        trial = ops.trial_stats(d, g, g_try, alpha, ok.to(i32))
        f_cur = torch.where(ok, f_try, torch.where(active, f_old, f_cur))

        live = active.clone()
        stat["Fx"][j] = torch.where(live, f_cur, stat["Fx"][j])
        stat["beta"][j] = torch.where(live, beta, stat["beta"][j])
        stat["gradx"][j] = torch.where(live, torch.where(ok, absg_new, absg_old), stat["gradx"][j])
        if disp:
            print(" {0}:\tmin fx={1:.3f}\tactive={2}".format(j, float(f_cur.min().item()), int(live.sum().item())))

        small = (trial[:, 0] <= xtol) & ((f_try - f_old).abs() <= ftol)
        finish(ok & small, j, f_try)
        moved = ok & active
        f_old = torch.where(moved, f_try, f_old)
        absg_old = torch.where(moved, absg_new, absg_old)
        absg_new = torch.where(moved, trial[:, 3], absg_new)
        stat["f_eval"] += moved.long()
        stat["g_eval"] += moved.long()
        finish(moved & (trial[:, 1] == 0.0), j, f_cur)

        grow = active & (ratio < 0.25)
        beta = torch.where(grow, torch.clamp(4.0 * beta, max=beta_hi), beta)
        shrink = active & (ratio > 0.75)
        beta = torch.where(shrink, torch.clamp(0.5 * beta, min=beta_lo), beta)

        restart = active & (n_acc == n)
        gamma = torch.where(moved, torch.clamp(trial[:, 2] / mu, min=0.0), zeros)
        # Every accepted problem moves (also one that has just finished: its x is final).  The
        # arrays change roles instead of being copied: (y, g_try) become (x, g).  Rows that must not
        # move are copied back: problems that were live at the top of this iteration and were not
        # accepted (rejected, or finished before the step).  Rows of problems that finished on an
        # accepted step are copied the other way, so that a finished problem has the same rows in
        # both sets and later exchanges leave it alone.
        x, y, g, g_try = y, x, g_try, g
        stay = live0 & ~ok
        if bool(stay.any().item()):
            ops.update(x, y, g, g_try, d, zeros, 8 * stay.to(i32))
        done = ok & ~active
        if bool(done.any().item()):
            ops.update(y, x, g_try, g, d, zeros, 8 * done.to(i32))
        slopes = ops.update(x, y, g, g_try, d, gamma, 4 * ok.to(i32) + 2 * restart.to(i32))
        n_acc = torch.where(restart, torch.zeros_like(n_acc), n_acc)

    # problems that ran out of iterations report the last accepted value (optimize.py:245)
    f_final = torch.where(active, f_old, f_final)
    A, b = ops.parts(x)
    return A, b, f_final, stat
