"""Host-side marshalling shared by the reference-signature entry points: NumPy in,
device kernels, NumPy out, with the reference's array shapes."""
import numpy as np

from .. import _lib, engine

# module-name suffix of an ``energy_mode`` function -> kernel id
_SYSTEM_BY_MODULE = {
    "sys_ornstein_uhlenbeck": _lib.SYS_OU,
    "sys_double_well": _lib.SYS_DW,
    "sys_lorenz_63": _lib.SYS_L63,
    "sys_lorenz_63_ut": _lib.SYS_L63_UT,
    "sys_lorenz_96_ut": _lib.SYS_L96_UT,
}


def system_of(fun):
    """Kernel id for ``mParam['fun']``.  Functions of this package carry
    ``_vgpa_system``; an ``energy_mode`` imported from a module named like one of the
    reference's ``dynamics/sys_*.py`` files is recognised by that name, so a driver
    that still passes the reference's function object gets the device kernel too.
    Anything else is a user plug-in and is evaluated on the host (-1)."""
    sid = getattr(fun, "_vgpa_system", None)
    if sid is not None:
        return sid
    mod = getattr(fun, "__module__", "") or ""
    if getattr(fun, "__name__", "") == "energy_mode":
        return _SYSTEM_BY_MODULE.get(mod.rsplit(".", 1)[-1], -1)
    return -1


def unpack_Ab(Ab, N, D):
    """core/varFreeEnergy.py:64-69: A then b, C order; accepts (n,1) or (n,)."""
    Ab = np.asarray(Ab, dtype=np.float64).reshape(-1)
    if D == 1:
        if Ab.size != 2 * N:
            raise ValueError(f"Ab has {Ab.size} entries, expected {2 * N}")
        return Ab[:N], Ab[N:]
    K = N * D * D
    if Ab.size != K + N * D:
        raise ValueError(f"Ab has {Ab.size} entries, expected {K + N * D}")
    return Ab[:K].reshape(N, D, D), Ab[K:].reshape(N, D)


def dev_batch1(x, D, kind):
    """Host array of one problem -> device tensor with a leading batch axis of 1.
    kind 'v': (N,D) [1-D: (N,) or (N,1)], kind 'm': (N,D,D) [1-D: (N,) or (N,1)]."""
    x = np.asarray(x, dtype=np.float64)
    if D == 1:
        return engine.to_dev(x.reshape(1, -1))
    shp = (1, -1, D) if kind == "v" else (1, -1, D, D)
    return engine.to_dev(x.reshape(shp))


def to_host(tensors):
    """Device tensors -> NumPy arrays, through page-locked host buffers: the copies run at
    PCIe speed (a pageable destination is staged at a few GB/s) and are queued together with
    one synchronisation at the end.  The arrays own their buffers (through the tensor they
    are views of); torch's host allocator recycles a buffer once its array is dropped."""
    import torch
    host = []
    for t in tensors:
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        host.append(h)
    torch.cuda.current_stream().synchronize()
    return [h.numpy() for h in host]


class AsyncPublisher:
    """Copies arrays to page-locked host memory on a stream of its own while the evaluation that
    produces them is still running: ``ready(name, tensor)`` (the ``on_ready`` hook of
    engine.free_energy) makes the copy stream wait for what the compute stream has enqueued so
    far and queues the copy behind it; ``collect()`` waits for the copies and returns the NumPy
    arrays by name.  The tensors must stay referenced until collect() has returned (the State of
    the evaluation holds them)."""
    _streams = {}

    def __init__(self):
        import torch
        dev = torch.cuda.current_device()
        if dev not in AsyncPublisher._streams:
            AsyncPublisher._streams[dev] = torch.cuda.Stream(device=dev)
        self._stream = AsyncPublisher._streams[dev]
        self._host = {}

    def ready(self, name, t):
        import torch
        if t is None:
            return
        done = torch.cuda.Event()
        done.record()                                   # on the compute stream: everything enqueued so far
        src = t[0]
        h = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
        with torch.cuda.stream(self._stream):
            self._stream.wait_event(done)
            h.copy_(src, non_blocking=True)
        self._host[name] = h

    def collect(self):
        self._stream.synchronize()
        return {k: h.numpy() for k, h in self._host.items()}


# ---------------------------------------------------------------------------
# make_problem uploads a dozen small arrays and inverts Sigma and R on the host: about half a
# millisecond per call, paid by every varFreeEnergy of an optimiser run whose sde_struct never
# changes.  The last problem is kept, keyed on the VALUES it was built from (the reference's drivers
# edit sde_struct in place, e.g. theta in the parameter-estimation loop).
# ---------------------------------------------------------------------------
_problem_cache = None


def _problem_key(system, sde, m0, S0, fun):
    def raw(x):
        a = np.asarray(x)
        return (a.dtype.str, a.shape, a.tobytes())
    px0 = sde["px0"]
    return (system, id(fun) if system < 0 else None, int(sde["N"]), int(sde["D"]), float(sde["dt"]),
            str(sde.get("ode_method", "EULER")).upper(),
            tuple(raw(sde[k]) if k in sde else None for k in ("Sig", "theta", "Rig", "H", "obsX", "obsY")),
            raw(px0["mu0"]), raw(px0["tau0"]), raw(m0), raw(S0))


def cached_problem(system, sde, m0, S0, fun=None):
    """engine.make_problem(system, sde, m0, S0, P=1, fun=fun), re-used while every value it depends on is unchanged."""
    global _problem_cache
    import torch
    key = _problem_key(system, sde, m0, S0, fun)
    c = _problem_cache
    if c is not None and c[0] == key and c[1] == torch.cuda.current_device():
        return c[2]
    prob = engine.make_problem(system, sde, m0, S0, P=1, fun=fun)
    _problem_cache = (key, torch.cuda.current_device(), prob)
    return prob


def host_vec(t, D):
    """Device (1,N,D) / (1,N) -> NumPy (N,D) / (N,1)."""
    a = to_host([t[0]])[0]
    return a.reshape(-1, 1) if D == 1 else a


def host_mat(t, D):
    """Device (1,N,D,D) / (1,N) -> NumPy (N,D,D) / (N,1)."""
    a = to_host([t[0]])[0]
    return a.reshape(-1, 1) if D == 1 else a


# ---------------------------------------------------------------------------
# The device state behind the arrays varFreeEnergy last published.  The reference's
# optimisers call gradL_Ab(x, mParam) right after varFreeEnergy(x, mParam); when the six
# arrays in mParam are still the very objects published (and unchanged, and x is the same
# point), gradL_Ab takes m, S, lam, Psi, <f>, <df/dx> from the device instead of uploading
# them again.
# ---------------------------------------------------------------------------
PUBLISHED_KEYS = ("mt", "St", "Efx", "Edf", "lamt", "Psit")
_published = None


def _fingerprint(a):
    """A strided sample of an array (64 values), to notice in-place changes cheaply."""
    flat = a.reshape(-1)
    step = max(1, flat.size // 64)
    return flat[::step][:64].tobytes()


def remember_published(mParam, prob, Ad, bd, st):
    """Called by varFreeEnergy after it has put the six arrays into ``mParam``.  The arrays are
    made read-only: code that wants to edit them takes a copy (as the reference's drivers do
    with deepcopy), and a copy is a different object, which sends gradL_Ab down the upload path."""
    global _published
    for k in PUBLISHED_KEYS:
        mParam[k].flags.writeable = False
    _published = dict(arrays={k: mParam[k] for k in PUBLISHED_KEYS},
                      prints={k: _fingerprint(mParam[k]) for k in PUBLISHED_KEYS},
                      prob=prob, Ad=Ad, bd=bd, st=st, sde=mParam["sde_struct"], fun=mParam["fun"],
                      consumed=_consumed_by_gradient(mParam["sde_struct"]))


def _consumed_by_gradient(sde):
    """The entries of ``sde_struct`` the reference's gradL_Ab reads at call time
    (core/gradL_Ab.py:56, :62, :118, :125), by value: a driver may edit the dictionary in
    place between varFreeEnergy and gradL_Ab, and the cached problem must not outlive that."""
    return (np.array(sde["Sig"], dtype=np.float64, copy=True).tobytes(), float(sde["dt"]),
            str(sde.get("ode_method", "EULER")).upper())


def recall_published(mParam, Ad, bd):
    """The cached (problem, state) if ``mParam`` still holds what was published for the
    point (Ad, bd); None otherwise."""
    import torch
    c = _published
    if c is None or mParam.get("sde_struct") is not c["sde"] or mParam.get("fun") is not c["fun"]:
        return None
    if _consumed_by_gradient(mParam["sde_struct"]) != c["consumed"]:
        return None
    for k in PUBLISHED_KEYS:
        a = mParam.get(k)
        if a is not c["arrays"][k] or _fingerprint(a) != c["prints"][k]:
            return None
    if Ad.shape != c["Ad"].shape or not (torch.equal(Ad, c["Ad"]) and torch.equal(bd, c["bd"])):
        return None
    return c["prob"], c["st"]
