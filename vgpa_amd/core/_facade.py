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


def host_vec(t, D):
    """Device (1,N,D) / (1,N) -> NumPy (N,D) / (N,1)."""
    a = t[0].cpu().numpy()
    return a.reshape(-1, 1) if D == 1 else a


def host_mat(t, D):
    """Device (1,N,D,D) / (1,N) -> NumPy (N,D,D) / (N,1)."""
    a = t[0].cpu().numpy()
    return a.reshape(-1, 1) if D == 1 else a
