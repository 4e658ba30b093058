"""Device-resident evaluation of the variational free energy and its gradient.

Everything here works on float64 torch tensors that live on the GPU, with a
leading batch axis P of independent problems, and calls the HIP kernels of
``libvgpa_hip.so`` through ctypes (``vgpa_amd._lib``).  PyTorch is used only to
own device memory, to pick the HIP stream, and for O(P) / O(D^2) glue (adding
the three energy terms, scaling hyper-parameter gradients).  There is no CPU
path: without a HIP device every call raises.

The pipeline of one evaluation (reference: core/varFreeEnergy.py:80-111,
core/gradL_Ab.py:156-180):

    fwd_scan -> obs_terms -> esde -> reduce_energy -> bwd_scan -> kl0 -> F
    grad_assemble -> dL/dA, dL/db
"""
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import call

F64 = torch.float64


def device():
    if not torch.cuda.is_available():
        raise RuntimeError("vgpa_amd needs a HIP device (MI355X); none is visible and there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def to_dev(x, shape=None):
    """Host array / scalar / tensor -> contiguous float64 device tensor."""
    if isinstance(x, torch.Tensor):
        t = x.to(device=device(), dtype=F64)
    else:
        a = np.asarray(x, dtype=np.float64)
        if not (a.flags.writeable and a.flags.c_contiguous):
            a = np.array(a, dtype=np.float64, order="C", copy=True)
        t = torch.from_numpy(a).to(device())
    if shape is not None:
        t = t.reshape(shape)
    return t.contiguous()


def _p(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "kernel operands must be contiguous device tensors"
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def method_id(method):
    """core/fwd_ode.py:57-82: case-insensitive name, ValueError otherwise."""
    try:
        return _lib.METHODS[str(method).upper()]
    except KeyError:
        raise ValueError("Unknown method of integration") from None


def trapz_weights(N, idx=None):
    """Weights w with mytrapz(f, dt, idx) == dt * sum_t w[t] f[t]
    (auxiliary/numerical.py:231-284): trapezoid rule on each segment [f, l)
    between consecutive observation indices; the sample at l is not part of the
    segment and nothing from idx[-1] on is integrated."""
    w = np.zeros(N, dtype=np.float64)
    if idx is None:
        segs = [(0, N)]
    else:
        segs, f = [], 0
        for l in np.asarray(idx).reshape(-1):
            segs.append((f, int(l)))
            f = int(l)
    for f, l in segs:
        f, l = max(f, 0), min(l, N)
        if l - f >= 2:
            w[f:l] += 1.0
            w[f] -= 0.5
            w[l - 1] -= 0.5
    return w


@dataclass
class Problem:
    """Everything that stays fixed while A(t), b(t) change: sizes, integrator,
    noise, drift parameters, observations, prior and initial moments.  Arrays are
    device tensors; parameter arrays are either shared by all P problems (stride 0)
    or carry a leading P axis."""
    system: int
    P: int
    N: int
    D: int
    dt: float
    method: int
    Sigma: torch.Tensor            # (D,D) or (P,D,D); 1-D: (1,) or (P,)
    SigInv: torch.Tensor           # same shape
    sig_inv_diag: torch.Tensor     # (D,) or (P,D)
    sig_is_diag: bool
    theta: torch.Tensor            # (n_theta,) or (P,n_theta)
    m0: torch.Tensor               # (P,D)
    S0: torch.Tensor               # (P,D,D); 1-D: (P,)
    m0_is_row: bool
    mu0: torch.Tensor              # (D,) or (P,D)
    tau0: torch.Tensor             # (D,D) or (P,D,D)
    obs_idx: torch.Tensor          # (M,) int32
    obs_flag: torch.Tensor         # (N,) int32
    obsY: torch.Tensor             # (M,d) or (P,M,d)
    H: torch.Tensor                # (D,d)
    Ri: torch.Tensor               # (d,d) "Ri" of chol_inv(R): inv(chol) inv(chol)'
    Qi: torch.Tensor               # (d,d) inv(chol(R))
    logconst: float                # M (d log 2pi + log det R)
    w: torch.Tensor                # (N,) trapezoid weights
    sym_fwd: bool = False          # S0 and Sigma are exactly symmetric
    sym_jump: bool = False         # 0.5 H' Ri H is exactly symmetric (H, Ri diagonal)
    fun: Optional[object] = None   # user energy function evaluated on the host (system < 0)
    sde_struct: Optional[dict] = None
    # (key, tensor): results that depend on a few fields only, recomputed when one of those
    # fields is replaced or written in place (see _tensors_key)
    kl0_cache: Optional[tuple] = field(default=None, repr=False)
    jumps_cache: Optional[tuple] = field(default=None, repr=False)

    @property
    def M(self):
        return int(self.obs_idx.numel())


def _chol_inv_host(R):
    """(Ri, Qi) of auxiliary/numerical.py:10-48 for the observation noise (a
    constant of the run): Qi = inv(chol(R)), Ri = Qi Qi'."""
    R = np.asarray(R, dtype=np.float64)
    if R.ndim == 1:
        R = np.diag(R)
    Qi = np.linalg.inv(np.linalg.cholesky(R))
    return Qi.dot(Qi.T), Qi


def make_problem(system, sde, m0, S0, P=1, theta=None, fun=None):
    """Build a Problem from the reference's ``sde_struct`` dictionary (keys as in
    lorenz-96-ut.ipynb cell 3) and the initial moments.  ``theta`` may override
    sde['theta'] with a (P, n_theta) array for a parameter grid; m0 / S0 may carry
    a leading P axis."""
    D, N = int(sde["D"]), int(sde["N"])
    dt = float(sde["dt"])
    method = method_id(sde.get("ode_method", "EULER"))
    if D > _lib.MAX_D:
        raise _lib.VgpaError(f"D={D} exceeds the largest supported state dimension {_lib.MAX_D}")
    obs_idx = np.asarray(sde["obsX"]).reshape(-1).astype(np.int64)
    if obs_idx.size and (obs_idx.min() < 0 or obs_idx.max() >= N):
        raise IndexError("observation index out of range")
    obs_flag = np.zeros(N, dtype=np.int32)
    obs_flag[obs_idx] = 1
    obsY = np.asarray(sde["obsY"], dtype=np.float64)
    # a user-supplied energy function (system < 0) owns its drift parameters: the reference's
    # varFreeEnergy / gradL_Ab never read sde_struct['theta'] themselves
    th_src = theta if theta is not None else (sde.get("theta", 0.0) if system < 0 else sde["theta"])
    th = np.asarray(th_src, dtype=np.float64)
    mu0 = np.asarray(sde["px0"]["mu0"], dtype=np.float64)
    tau0 = np.asarray(sde["px0"]["tau0"], dtype=np.float64)
    m0a = np.asarray(m0.cpu() if isinstance(m0, torch.Tensor) else m0, dtype=np.float64)
    S0a = np.asarray(S0.cpu() if isinstance(S0, torch.Tensor) else S0, dtype=np.float64)
    if D == 1:
        Sigma = np.asarray(sde["Sig"], dtype=np.float64).reshape(-1)
        SigInv = 1.0 / Sigma
        sid = SigInv
        sig_is_diag = True
        R = float(np.asarray(sde["Rig"]).reshape(-1)[0])
        Ri, Qi = np.array([[1.0 / R]]), np.array([[1.0 / np.sqrt(R)]])
        H = np.asarray(sde["H"], dtype=np.float64).reshape(1, 1)
        d = 1
        logconst = obs_idx.size * (_LOG2PI + np.log(R))
        obsY = obsY.reshape((-1, obs_idx.size, 1)) if obsY.size != obs_idx.size else obsY.reshape(obs_idx.size, 1)
        th = th.reshape(-1, 1) if th.size > 1 else th.reshape(1)
        m0_is_row = False
        m0a = np.broadcast_to(m0a.reshape(-1, 1) if m0a.size > 1 else m0a.reshape(1, 1), (P, 1))
        S0a = np.broadcast_to(S0a.reshape(-1) if S0a.size > 1 else S0a.reshape(1), (P,))
        mu0 = mu0.reshape(-1)
        tau0 = tau0.reshape(-1)
    else:
        Sigma = np.asarray(sde["Sig"], dtype=np.float64)
        SigInv = np.linalg.inv(Sigma)                      # core/gradL_Ab.py:137
        sid = np.diagonal(SigInv, axis1=-2, axis2=-1).copy()
        sig_is_diag = bool(np.count_nonzero(SigInv - sid[..., None] * np.eye(D)) == 0)
        R = np.asarray(sde["Rig"], dtype=np.float64)
        Ri, Qi = _chol_inv_host(R)
        H = np.asarray(sde["H"], dtype=np.float64)
        d = obsY.shape[-1]
        Rm = np.diag(R) if R.ndim == 1 else R
        logconst = obs_idx.size * (d * _LOG2PI + 2.0 * np.sum(np.log(np.linalg.cholesky(Rm).diagonal())))
        m0_is_row = bool(P == 1 and m0a.ndim == 2 and m0a.shape == (1, D))
        m0a = np.broadcast_to(m0a.reshape(-1, D), (P, D))
        S0a = np.broadcast_to(S0a.reshape(-1, D, D), (P, D, D))
    prob = Problem(
        system=system, P=P, N=N, D=D, dt=dt, method=method,
        Sigma=to_dev(Sigma), SigInv=to_dev(SigInv), sig_inv_diag=to_dev(sid), sig_is_diag=sig_is_diag,
        theta=to_dev(th), m0=to_dev(m0a), S0=to_dev(S0a), m0_is_row=m0_is_row,
        mu0=to_dev(mu0), tau0=to_dev(tau0),
        obs_idx=torch.from_numpy(obs_idx.astype(np.int32)).to(device()),
        obs_flag=torch.from_numpy(obs_flag).to(device()),
        obsY=to_dev(obsY), H=to_dev(H), Ri=to_dev(Ri), Qi=to_dev(Qi), logconst=float(logconst),
        w=to_dev(trapz_weights(N, obs_idx)), fun=fun, sde_struct=sde)
    if D > 1:
        prob.sym_fwd = bool(np.array_equal(S0a, np.swapaxes(S0a, -1, -2)) and
                            np.array_equal(Sigma, np.swapaxes(Sigma, -1, -2)))
        isdiag = lambda M: np.count_nonzero(M - np.diag(np.diagonal(M))) == 0   # noqa: E731
        prob.sym_jump = bool(H.shape[0] == H.shape[1] and isdiag(H) and isdiag(Ri))
    return prob


class _TensorsKey:
    """The tensors a cached result was computed from, with their in-place version counters.
    Unequal to a later key when a field has been reassigned to another tensor object or a
    tensor has been written in place.  (It keeps the tensors alive, so a new tensor cannot
    reuse the address of an old one unnoticed.)"""

    def __init__(self, *tensors, extra=None):
        self.tensors, self.versions, self.extra = tensors, tuple(t._version for t in tensors), extra

    def __eq__(self, other):
        return (isinstance(other, _TensorsKey) and len(self.tensors) == len(other.tensors)
                and all(a is b for a, b in zip(self.tensors, other.tensors))
                and self.versions == other.versions and self.extra == other.extra)

    def __ne__(self, other):
        return not self.__eq__(other)

    def __deepcopy__(self, memo):
        return None        # a copied Problem recomputes


def _tensors_key(*tensors, extra=None):
    return _TensorsKey(*tensors, extra=extra)


def esde_output_is_symmetric(system, D):
    """Whether the energy kernel of this system writes an exactly symmetric
    dEsde/dS (mirrored in the kernel), so that the backward scan may use its
    one-product form."""
    return system == _lib.SYS_L63 or (system == _lib.SYS_L96_UT and D in (16, 40))


_LOG2PI = 1.8378770664093453   # core/likelihood.py@10


# ---------------------------------------------------------------------------
# thin wrappers, one per C entry point (device tensors in, device tensors out)
# ---------------------------------------------------------------------------
def _mat_shape(P, N, D):
    return ((P, N), (P, N)) if D == 1 else ((P, N, D), (P, N, D, D))


def _bitwise_symmetric(t):
    """True when a (..., D, D) tensor equals its transpose exactly (small tensors
    only: this synchronises)."""
    return bool(torch.equal(t, t.transpose(-1, -2)))


def sym_pack(dense):
    """vgpa_sym_pack: (..., 40, 40) symmetric matrices -> (..., 1152), the packed layout the
    Lorenz-96 Euler kernels accept (see ``include/vgpa.h``)."""
    if dense.shape[-1] != 40 or dense.shape[-2] != 40:
        raise ValueError("sym_pack: the packed layout exists for 40 x 40 matrices only")
    dense = dense.contiguous()
    out = torch.empty(dense.shape[:-2] + (_lib.SYM_PACKED_40,), dtype=F64, device=dense.device)
    call("vgpa_sym_pack", dense.numel() // 1600, 40, _p(dense), _p(out), _stream())
    return out


def sym_unpack(packed):
    """vgpa_sym_unpack: (..., 1152) -> (..., 40, 40), mirrored above the block diagonal."""
    if packed.shape[-1] != _lib.SYM_PACKED_40:
        raise ValueError("sym_unpack: the last axis must hold %d doubles" % _lib.SYM_PACKED_40)
    packed = packed.contiguous()
    out = torch.empty(packed.shape[:-1] + (40, 40), dtype=F64, device=packed.device)
    call("vgpa_sym_unpack", packed.numel() // _lib.SYM_PACKED_40, 40, _p(packed), _p(out), _stream())
    return out


def fwd_scan(method, A, b, m0, S0, Sigma, dt, symmetric=None, packed=False):
    """vgpa_fwd_scan.  A (P,N,D,D) / (P,N) -> m, S.  ``symmetric`` selects the
    one-product-per-stage kernels; None checks S0 and Sigma for exact symmetry.
    ``packed``: S comes back as (P,N,1152) in the packed symmetric layout (Euler, D = 40,
    symmetric only; anything else raises)."""
    P, N = A.shape[0], A.shape[1]
    D = 1 if A.dim() == 2 else A.shape[-1]
    vs, ms = _mat_shape(P, N, D)
    m = torch.empty(vs, dtype=F64, device=A.device)
    S = torch.empty((P, N, _lib.SYM_PACKED_40) if packed else ms, dtype=F64, device=A.device)
    sstride = _par_stride(Sigma, P, D * D)
    if symmetric is None:
        symmetric = D > 1 and _bitwise_symmetric(S0) and _bitwise_symmetric(Sigma)
    flags = (_lib.SCAN_SYMMETRIC if (symmetric and D > 1) else 0) | (_lib.SCAN_PACKED if packed else 0)
    call("vgpa_fwd_scan", method, P, N, D, _p(A), _p(b), _p(m0), _p(S0), _p(Sigma), sstride, float(dt), flags,
         _p(m), _p(S), _stream())
    return m, S


def bwd_scan(method, A, dEdm, dEdS, dEobs_dm, dt, dEobs_dS=None, obs_flag=None, jumpS=None, symmetric=False,
             lam_last=None, Psi_last=None):
    """vgpa_bwd_scan -> lam, Psi.  ``symmetric`` promises exactly symmetric dEdS and
    jumps (see VGPA_SCAN_SYMMETRIC); ``lam_last`` / ``Psi_last`` give the state of the
    last row (zero when omitted)."""
    P, N = A.shape[0], A.shape[1]
    D = 1 if A.dim() == 2 else A.shape[-1]
    lam = torch.empty_like(dEdm)
    Psi = torch.empty_like(dEdS)
    jstride = 0
    if jumpS is not None and jumpS.numel() == P * D * D and P > 1:
        jstride = D * D
    call("vgpa_bwd_scan", method, P, N, D, _p(A), _p(dEdm), _p(dEdS), _p(dEobs_dm), _p(dEobs_dS),
         _p(obs_flag), _p(jumpS), jstride, float(dt), _lib.SCAN_SYMMETRIC if (symmetric and D > 1) else 0,
         _p(lam_last), _p(Psi_last), _p(lam), _p(Psi), _stream())
    return lam, Psi


def bwd_scan_grad(prob, A, b, m, S, dEdm, dEdS, dEobs_dm, Ef, Edf, jumpS=None, dEobs_dS=None, symmetric=False,
                  state_symmetric=False, outer=False, lam_last=None, Psi_last=None, out=None, costate=True,
                  packed=False):
    """vgpa_bwd_scan_grad -> lam, Psi, gLA, gLb: the backward scan and the gradient assembly
    in one call.  With the Euler method at D = 40, symmetric inputs (``symmetric``: dEdS and
    the jumps; ``state_symmetric``: S), an implicit Lorenz-96 <df/dx> (Edf None) and a diagonal
    SigInv one kernel does both; otherwise the library runs the two kernels in sequence.

    ``costate=False``: the caller does not need Psi(t) beyond t = 0 (an optimiser wants F and the
    gradient only).  Where the one-kernel path runs, Psi then comes back as (P, 1, D, D) holding
    Psi(0) and the other rows are never written; elsewhere the argument changes nothing.

    ``packed``: S and dEdS are given, and Psi comes back, in the packed symmetric layout
    ((P, N, 1152), see ``sym_pack``); only where the one-kernel Euler path runs, anything else raises.
    ``packed="dEdS"``: dEdS alone is packed, S and Psi are dense."""
    P, N, D = prob.P, prob.N, prob.D
    lam = torch.empty_like(dEdm)
    sflags = (_lib.SCAN_SYMMETRIC if (symmetric and D > 1) else 0) | \
             (_lib.SCAN_STATE_SYMMETRIC if (state_symmetric and D > 1) else 0)
    gflags = (_lib.GRAD_OUTER if outer else 0) | (_lib.GRAD_SIGINV_DIAG if prob.sig_is_diag else 0)
    if not costate and _lib.lib.vgpa_bwd_scan_grad_is_fused(prob.method, P, D, sflags, gflags, int(dEobs_dS is not None),
                                                           int(Edf is not None), prob.system):
        sflags |= _lib.SCAN_PSI_ROW0
        Psi = torch.empty((P, 1, _lib.SYM_PACKED_40) if packed is True else (P, 1, D, D), dtype=F64, device=dEdS.device)
    else:
        Psi = torch.empty_like(S)
    if packed is True:
        sflags |= _lib.SCAN_PACKED
    elif packed == "dEdS":
        sflags |= _lib.SCAN_PACKED_DEDS
    elif packed:
        raise ValueError("bwd_scan_grad: packed is False, True or 'dEdS'")
    if out is None:
        gLA, gLb = torch.empty_like(A), torch.empty_like(b)
    else:
        gLA, gLb = out
        if gLA.shape != A.shape or gLb.shape != b.shape or gLA.dtype != F64 or gLb.dtype != F64:
            raise ValueError("bwd_scan_grad: out tensors must match A and b (float64)")
    jstride = 0
    if jumpS is not None and jumpS.numel() == P * D * D and P > 1:
        jstride = D * D
    n_theta = 3 if prob.system in (_lib.SYS_L63, _lib.SYS_L63_UT) else 1
    call("vgpa_bwd_scan_grad", prob.method, P, N, D, _p(A), _p(b), _p(m), _p(S), _p(dEdm), _p(dEdS), _p(dEobs_dm),
         _p(dEobs_dS), _p(prob.obs_flag), _p(jumpS), jstride, _p(Ef), _p(Edf), prob.system, _p(prob.theta),
         _par_stride(prob.theta, P, n_theta), _p(prob.SigInv), _par_stride(prob.SigInv, P, D * D), float(prob.dt),
         sflags, gflags, _p(lam_last), _p(Psi_last), _p(lam), _p(Psi), _p(gLA), _p(gLb), _stream())
    return lam, Psi, gLA, gLb


def _par_stride(t, P, per_problem_numel):
    """0 when a parameter tensor is shared, else its per-problem element count."""
    return per_problem_numel if (t.numel() == P * per_problem_numel and P > 1) else 0


def esde(system, A, b, m, S, sig_inv_diag, theta, want_edf=True, want_hyper=True):
    """vgpa_esde -> dict(Esde_t, Ef, Edf, dEdm, dEdS, dth_t, dSig_t, status)."""
    P, N = A.shape[0], A.shape[1]
    D = 1 if A.dim() == 2 else A.shape[-1]
    vs, ms = _mat_shape(P, N, D)
    dev = A.device
    n_theta = 3 if system in (_lib.SYS_L63, _lib.SYS_L63_UT) else 1
    out = dict(
        Esde_t=torch.empty((P, N), dtype=F64, device=dev),
        Ef=torch.empty(vs, dtype=F64, device=dev),
        Edf=torch.empty(ms, dtype=F64, device=dev) if (want_edf or D == 1) else None,
        dEdm=torch.empty(vs, dtype=F64, device=dev),
        dEdS=torch.empty(ms, dtype=F64, device=dev),
        dth_t=torch.empty(vs, dtype=F64, device=dev) if want_hyper else None,
        dSig_t=torch.empty(vs, dtype=F64, device=dev) if (want_hyper and D > 1) else None,
        status=torch.zeros((P, N), dtype=torch.int32, device=dev),
    )
    call("vgpa_esde", system, P, N, D, _p(A), _p(b), _p(m), _p(S),
         _p(sig_inv_diag), _par_stride(sig_inv_diag, P, D),
         _p(theta), _par_stride(theta, P, n_theta),
         _p(out["Esde_t"]), _p(out["Ef"]), _p(out["Edf"]), _p(out["dEdm"]), _p(out["dEdS"]),
         _p(out["dth_t"]), _p(out["dSig_t"]), _p(out["status"]), _stream())
    return out


def esde_l96_packed(A, b, m, S, sig_inv_diag, theta, want_hyper=True, packed_S=True):
    """vgpa_esde_l96_ut_packed -> the dict of ``esde`` with dEdS (P,N,1152) and no Edf; S is (P,N,1152),
    or dense (P,N,40,40) with ``packed_S=False``."""
    P, N, D = A.shape[0], A.shape[1], A.shape[-1]
    if D != 40 or S.shape != ((P, N, _lib.SYM_PACKED_40) if packed_S else (P, N, D, D)):
        raise ValueError("esde_l96_packed: needs D == 40 and S of shape (P, N, %d), or (P, N, 40, 40) with packed_S=False"
                         % _lib.SYM_PACKED_40)
    dev = A.device
    out = dict(
        Esde_t=torch.empty((P, N), dtype=F64, device=dev),
        Ef=torch.empty((P, N, D), dtype=F64, device=dev),
        Edf=None,
        dEdm=torch.empty((P, N, D), dtype=F64, device=dev),
        dEdS=torch.empty((P, N, _lib.SYM_PACKED_40), dtype=F64, device=dev),
        dth_t=torch.empty((P, N, D), dtype=F64, device=dev) if want_hyper else None,
        dSig_t=torch.empty((P, N, D), dtype=F64, device=dev) if want_hyper else None,
        status=torch.zeros((P, N), dtype=torch.int32, device=dev),
    )
    layout = _lib.PACKED_DEDS | (_lib.PACKED_S if packed_S else 0)
    call("vgpa_esde_l96_ut_packed", P, N, layout, _p(A), _p(b), _p(m), _p(S), _p(sig_inv_diag), _par_stride(sig_inv_diag, P, D),
         _p(theta), _par_stride(theta, P, 1), _p(out["Esde_t"]), _p(out["Ef"]), _p(out["dEdm"]), _p(out["dEdS"]),
         _p(out["dth_t"]), _p(out["dSig_t"]), _p(out["status"]), _stream())
    return out


def reduce_energy(x, w, scale):
    """vgpa_reduce_energy: x (P,N) or (P,N,C) -> (P,) or (P,C)."""
    P, N = x.shape[0], x.shape[1]
    C = 1 if x.dim() == 2 else x.shape[2]
    out = torch.empty((P,) if x.dim() == 2 else (P, C), dtype=F64, device=x.device)
    call("vgpa_reduce_energy", P, N, C, _p(x), _p(w), float(scale), _p(out), _stream())
    return out


def obs_terms(prob, m, S, packed=False):
    """vgpa_obs_terms -> Eobs (P,), dEobs_dm (like m), jumpS (D,D).  ``packed``: S is (P,N,1152)
    in the packed symmetric layout (D = 40)."""
    P, N, D = prob.P, prob.N, prob.D
    Eobs = torch.empty((P,), dtype=F64, device=m.device)
    dm = torch.empty_like(m)
    # the jump of Psi at an observation, 0.5 H' Ri H, depends on the operators only: computed once
    # per Problem and pair of operators
    key = _tensors_key(prob.H, prob.Ri)
    fresh = prob.jumps_cache is None or prob.jumps_cache[0] != key
    jumpS = torch.empty((D, D) if D > 1 else (1,), dtype=F64, device=m.device) if fresh else None
    d = prob.obsY.shape[-1]
    ystride = prob.M * d if (prob.obsY.dim() == 3 and P > 1) else 0
    work = torch.empty((P, max(prob.M, 1)), dtype=F64, device=m.device)
    if packed:
        if D != 40:
            raise ValueError("obs_terms: the packed layout exists for D == 40 only")
        call("vgpa_obs_terms_packed", P, N, prob.M, d, _p(m), _p(S), _p(prob.obs_idx), _p(prob.obsY), ystride,
             _p(prob.H), _p(prob.Ri), _p(prob.Qi), prob.logconst, _p(work), _p(Eobs), _p(dm), _p(jumpS), _stream())
    else:
        call("vgpa_obs_terms", P, N, D, prob.M, d, _p(m), _p(S), _p(prob.obs_idx), _p(prob.obsY), ystride,
             _p(prob.H), _p(prob.Ri), _p(prob.Qi), prob.logconst, _p(work), _p(Eobs), _p(dm), _p(jumpS), _stream())
    if fresh:
        prob.jumps_cache = (key, jumpS)
    return Eobs, dm, prob.jumps_cache[1]


def kl0(prob):
    """vgpa_kl0 -> KL0 (P,).  Independent of A, b: cached on the Problem."""
    key = _tensors_key(prob.m0, prob.S0, prob.mu0, prob.tau0, extra=prob.m0_is_row)
    if prob.kl0_cache is None or prob.kl0_cache[0] != key:
        P, D = prob.P, prob.D
        out = torch.empty((P,), dtype=F64, device=prob.m0.device)
        call("vgpa_kl0", P, D, _p(prob.m0), _p(prob.S0), _p(prob.mu0), _par_stride(prob.mu0, P, D),
             _p(prob.tau0), _par_stride(prob.tau0, P, D * D), int(prob.m0_is_row), _p(out), _stream())
        prob.kl0_cache = (key, out)
    return prob.kl0_cache[1]


def grad_assemble(prob, A, b, m, S, lam, Psi, Ef, Edf, outer=False, out=None):
    """vgpa_grad_assemble -> gLA (like A), gLb (like b), already times dt.  With
    ``Edf=None`` the kernel derives <df/dx> from the mean (Lorenz systems only).
    ``out=(gLA, gLb)`` writes into existing contiguous tensors."""
    P, N, D = prob.P, prob.N, prob.D
    if out is None:
        gLA, gLb = torch.empty_like(A), torch.empty_like(b)
    else:
        gLA, gLb = out
        if gLA.shape != A.shape or gLb.shape != b.shape or gLA.dtype != F64 or gLb.dtype != F64:
            raise ValueError("grad_assemble: out tensors must match A and b (float64)")
    flags = (_lib.GRAD_OUTER if outer else 0) | (_lib.GRAD_SIGINV_DIAG if prob.sig_is_diag else 0)
    n_theta = 3 if prob.system in (_lib.SYS_L63, _lib.SYS_L63_UT) else 1
    call("vgpa_grad_assemble", P, N, D, _p(A), _p(b), _p(m), _p(S), _p(lam), _p(Psi), _p(Ef), _p(Edf),
         prob.system, _p(prob.theta), _par_stride(prob.theta, P, n_theta),
         _p(prob.SigInv), _par_stride(prob.SigInv, P, D * D), float(prob.dt), flags, _p(gLA), _p(gLb), _stream())
    return gLA, gLb


def ut_sigma_points(xbar, Pxx):
    """vgpa_ut_sigma_points: xbar (P,D), Pxx (P,D,D) -> chi (P,2D+1,D), status (P,)."""
    P, D = xbar.shape
    chi = torch.empty((P, 2 * D + 1, D), dtype=F64, device=xbar.device)
    status = torch.zeros((P,), dtype=torch.int32, device=xbar.device)
    call("vgpa_ut_sigma_points", P, D, _p(xbar), _p(Pxx), _p(chi), _p(status), _stream())
    return chi, status


def ut_moments(Y, want_cov=True):
    """vgpa_ut_moments: Y (P,K,C) -> ybar (P,C), Pyy (P,C,C) or None."""
    P, K, C = Y.shape
    ybar = torch.empty((P, C), dtype=F64, device=Y.device)
    Pyy = torch.empty((P, C, C), dtype=F64, device=Y.device) if want_cov else None
    call("vgpa_ut_moments", P, K, C, _p(Y), _p(ybar), _p(Pyy), _stream())
    return ybar, Pyy


def grad_esde_ms(x, Fx, mt, St, At, bt, dSigI):
    """vgpa_grad_esde_ms: x, Fx (K,D) -> (K, D + D*D)."""
    K, D = x.shape
    out = torch.empty((K, D * (D + 1)), dtype=F64, device=x.device)
    call("vgpa_grad_esde_ms", K, D, _p(x), _p(Fx), _p(mt), _p(St), _p(At), _p(bt), _p(dSigI), _p(out), _stream())
    return out


def mom_gauss(m, S, order, deriv):
    """vgpa_mom_gauss on flat device tensors."""
    out = torch.empty_like(m)
    call("vgpa_mom_gauss", m.numel(), _p(m), _p(S), int(order), int(deriv), _p(out), _stream())
    return out


def sde_paths(system, x0, theta, kdiag, noise, dt, burn=0, dtau=0.0, add0=None):
    """vgpa_sde_path: x0 (P,D), theta (n,) / (P,n), kdiag (D,) / (P,D), noise (P,N,D) -> X (P,N,D)."""
    P, N, D = noise.shape
    n_theta = 3 if system in (_lib.SYS_L63, _lib.SYS_L63_UT) else 1
    X = torch.empty((P, N, D), dtype=F64, device=noise.device)
    call("vgpa_sde_path", system, P, N, D, _p(x0), _p(add0), _p(theta), _par_stride(theta, P, n_theta),
         _p(kdiag), _par_stride(kdiag, P, D), int(burn), float(dtau), float(dt), _p(noise), _p(X), _stream())
    return X


def init_ab0(mt0, sig_diag, s0_diag, dt):
    """vgpa_init_ab0: mt0 (P,N,D), diagonals (D,) / (P,D) -> A0 (P,N,D,D), b0 (P,N,D)."""
    P, N, D = mt0.shape
    A0 = torch.empty((P, N, D, D), dtype=F64, device=mt0.device)
    b0 = torch.empty((P, N, D), dtype=F64, device=mt0.device)
    call("vgpa_init_ab0", P, N, D, _p(mt0), _p(sig_diag), _par_stride(sig_diag, P, D), _p(s0_diag),
         _par_stride(s0_diag, P, D), float(dt), _p(A0), _p(b0), _stream())
    return A0, b0


# ---------------------------------------------------------------------------
# the evaluation
# ---------------------------------------------------------------------------
def hyper_gradients(system, D, SigInv, sig_inv_diag, Esde, tr_th, tr_sig):
    """dEsde/dtheta and dEsde/dSigma from the time integrals of the per-step integrands
    (dynamics/sys_lorenz_96_ut.py:250-256, sys_lorenz_63_ut.py:258-264,
    sys_ornstein_uhlenbeck.py:153-156, sys_double_well.py:181-184).  Linear in the
    integrals, so sums of time-window partial integrals give the same result."""
    if D == 1:
        k = 1.0 if system == _lib.SYS_OU else 4.0
        return k * SigInv.reshape(-1) * tr_th, -SigInv.reshape(-1) * Esde
    return sig_inv_diag * tr_th, -0.5 * SigInv @ torch.diag_embed(tr_sig) @ SigInv


def energy_terms(system, A, b, m, S, SigInv, sig_inv_diag, theta, w, dt, want_hyper=True, want_edf=True, packed=False):
    """The ``energy_mode`` contract on device: per-step kernel + time reductions.
    Returns a dict with Esde (P,), Ef, Edf, dEdm, dEdS, dth, dSig, status; with
    ``want_hyper`` also the raw time integrals tr_th, tr_sig the hyper-gradients are
    formed from (what time windows add up)."""
    D = 1 if A.dim() == 2 else A.shape[-1]
    if packed:
        if system != _lib.SYS_L96_UT or want_edf:
            raise ValueError("energy_terms: packed matrices need the Lorenz-96 system and want_edf=False")
        out = esde_l96_packed(A, b, m, S, sig_inv_diag, theta, want_hyper, packed_S=(packed is True))
    else:
        out = esde(system, A, b, m, S, sig_inv_diag, theta, want_edf, want_hyper)
    out["Esde"] = reduce_energy(out["Esde_t"], w, dt)
    if want_hyper:
        out["tr_th"] = reduce_energy(out["dth_t"], w, dt)
        out["tr_sig"] = reduce_energy(out["dSig_t"], w, dt) if D > 1 else None
        out["dth"], out["dSig"] = hyper_gradients(system, D, SigInv, sig_inv_diag, out["Esde"], out["tr_th"],
                                                  out["tr_sig"])
    return out


def _host_energy(prob, A, b, m, S):
    """A user-supplied ``fun(A, b, m, S, sde_struct)`` (the reference's plug-in
    point, core/varFreeEnergy.py:93) is Python code: it runs on the host on copies
    of the device arrays, one problem at a time."""
    outs = []
    for p in range(prob.P):
        r = prob.fun(A[p].cpu().numpy(), b[p].cpu().numpy(), _host_vec(m[p], prob.D), _host_vec(S[p], prob.D),
                     prob.sde_struct)
        outs.append(r)
    stack = lambda k, shp: to_dev(np.stack([np.asarray(o[k], dtype=np.float64).reshape(shp) for o in outs]))  # noqa
    N, D = prob.N, prob.D
    vs, ms = ((N,), (N,)) if D == 1 else ((N, D), (N, D, D))
    return dict(Esde=stack(0, ()), Ef=stack(1, vs), Edf=stack(2, ms), dEdm=stack(3, vs), dEdS=stack(4, ms),
                dth=None, dSig=None, status=torch.zeros((prob.P, N), dtype=torch.int32, device=A.device))


def _host_vec(t, D):
    a = t.cpu().numpy()
    return a.reshape(-1, 1) if D == 1 else a


@dataclass
class State:
    """Everything one free-energy evaluation produces (device tensors)."""
    F: torch.Tensor
    KL0: torch.Tensor
    Eobs: torch.Tensor
    Esde: torch.Tensor
    m: torch.Tensor
    S: torch.Tensor
    lam: Optional[torch.Tensor]
    Psi: Optional[torch.Tensor]
    Ef: torch.Tensor
    Edf: torch.Tensor
    dEdm: torch.Tensor
    dEdS: torch.Tensor
    dEobs_dm: torch.Tensor
    jumpS: torch.Tensor
    status: torch.Tensor
    dth: Optional[torch.Tensor] = None
    dSig: Optional[torch.Tensor] = None


def free_energy(prob, A, b, want_hyper=False, want_edf=True, need_adjoint=True, on_ready=None):
    """core/varFreeEnergy.py:80-111 for a batch: returns a State with F (P,).
    ``want_edf=False`` (Lorenz systems) skips the dense <df/dx> array; ``gradient``
    then lets the assembly kernel derive it from the mean.  ``need_adjoint=False``
    skips the backward scan (lam, Psi are None): enough when only F is wanted.
    ``on_ready(name, tensor)`` is called for m, S, Ef, Edf, lam, Psi as soon as the kernel that
    produces the array has been enqueued (a caller can start copying it out on another stream
    while the rest of the evaluation runs)."""
    ready = on_ready if on_ready is not None else (lambda name, t: None)
    m, S = fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=prob.sym_fwd)
    ready("m", m)
    ready("S", S)
    Eobs, dEobs_dm, jumpS = obs_terms(prob, m, S)
    if prob.system < 0:
        en = _host_energy(prob, A, b, m, S)
    else:
        en = energy_terms(prob.system, A, b, m, S, prob.SigInv, prob.sig_inv_diag, prob.theta, prob.w,
                          prob.dt, want_hyper, want_edf=want_edf or prob.D == 1)
    ready("Ef", en["Ef"])
    if en["Edf"] is not None:
        ready("Edf", en["Edf"])
    sym = prob.sym_jump and prob.system >= 0 and esde_output_is_symmetric(prob.system, prob.D)
    lam = Psi = None
    if need_adjoint:
        lam, Psi = bwd_scan(prob.method, A, en["dEdm"], en["dEdS"], dEobs_dm, prob.dt,
                            obs_flag=prob.obs_flag, jumpS=jumpS, symmetric=sym)
        ready("lam", lam)
        ready("Psi", Psi)
    KL0 = kl0(prob)
    F = KL0 + Eobs + en["Esde"]
    return State(F=F, KL0=KL0, Eobs=Eobs, Esde=en["Esde"], m=m, S=S, lam=lam, Psi=Psi, Ef=en["Ef"],
                 Edf=en["Edf"], dEdm=en["dEdm"], dEdS=en["dEdS"], dEobs_dm=dEobs_dm, jumpS=jumpS,
                 status=en["status"], dth=en.get("dth"), dSig=en.get("dSig"))


def gradient(prob, A, b, st, outer=False, out=None):
    """core/gradL_Ab.py:156-180 for a batch: (gLA, gLb), already times dt."""
    return grad_assemble(prob, A, b, st.m, st.S, st.lam, st.Psi, st.Ef, st.Edf, outer=outer, out=out)


def packed_supported(prob, outer=False):
    """Whether ``free_energy_and_gradient(..., packed=True)`` can run for this problem: the Lorenz-96
    D = 40 Euler case where the backward scan and the gradient are one kernel."""
    if prob.system != _lib.SYS_L96_UT or prob.D != 40 or prob.method != _lib.EULER or not (prob.sym_fwd and prob.sym_jump):
        return False
    sflags = _lib.SCAN_SYMMETRIC | _lib.SCAN_STATE_SYMMETRIC
    gflags = (_lib.GRAD_OUTER if outer else 0) | (_lib.GRAD_SIGINV_DIAG if prob.sig_is_diag else 0)
    return bool(_lib.lib.vgpa_bwd_scan_grad_is_fused(prob.method, prob.P, prob.D, sflags, gflags, 0, 0, prob.system))


def free_energy_and_gradient(prob, A, b, want_hyper=False, want_edf=False, outer=False, out=None, costate=True,
                             packed=False):
    """core/varFreeEnergy.py:80-111 followed by core/gradL_Ab.py:156-180 for a batch, with the
    backward scan and the gradient assembly in one library call (one kernel for the Lorenz-96
    D = 40 Euler case, see ``bwd_scan_grad``).  Returns (State, gLA, gLb).  ``costate=False``: State.Psi
    may hold row 0 only (see ``bwd_scan_grad``); for callers that use F and the gradient alone.

    ``packed=True`` (only where ``packed_supported(prob)``): the symmetric matrices of every time row
    -- State.S, State.dEdS, State.Psi -- are kept as (P, N, 1152) in the packed layout of ``sym_pack``
    instead of (P, N, 40, 40); ``sym_unpack`` expands them.  F, the gradient and every vector are the
    same as without it, bit for bit (same products, same order); the evaluation moves 16 % fewer bytes.
    State.Edf is None.  ``packed="dEdS"``: only State.dEdS is packed -- it is an intermediate between the
    energy kernel and the backward scan, not one of the arrays the reference publishes -- and S, Psi are
    the dense arrays of the contract."""
    if packed and (prob.system < 0 or want_edf or not packed_supported(prob, outer)):
        raise ValueError("free_energy_and_gradient: packed=True needs the Lorenz-96 D = 40 Euler problem with diagonal "
                         "noise, more than 160 problems, and want_edf=False (see packed_supported)")
    if prob.system < 0:
        st = free_energy(prob, A, b, want_hyper=want_hyper, want_edf=True)
        gLA, gLb = gradient(prob, A, b, st, outer=outer, out=out)
        return st, gLA, gLb
    if packed not in (False, True, "dEdS"):
        raise ValueError("free_energy_and_gradient: packed is False, True or 'dEdS'")
    m, S = fwd_scan(prob.method, A, b, prob.m0, prob.S0, prob.Sigma, prob.dt, symmetric=prob.sym_fwd, packed=(packed is True))
    Eobs, dEobs_dm, jumpS = obs_terms(prob, m, S, packed=(packed is True))
    implicit_ok = prob.D > 1 and prob.system in (_lib.SYS_L96_UT, _lib.SYS_L63, _lib.SYS_L63_UT)
    en = energy_terms(prob.system, A, b, m, S, prob.SigInv, prob.sig_inv_diag, prob.theta, prob.w, prob.dt,
                      want_hyper, want_edf=want_edf or not implicit_ok, packed=packed)
    sym = prob.sym_jump and esde_output_is_symmetric(prob.system, prob.D)
    lam, Psi, gLA, gLb = bwd_scan_grad(prob, A, b, m, S, en["dEdm"], en["dEdS"], dEobs_dm, en["Ef"], en["Edf"],
                                       jumpS=jumpS, symmetric=sym, state_symmetric=prob.sym_fwd, outer=outer, out=out,
                                       costate=costate, packed=packed)
    KL0 = kl0(prob)
    F = KL0 + Eobs + en["Esde"]
    st = State(F=F, KL0=KL0, Eobs=Eobs, Esde=en["Esde"], m=m, S=S, lam=lam, Psi=Psi, Ef=en["Ef"], Edf=en["Edf"],
               dEdm=en["dEdm"], dEdS=en["dEdS"], dEobs_dm=dEobs_dm, jumpS=jumpS, status=en["status"],
               dth=en.get("dth"), dSig=en.get("dSig"))
    return st, gLA, gLb


class CapturedEvaluation:
    """One free-energy + gradient evaluation recorded as a HIP graph.

    For small problems (a single 1-D or Lorenz-63 smoothing run) the dozen kernel launches and the
    tensor bookkeeping around them cost more than the kernels; replaying a graph issues them with
    one call.  ``A`` and ``b`` are the buffers the graph reads: write new parameter values INTO them
    (``A.copy_(...)``) and call ``replay()``; ``state``, ``gLA``, ``gLb`` are the buffers it writes,
    the same objects after every replay.  Everything the evaluation allocates comes from the graph's
    private pool, so nothing is allocated at replay.  The per-step status is not checked here: call
    ``check_status(self.state.status)`` when a result is consumed.
    """

    def __init__(self, prob, A, b, want_hyper=False, outer=False):
        if prob.system < 0:
            raise ValueError("a host plug-in energy function cannot be captured")
        self.prob, self.A, self.b = prob, A, b
        kl0(prob)                                        # cached on the problem: outside the graph
        # one eager run first: lazy initialisation (kernel attributes, library handles) must not
        # happen inside a capture
        free_energy_and_gradient(prob, A, b, want_hyper=want_hyper, outer=outer)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.state, self.gLA, self.gLb = free_energy_and_gradient(prob, A, b, want_hyper=want_hyper, outer=outer)

    def replay(self):
        self.graph.replay()
        return self.state, self.gLA, self.gLb


def check_status(status):
    """Raise like the reference when a covariance lost positive definiteness
    (auxiliary/numerical.py:199-202 lets LinAlgError escape when the diagonal
    fallback fails too)."""
    if status is None or status.numel() == 0:
        return
    bad = 0
    for bit in (_lib.ST_CHOL_FAILED, _lib.ST_UNSUPPORTED):
        if bool((status & bit).any().item()):
            bad |= bit
    if bad & _lib.ST_CHOL_FAILED:
        raise np.linalg.LinAlgError("Matrix is not positive definite")
    if bad & _lib.ST_UNSUPPORTED:
        raise np.linalg.LinAlgError(
            "S(t) is not positive definite; the diagonal sigma-point fallback is not available on the "
            "device for this system")
