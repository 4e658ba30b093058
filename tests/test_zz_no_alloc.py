"""The observation terms through the entry point that allocates nothing
(vgpa_obs_terms_static, include/vgpa.h): same bits as vgpa_obs_terms with a caller-provided
work array, on two calls in a row that share the library's fixed scratch array, the second
with a smaller input; sizes beyond the array are refused."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs(P, N, D, M, seed):
    import torch
    gen = torch.Generator(device="cuda").manual_seed(seed)
    dev = torch.device("cuda", 0)
    rnd = lambda *s: torch.randn(s, dtype=torch.float64, device=dev, generator=gen)   # noqa: E731
    m, S = rnd(P, N, D), rnd(P, N, D, D)
    obs_idx = torch.from_numpy(np.sort(np.random.default_rng(seed).choice(N, M, replace=False)).astype(np.int32)).to(dev)
    obsY = rnd(P, M, D)
    H = torch.eye(D, dtype=torch.float64, device=dev) + 0.1 * rnd(D, D)
    Q = torch.linalg.inv(torch.linalg.cholesky(torch.eye(D, dtype=torch.float64, device=dev) * 0.7 + 0.05 * (lambda X: X @ X.T)(rnd(D, D))))
    return m, S, obs_idx, obsY, H, (Q @ Q.T).contiguous(), Q.contiguous()


def _both(P, N, D, M, seed):
    import torch
    from vgpa_amd import _lib
    m, S, obs_idx, obsY, H, Ri, Qi = _inputs(P, N, D, M, seed)
    dev = m.device
    stream = torch.cuda.current_stream().cuda_stream
    outs = []
    for static in (False, True):
        Eobs = torch.full((P,), float("nan"), dtype=torch.float64, device=dev)
        dm = torch.full((P, N, D), float("nan"), dtype=torch.float64, device=dev)
        jumpS = torch.full((D, D), float("nan"), dtype=torch.float64, device=dev)
        common = (P, N, D, M, D, m.data_ptr(), S.data_ptr(), obs_idx.data_ptr(), obsY.data_ptr(), M * D,
                  H.data_ptr(), Ri.data_ptr(), Qi.data_ptr(), 1.25)
        if static:
            _lib.call("vgpa_obs_terms_static", *common, Eobs.data_ptr(), dm.data_ptr(), jumpS.data_ptr(), stream)
        else:
            work = torch.full((P, M), float("nan"), dtype=torch.float64, device=dev)
            _lib.call("vgpa_obs_terms", *common, work.data_ptr(), Eobs.data_ptr(), dm.data_ptr(), jumpS.data_ptr(), stream)
        torch.cuda.synchronize()
        outs.append((Eobs, dm, jumpS))
    return outs


def test_static_scratch_entry_point_matches_the_allocating_one():
    import torch
    # first call: large; second call: smaller input in the same (stale) scratch array
    for (P, N, D, M, seed) in ((37, 64, 40, 23, 1), (5, 31, 6, 7, 2)):
        (E0, dm0, j0), (E1, dm1, j1) = _both(P, N, D, M, seed)
        assert torch.isfinite(E0).all() and torch.isfinite(E1).all()
        assert torch.equal(E0, E1) and torch.equal(dm0, dm1) and torch.equal(j0, j1)
    # the 1-D path needs no scratch and goes through unchanged
    from vgpa_amd import _lib
    dev = torch.device("cuda", 0)
    P, N, M = 9, 50, 6
    m = torch.randn((P, N), dtype=torch.float64, device=dev)
    S = torch.rand((P, N), dtype=torch.float64, device=dev) + 0.1
    idx = torch.tensor([3, 4, 10, 20, 33, 49], dtype=torch.int32, device=dev)
    y = torch.randn((M, 1), dtype=torch.float64, device=dev)
    one = torch.ones(1, dtype=torch.float64, device=dev)
    res = []
    for name in ("vgpa_obs_terms", "vgpa_obs_terms_static"):
        Eobs = torch.empty((P,), dtype=torch.float64, device=dev)
        dm = torch.empty((P, N), dtype=torch.float64, device=dev)
        js = torch.empty((1,), dtype=torch.float64, device=dev)
        args = [P, N, 1, M, 1, m.data_ptr(), S.data_ptr(), idx.data_ptr(), y.data_ptr(), 0, one.data_ptr(),
                one.data_ptr(), None, 0.5]
        if name == "vgpa_obs_terms":
            args.append(None)
        _lib.call(name, *args, Eobs.data_ptr(), dm.data_ptr(), js.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        res.append((Eobs, dm, js))
    assert all(torch.equal(a, b) for a, b in zip(*res))


def test_static_scratch_entry_point_refuses_oversize():
    import torch
    from vgpa_amd import _lib
    dev = torch.device("cuda", 0)
    D, N = 2, 4
    M = 4
    P = _lib.OBS_STATIC_MAX // M + 1
    z = torch.zeros(8, dtype=torch.float64, device=dev)
    zi = torch.zeros(8, dtype=torch.int32, device=dev)
    with pytest.raises(_lib.VgpaError, match="exceeds the fixed scratch"):
        _lib.call("vgpa_obs_terms_static", P, N, D, M, D, z.data_ptr(), z.data_ptr(), zi.data_ptr(), z.data_ptr(), 0,
                  z.data_ptr(), z.data_ptr(), z.data_ptr(), 0.0, z.data_ptr(), z.data_ptr(), z.data_ptr(), None)


def test_static_scratch_entry_point_under_graph_capture():
    """What the entry point is for: it can be recorded into a HIP graph (a capture fails on an allocation
    or a synchronising call) and replayed on inputs changed in place, with the eager call's bits; device
    memory in use does not change across a call."""
    import torch
    from vgpa_amd import _lib
    P, N, D, M = 11, 40, 5, 9
    m, S, obs_idx, obsY, H, Ri, Qi = _inputs(P, N, D, M, 5)
    dev = m.device
    Eobs = torch.empty((P,), dtype=torch.float64, device=dev)
    dm = torch.empty((P, N, D), dtype=torch.float64, device=dev)
    jumpS = torch.empty((D, D), dtype=torch.float64, device=dev)

    def run():
        _lib.call("vgpa_obs_terms_static", P, N, D, M, D, m.data_ptr(), S.data_ptr(), obs_idx.data_ptr(), obsY.data_ptr(),
                  M * D, H.data_ptr(), Ri.data_ptr(), Qi.data_ptr(), 1.25, Eobs.data_ptr(), dm.data_ptr(), jumpS.data_ptr(),
                  torch.cuda.current_stream().cuda_stream)

    run()                                   # the first call on the device resolves the array's address
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    run()
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        run()
    for k in range(2):
        m.add_(0.25 * (k + 1))
        S.mul_(1.0 + 0.1 * (k + 1))
        graph.replay()
        torch.cuda.synchronize()
        got = (Eobs.clone(), dm.clone(), jumpS.clone())
        Eobs.fill_(float("nan")), dm.fill_(float("nan")), jumpS.fill_(float("nan"))
        run()
        torch.cuda.synchronize()
        assert torch.isfinite(Eobs).all()
        assert torch.equal(got[0], Eobs) and torch.equal(got[1], dm) and torch.equal(got[2], jumpS)
