"""The vgpa_scg_* passes (the vector algebra of the batched SCG driver) against their
torch statement, through the C-ABI.

Element-wise results must be bit-identical (the kernels round products and sums
separately, as torch does); the per-problem sums are formed in another order than
torch's, so they are compared to 1e-13 relative to sum|terms|.  Maxima are exact.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops(P, shape_A, shape_b):
    from vgpa_amd.auxiliary.optimize import _DeviceParamOps, _TorchParamOps
    dev = torch.device("cuda", 0)
    A0 = torch.zeros((P,) + shape_A, dtype=torch.float64, device=dev)
    b0 = torch.zeros((P,) + shape_b, dtype=torch.float64, device=dev)
    return _DeviceParamOps(A0, b0), _TorchParamOps(A0, b0)


def _vec(ops, gen):
    return torch.randn((ops.P * ops.n,), dtype=torch.float64, device=ops.dev, generator=gen)


def _close(a, b, scale):
    assert torch.all((a - b).abs() <= 1e-13 * scale), (a, b)


# (N, D) pairs: D = 1 with an odd N (scalar loads), small and large even sizes, and a row
# longer than one chunk.
@pytest.mark.parametrize("P,shape_A,shape_b", [(5, (2001,), (2001,)), (7, (33, 3, 3), (33, 3)),
                                               (3, (1000, 40, 40), (1000, 40)), (4, (1,), (1,)),
                                               (9, (64, 2, 2), (64, 2))])
def test_passes_match_torch_statement(P, shape_A, shape_b):
    hip, ref = _ops(P, shape_A, shape_b)
    gen = torch.Generator(device=hip.dev).manual_seed(17 + P)
    x, d, g, gt, xt = (_vec(hip, gen) for _ in range(5))
    coef = torch.randn((P,), dtype=torch.float64, device=hip.dev, generator=gen)
    n = hip.n

    # dots, with and without modes
    mode = (torch.arange(P, device=hip.dev) % 3).to(torch.int32)
    for m in (None, mode):
        a, b = hip.dots(d, g, m), ref.dots(d, g, m)
        _close(a, b, n * 10.0)
        if m is not None:
            assert torch.all(a[m == 0] == 0)

    # step: modes 0 (row untouched), 1, 2 (direction replaced by -g)
    y1, y2 = torch.full_like(x, 7.0), torch.full_like(x, 7.0)
    d1, d2 = d.clone(), d.clone()
    hip.step(x, d1, g, coef, mode, y1)
    ref.step(x, d2, g, coef, mode, y2)
    assert torch.equal(y1, y2) and torch.equal(d1, d2)
    for rows_y, rows_d, rows_d0 in zip(hip.rows(y1), hip.rows(d1), hip.rows(d)):
        assert torch.all(rows_y[mode == 0] == 7.0)
        assert torch.equal(rows_d[mode != 2], rows_d0[mode != 2])

    # curvature and trial statistics
    pm = (mode != 1).to(torch.int32)
    _close(hip.curvature(d, gt, g, pm), ref.curvature(d, gt, g, pm), n * 10.0)
    a, b = hip.trial_stats(d, g, gt, coef, pm), ref.trial_stats(d, g, gt, coef, pm)
    assert torch.equal(a[:, 0], b[:, 0])
    _close(a[:, 1:], b[:, 1:], n * 10.0)

    # update: all four combinations of (moved, restart); then the modes of a caller that exchanges
    # its arrays instead of copying: 4 (new direction from g as it stands), 6 (restart), 8 (copy only)
    for um in ((torch.arange(P, device=hip.dev) % 4).to(torch.int32),
               torch.tensor([0, 4, 6, 8], device=hip.dev, dtype=torch.int32)[torch.arange(P, device=hip.dev) % 4]):
        s1 = [t.clone() for t in (x, g, d)]
        s2 = [t.clone() for t in (x, g, d)]
        a = hip.update(s1[0], xt, s1[1], gt, s1[2], coef, um)
        b = ref.update(s2[0], xt, s2[1], gt, s2[2], coef, um)
        for t1, t2 in zip(s1, s2):
            assert torch.equal(t1, t2)
        _close(a, b, n * 30.0)
        assert torch.all(a[(um == 0) | (um == 8)] == 0)
        for rx, rg, rd, x0, g0, d0, rxt, rgt in zip(hip.rows(s1[0]), hip.rows(s1[1]), hip.rows(s1[2]), hip.rows(x),
                                                     hip.rows(g), hip.rows(d), hip.rows(xt), hip.rows(gt)):
            keep = (um == 4) | (um == 6)            # x and g stay, d changes
            assert torch.equal(rx[keep], x0[keep]) and torch.equal(rg[keep], g0[keep])
            copied = um == 8                        # x and g are copied, d stays
            assert torch.equal(rx[copied], rxt[copied]) and torch.equal(rg[copied], rgt[copied])
            assert torch.equal(rd[copied], d0[copied])
    # repeated call on the same data: same bits
    assert torch.equal(hip.dots(d, g), hip.dots(d, g))


def test_bad_arguments_are_rejected():
    from vgpa_amd import engine
    from vgpa_amd._lib import VgpaError
    hip, _ = _ops(2, (8,), (8,))
    v = hip.new()
    out = torch.empty((2, 4), dtype=torch.float64, device=hip.dev)
    with pytest.raises((VgpaError, ValueError)):
        engine.call("vgpa_scg_dots", 2, 8, 8, v.data_ptr(), v.data_ptr(), None, 0, hip.work.data_ptr(), out.data_ptr(),
                    engine._stream())
    with pytest.raises((VgpaError, ValueError)):
        engine.call("vgpa_scg_dots", 2, 0, 8, v.data_ptr(), v.data_ptr(), None, 1, hip.work.data_ptr(), out.data_ptr(),
                    engine._stream())
