"""Kernel-level parity on a real MI355X: every C-ABI op against the CPU oracle's op
(stock PyTorch-CPU fp64 of the same reference call site).  Tolerances are fp32-roundoff
scaled by the reduction length."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import scvae_oracle as O


@pytest.fixture(scope="module")
def ops():
    from scrubvae_amd import ops as _ops
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return _ops


def to_nlc(x_ncl, ld=None):
    """[B,C,L] cpu -> [B*L, ld] cuda float32, zero padded."""
    B, Cc, L = x_ncl.shape
    ld = ld or (Cc + 15) // 16 * 16
    out = torch.zeros(B * L, ld, dtype=torch.float32)
    out[:, :Cc] = x_ncl.permute(0, 2, 1).reshape(B * L, Cc).float()
    return out.cuda()


def from_nlc(y, B, L, Cc):
    return y.cpu()[:, :Cc].reshape(B, L, Cc).permute(0, 2, 1).double()


def relerr(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


CONV_CASES = [
    # B, L, Cin, Cout, k, stride, pad, transposed
    (3, 64, 111, 64, 7, 1, 3, False),    # conv_in, ragged Cin
    (5, 64, 64, 64, 5, 2, 2, False),     # res0.residual.0
    (5, 32, 64, 128, 5, 1, 2, False),    # res0.residual.3
    (4, 16, 256, 512, 5, 2, 2, False),   # res2.skip (N=512 -> 4 column tiles)
    (2, 5, 40, 24, 5, 2, 2, False),      # odd length, tiny ragged channels
    (9, 4, 128, 64, 5, 1, 2, True),      # dec residual.0 (ConvT s1)
    (9, 4, 64, 48, 5, 2, 2, True),       # dec residual.3 (ConvT s2: 4 -> 7, parity split)
    (3, 13, 32, 32, 5, 2, 2, True),      # 13 -> 25
    (7, 8, 128, 64, 6, 1, 2, False),     # skip conv after upsample (k+1, even kernel)
    (2, 49, 64, 141, 22, 1, 3, True),    # conv_out: N=141 -> padded 144, 3 x BN=64
    (200, 1, 512, 64, 1, 1, 0, False),   # Linear as 1x1 conv, M=200
    (33, 1, 40, 4096, 1, 1, 0, False),   # fc_in-like, wide N
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(ops, case):
    B, L, Cin, Cout, k, s, p, tr = case
    g = torch.Generator().manual_seed(sum(case[:7]))
    x = torch.randn(B, Cin, L, generator=g, dtype=torch.float64)
    w = torch.randn(*((Cin, Cout, k) if tr else (Cout, Cin, k)), generator=g, dtype=torch.float64) / math.sqrt(Cin * k)
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    x.requires_grad_(True); w.requires_grad_(True); b.requires_grad_(True)
    y = (F.conv_transpose1d if tr else F.conv1d)(x, w, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    Lout = y.shape[-1]

    cv = ops.Conv(B, L, Cin, Cout, k, s, p, 1, tr)
    assert cv.l_out == Lout
    xd = to_nlc(x.detach())
    wd = ops.conv_weight_to_tio(w.detach().float(), tr).cuda().contiguous()
    bd = torch.zeros(cv.c_out_p); bd[:Cout] = b.detach().float(); bd = bd.cuda()
    yd = torch.full((B * Lout, cv.c_out_p), float("nan"), device="cuda")
    cv.fwd(xd, wd, bd, yd)
    torch.cuda.synchronize()
    tol = 2e-6 * math.sqrt(Cin * k) + 2e-6
    assert relerr(from_nlc(yd, B, Lout, Cout), y.detach()) < tol
    assert float(yd[:, Cout:].abs().max() if cv.c_out_p > Cout else 0) == 0.0  # pads stay zero
    # accumulate
    cv.fwd(xd, wd, bd, yd, accumulate=True)
    assert relerr(from_nlc(yd, B, Lout, Cout), 2 * y.detach()) < tol

    dyd = to_nlc(dy)
    dxd = torch.full((B * L, cv.c_in_p), float("nan"), device="cuda")
    cv.dgrad(dyd, wd, dxd)
    assert relerr(from_nlc(dxd, B, L, Cin), x.grad) < 2e-6 * math.sqrt(Cout * k) + 2e-6

    ws = torch.empty(cv.wgrad_workspace_bytes() // 4 + 4, device="cuda")
    dwd = torch.full(cv.weight_shape, float("nan"), device="cuda")
    dbd = torch.full((cv.c_out_p,), float("nan"), device="cuda")
    cv.wgrad(xd, dyd, dwd, dbd, ws)
    dw = ops.conv_weight_from_tio(dwd.cpu(), Cin, Cout, tr)
    assert relerr(dw, w.grad) < 2e-6 * math.sqrt(B * Lout) + 2e-6
    assert relerr(dbd.cpu()[:Cout], b.grad) < 2e-6 * math.sqrt(B * Lout) + 2e-6
    cv.wgrad(xd, dyd, dwd, dbd, ws, accumulate=True)
    assert relerr(ops.conv_weight_from_tio(dwd.cpu(), Cin, Cout, tr), 2 * w.grad) < 2e-6 * math.sqrt(B * Lout) + 2e-6


def test_conv_rejects_bad_shapes(ops):
    cv = ops.Conv(2, 8, 16, 16, 5, 2, 2)
    cv.desc.l_out = 99
    x = torch.zeros(16, 16, device="cuda")
    with pytest.raises(RuntimeError, match="l_out"):
        cv.fwd(x, x, None, x)


def test_pack_input(ops):
    B, W, J = 3, 8, 18
    x6d = torch.randn(B, W, J, 6)
    root = torch.rand(B, W, 3) * 4 - 1
    arena = [-1.0, -2.0, -0.5, 3.0, 2.0, 1.5]
    out = torch.full((B * W, 112), float("nan"), device="cuda")
    ops.pack_input(x6d.cuda(), root.cuda(), arena, out, J)
    a = torch.tensor(arena).reshape(2, 3)
    ref = torch.cat([x6d.reshape(B * W, -1), O.normalize_root(root, a).reshape(B * W, 3)], -1)
    assert relerr(out.cpu()[:, :111], ref) < 1e-6
    assert float(out[:, 111:].abs().max()) == 0.0


@pytest.mark.parametrize("rows,Cc", [(700, 48), (4096, 144), (37, 16)])
def test_bn_prelu_fwd_bwd(ops, rows, Cc):
    """Train-mode BatchNorm1d(eps=1e-4)+PReLU forward/backward vs torch CPU fp64."""
    g = torch.Generator().manual_seed(rows)
    x = (torch.randn(rows, Cc, generator=g, dtype=torch.float64) * 2 + 0.7).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(Cc, generator=g, dtype=torch.float64)).requires_grad_(True)
    beta = (0.1 * torch.randn(Cc, generator=g, dtype=torch.float64)).requires_grad_(True)
    alpha = torch.tensor([0.25], dtype=torch.float64, requires_grad=True)
    rm, rv = torch.zeros(Cc, dtype=torch.float64), torch.ones(Cc, dtype=torch.float64)
    # torch wants [N,C]; BatchNorm over rows
    y = F.prelu(F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-4), alpha)
    dy = torch.randn(rows, Cc, generator=g, dtype=torch.float64)
    y.backward(dy)

    dev = lambda t: t.detach().float().cuda().contiguous()
    xd, gd, bd, ad = dev(x), dev(gamma), dev(beta), dev(alpha)
    nch = ops.bn_chunks(rows)
    part = torch.empty(nch, 2, Cc, device="cuda")
    sums = torch.empty(2, Cc, device="cuda")
    rmd, rvd = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    mean, rstd, scale, shift = (torch.empty(Cc, device="cuda") for _ in range(4))
    ops.bn_stats_partial(xd, rows, Cc, Cc, part)
    ops.bn_reduce_partials(part, nch, Cc, sums)
    ops.bn_finalize(sums, rows, Cc, gd, bd, 1e-4, 0.1, rmd, rvd, mean, rstd, scale, shift)
    yd = torch.empty_like(xd)
    ops.affine_prelu_fwd(xd, scale, shift, ad, yd, rows, Cc, Cc)
    assert relerr(yd.cpu(), y.detach()) < 5e-6
    assert relerr(rmd.cpu(), rm) < 5e-6 and relerr(rvd.cpu(), rv) < 5e-6

    dyd = dev(dy)
    dap = torch.empty(2 * nch * ((Cc + 63) // 64), device="cuda")
    ops.affine_prelu_bwd_partial(dyd, xd, scale, shift, mean, rstd, ad, rows, Cc, Cc, part, dap)
    ops.bn_reduce_partials(part, nch, Cc, sums)
    dxd = torch.empty_like(xd)
    dg, db, da = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda"), torch.zeros(1, device="cuda")
    ops.affine_prelu_bwd_apply(dyd, xd, scale, shift, mean, rstd, gd, ad, sums, rows, dxd, rows, Cc, Cc, dg, db, da, dap, dap.numel(), False)
    assert relerr(dxd.cpu(), x.grad) < 2e-5
    assert relerr(dg.cpu(), gamma.grad) < 2e-5 and relerr(db.cpu(), beta.grad) < 2e-5
    assert relerr(da.cpu(), alpha.grad) < 2e-5


@pytest.mark.parametrize("rows,Cc,accumulate", [(5000, 64, False), (70000, 256, True), (37, 16, False), (300000, 1024, False)])
def test_backward_apply_leaves_the_column_sums_of_dx(ops, rows, Cc, accumulate):
    """The bias gradient of a conv whose output feeds an activation stage = column sums of that stage's dx (autograd of the
    reference's `bias=True` convs): ops.affine_prelu_bwd_apply leaves per-workgroup partials, ColsumBatch.add_partials reduces
    them -- against the sums of the dx the same launch wrote (bare PReLU stage: the sums are far from zero)."""
    g = torch.Generator().manual_seed(rows + Cc)
    x = torch.randn(rows, Cc, generator=g).cuda()
    dy = (torch.randn(rows, Cc, generator=g) + 0.3).cuda()
    alpha = torch.tensor([0.25], device="cuda")
    n = ops.affine_prelu_colsum_rows(rows, Cc)
    assert 0 < n <= 1024
    part = torch.full((n, Cc), float("nan"), device="cuda")
    dx = torch.empty_like(x)
    dap = torch.zeros(4, device="cuda")
    ops.affine_prelu_bwd_apply(dy, x, None, None, None, None, None, alpha, None, 1.0, dx, rows, Cc, Cc, None, None, None, dap, 0,
                               False, colsum_part=part)
    ref_dx = torch.where(x > 0, dy, 0.25 * dy)
    assert torch.equal(dx, ref_dx)
    out = torch.full((Cc,), 3.0, device="cuda")
    batch = ops.ColsumBatch()
    batch.add_partials(part, n, Cc, out)
    batch.flush(lambda nbytes: torch.empty(nbytes // 4 + 16, device="cuda"), accumulate=accumulate)
    ref = dx.double().sum(0) + (3.0 if accumulate else 0.0)
    assert float((out.double() - ref).abs().max()) < 2e-6 * float(dx.double().abs().sum(0).max())
    assert ops.affine_prelu_colsum_rows(100, 48) == 0 and ops.affine_prelu_colsum_rows(100, 144) == 0  # C / 4 not a power of two


@pytest.mark.parametrize("rows,Cc", [(700, 48), (37, 16)])
def test_bn_tanh_fwd_bwd(ops, rows, Cc):
    """model.activation == "tanh": the same kernels with a NULL slope pointer = train-mode BatchNorm + tanh, vs torch fp64."""
    g = torch.Generator().manual_seed(rows + 1)
    x = (torch.randn(rows, Cc, generator=g, dtype=torch.float64) * 2 + 0.7).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(Cc, generator=g, dtype=torch.float64)).requires_grad_(True)
    beta = (0.1 * torch.randn(Cc, generator=g, dtype=torch.float64)).requires_grad_(True)
    rm, rv = torch.zeros(Cc, dtype=torch.float64), torch.ones(Cc, dtype=torch.float64)
    y = torch.tanh(F.batch_norm(x, rm, rv, gamma, beta, True, 0.1, 1e-4))
    dy = torch.randn(rows, Cc, generator=g, dtype=torch.float64)
    y.backward(dy)
    dev = lambda t: t.detach().float().cuda().contiguous()
    xd, gd, bd, dyd = dev(x), dev(gamma), dev(beta), dev(dy)
    nch = ops.bn_chunks(rows)
    part, sums = torch.empty(nch, 2, Cc, device="cuda"), torch.empty(2, Cc, device="cuda")
    rmd, rvd = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    mean, rstd, scale, shift = (torch.empty(Cc, device="cuda") for _ in range(4))
    ops.bn_stats_partial(xd, rows, Cc, Cc, part)
    ops.bn_reduce_partials(part, nch, Cc, sums)
    ops.bn_finalize(sums, rows, Cc, gd, bd, 1e-4, 0.1, rmd, rvd, mean, rstd, scale, shift)
    yd = torch.empty_like(xd)
    ops.affine_prelu_fwd(xd, scale, shift, None, yd, rows, Cc, Cc)
    assert relerr(yd.cpu(), y.detach()) < 5e-6
    dap = torch.empty(2 * nch * ((Cc + 63) // 64), device="cuda")
    ops.affine_prelu_bwd_partial(dyd, xd, scale, shift, mean, rstd, None, rows, Cc, Cc, part, dap)
    ops.bn_reduce_partials(part, nch, Cc, sums)
    dxd = torch.empty_like(xd)
    dg, db = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    ops.affine_prelu_bwd_apply(dyd, xd, scale, shift, mean, rstd, gd, None, sums, rows, dxd, rows, Cc, Cc, dg, db, None, dap, dap.numel(), False)
    assert relerr(dxd.cpu(), x.grad) < 2e-5
    assert relerr(dg.cpu(), gamma.grad) < 2e-5 and relerr(db.cpu(), beta.grad) < 2e-5
    assert float(dap.abs().max()) == 0.0  # no slope, no slope-gradient partials


def test_bare_prelu_bwd(ops):
    rows, Cc = 300, 64
    torch.manual_seed(3)
    x = torch.randn(rows, Cc, dtype=torch.float64, requires_grad=True)
    alpha = torch.tensor([0.3], dtype=torch.float64, requires_grad=True)
    y = F.prelu(x, alpha)
    dy = torch.randn(rows, Cc, dtype=torch.float64)
    y.backward(dy)
    dev = lambda t: t.detach().float().cuda().contiguous()
    xd, ad, dyd = dev(x), dev(alpha), dev(dy)
    yd = torch.empty_like(xd)
    ops.affine_prelu_fwd(xd, None, None, ad, yd, rows, Cc, Cc)
    assert relerr(yd.cpu(), y.detach()) < 1e-6
    nch = ops.bn_chunks(rows)
    part = torch.empty(nch, 2, Cc, device="cuda")
    dap = torch.empty(2 * nch * ((Cc + 63) // 64), device="cuda")
    ops.affine_prelu_bwd_partial(dyd, xd, None, None, None, None, ad, rows, Cc, Cc, part, dap)
    dxd = torch.empty_like(xd)
    da = torch.zeros(1, device="cuda")
    ops.affine_prelu_bwd_apply(dyd, xd, None, None, None, None, None, ad, None, 1.0, dxd, rows, Cc, Cc, None, None, da, dap, dap.numel(), False)
    assert relerr(dxd.cpu(), x.grad) < 1e-6 and relerr(da.cpu(), alpha.grad) < 1e-4


@pytest.mark.parametrize("B,L,Cc", [(3, 4, 32), (2, 7, 16), (5, 1, 16)])
def test_upsample2(ops, B, L, Cc):
    x = torch.randn(B, Cc, L, dtype=torch.float64, requires_grad=True)
    y = F.interpolate(x, scale_factor=2, mode="linear", align_corners=False)
    dy = torch.randn_like(y)
    y.backward(dy)
    xd = to_nlc(x.detach())
    yd = torch.empty(B * 2 * L, Cc, device="cuda")
    ops.upsample2_fwd(xd, yd, B, L, Cc, Cc)
    assert relerr(from_nlc(yd, B, 2 * L, Cc), y.detach()) < 1e-6
    dxd = torch.empty(B * L, Cc, device="cuda")
    ops.upsample2_bwd(to_nlc(dy), dxd, B, L, Cc, Cc)
    assert relerr(from_nlc(dxd, B, L, Cc), x.grad) < 1e-6


@pytest.mark.parametrize("n,nrhs,batch,with_diag", [(32, 3, 2, True), (33, 1, 2, True), (5, 7, 1, False), (64, 64, 3, False), (1, 1, 1, True)])
def test_small_solve_vs_fp64(ops, n, nrhs, batch, with_diag):
    """Batched LU solve of the streaming scrubbers' normal equations (svae_small_solve; reference: torch.linalg.solve in
    MovingAvgLeastSquares.forward, disentangle.py:466-486, and direct_lsq_loss, losses.py:173-179) against fp64: Gram matrices of
    random designs (condition numbers of 1e2-1e4, like exponentially-forgotten latent covariances) plus a non-symmetric system
    that needs the row pivoting."""
    g = torch.Generator().manual_seed(100 * n + nrhs)
    X = torch.randn(batch, 4 * n + 8, n, generator=g, dtype=torch.float64)
    A = X.transpose(1, 2) @ X
    A[-1] = torch.randn(n, n, generator=g, dtype=torch.float64) + torch.diag(torch.zeros(n, dtype=torch.float64))  # general matrix
    if n > 1:
        A[-1][0, 0] = 0.0  # the first pivot must come from another row
    B = torch.randn(batch, n, nrhs, generator=g, dtype=torch.float64)
    d = torch.rand(n, generator=g, dtype=torch.float64) if with_diag else None
    ref = torch.linalg.solve(A + (torch.diag(d) if with_diag else 0), B)
    out = ops.small_solve(A.float().cuda(), B.float().cuda(), None if d is None else d.float().cuda())
    f32 = torch.linalg.solve((A + (torch.diag(d) if with_diag else 0)).float(), B.float()).double()  # the reference's own fp32 arithmetic
    e_hip = float((out.double().cpu() - ref).abs().max() / ref.abs().max())
    e_f32 = float((f32 - ref).abs().max() / ref.abs().max())
    assert e_hip < max(20 * e_f32, 1e-5), (e_hip, e_f32)
    one = ops.small_solve(A[0].float().cuda(), B[0].float().cuda(), None if d is None else d.float().cuda())
    assert torch.equal(one, out[0])  # the unbatched call is the same kernel on one system


@pytest.mark.parametrize("accumulate", [False, True])
def test_colsum_batched_with_shared_matrices(ops, accumulate):
    """Bias gradients (column sums of every conv's dY, autograd of the reference's `bias=True` convs) in two launches for a list of
    tasks; a matrix queued twice -- the second conv and the skip conv of a residual block share their dY -- is read once, both
    outputs get the sums."""
    g = torch.Generator().manual_seed(3)
    A = torch.randn(1500, 48, generator=g).cuda()
    Bm = torch.randn(700, 80, generator=g).cuda()[:, :64]  # ld 80, 64 columns
    Cc = torch.randn(513, 16, generator=g).cuda()
    outs = [torch.full((48,), 0.5, device="cuda"), torch.full((64,), 0.5, device="cuda"), torch.full((48,), -1.0, device="cuda"),
            torch.full((16,), 0.5, device="cuda"), torch.full((64,), 2.0, device="cuda")]
    init = [o.clone() for o in outs]
    batch = ops.ColsumBatch()
    for x, o in zip((A, Bm, A, Cc, Bm), outs):
        batch.add(x, x.shape[0], x.shape[1], x.stride(0), o)
    ws = {}
    def get_ws(n):
        ws["t"] = torch.empty(n // 4 + 16, device="cuda")
        return ws["t"]
    batch.flush(get_ws, accumulate=accumulate)
    for x, o, o0 in zip((A, Bm, A, Cc, Bm), outs, init):
        ref = x.double().sum(0) + (o0.double() if accumulate else 0)
        assert float((o.double() - ref).abs().max()) < 1e-4 * float(ref.abs().max() + 1)
    assert torch.equal(outs[0] - (init[0] if accumulate else 0), outs[2] - (init[2] if accumulate else 0)) or accumulate


def test_small_inverse_autograd_vs_fp64(ops):
    """LinearProjection's (W W^T)^-1 (reference disentangle.py:717-734: torch.linalg.solve(nrm, x.T)) through ops.small_inverse_autograd:
    the value and the gradient with respect to W of the projected latent against fp64 autograd."""
    g = torch.Generator().manual_seed(5)
    w = torch.randn(3, 16, generator=g, dtype=torch.float64, requires_grad=True)
    z = torch.randn(50, 16, generator=g, dtype=torch.float64, requires_grad=True)
    c = torch.randn(50, 16, generator=g, dtype=torch.float64)
    x = z @ w.T
    ref = z - torch.linalg.solve(w @ w.T, x.T).T @ w
    (ref * c).sum().backward()
    wd, zd = w.detach().float().cuda().requires_grad_(True), z.detach().float().cuda().requires_grad_(True)
    out = zd - ((zd @ wd.T) @ ops.small_inverse_autograd(wd @ wd.T).T) @ wd
    (out * c.float().cuda()).sum().backward()
    assert float((out.detach().double().cpu() - ref.detach()).abs().max()) < 2e-5
    assert float((wd.grad.double().cpu() - w.grad).abs().max() / w.grad.abs().max()) < 2e-5
    assert float((zd.grad.double().cpu() - z.grad).abs().max() / z.grad.abs().max()) < 2e-5


@pytest.mark.parametrize("n,B,pairs", [(32, 300, 8), (5, 37, 3), (40, 257, 2), (64, 64, 1), (1, 10, 2)])
def test_gauss_ll_vs_fp64(ops, n, B, pairs):
    """svae_gauss_ll (QuadraticDiscriminantFilter.cgll, reference disentangle.py:129-134: torch.linalg.solve + torch.logdet)
    against fp64 torch, values and the gradient with respect to x through the autograd wrapper; the last pair is a general
    (non-symmetric, first pivot zero) matrix so that the row pivoting and the S^-T half of the gradient are exercised."""
    g = torch.Generator().manual_seed(7 * n + pairs)
    X = torch.randn(pairs, 4 * n + 8, n, generator=g, dtype=torch.float64)
    S = X.transpose(1, 2) @ X / (4 * n + 8) + 0.1 * torch.eye(n, dtype=torch.float64)
    if n > 1:
        S[-1] = torch.randn(n, n, generator=g, dtype=torch.float64) * 0.3 + torch.eye(n, dtype=torch.float64)
        S[-1][0, 0] = 0.0
        if torch.linalg.det(S[-1]) < 0:  # keep the determinant positive: swap two rows
            S[-1][[1, 2 % n]] = S[-1][[2 % n, 1]]
    m = torch.randn(pairs, n, generator=g, dtype=torch.float64)
    x = torch.randn(B, n, generator=g, dtype=torch.float64, requires_grad=True)
    wts = torch.randn(pairs, B, generator=g, dtype=torch.float64)
    ref = torch.stack([-0.5 * (torch.logdet(S[p]) + ((x - m[p]) * torch.linalg.solve(S[p], (x - m[p]).T).T).sum(1)) for p in range(pairs)])
    (ref * wts).sum().backward()
    xd = x.detach().float().cuda().requires_grad_(True)
    out = ops.gauss_ll_autograd(xd, m.float().cuda(), S.float().cuda())
    (out * wts.float().cuda()).sum().backward()
    assert float((out.detach().double().cpu() - ref.detach()).abs().max() / ref.detach().abs().max()) < 2e-5
    assert float((xd.grad.double().cpu() - x.grad).abs().max() / x.grad.abs().max()) < 2e-5
    # a strided view of a wider buffer (mu inside the heads' output) is read in place
    wide = torch.zeros(B, n + 3, device="cuda")
    wide[:, :n] = xd.detach()
    ll2, _ = ops.gauss_ll(wide[:, :n], m.float().cuda(), S.float().cuda(), want_grad=False)
    assert torch.equal(ll2, out.detach())


def test_gauss_ll_degenerate_matrices(ops):
    """torch.logdet semantics: NaN for a negative determinant, -inf (log-likelihood +inf or NaN) for a singular matrix."""
    S = torch.stack([torch.diag(torch.tensor([1.0, -2.0, 3.0])), torch.diag(torch.tensor([1.0, 0.0, 3.0])), torch.eye(3)]).cuda()
    x = torch.randn(5, 3, device="cuda")
    ll, _ = ops.gauss_ll(x, torch.zeros(3, 3, device="cuda"), S, want_grad=False)
    assert torch.isnan(ll[0]).all()
    assert not torch.isfinite(ll[1]).any()
    ref = -0.5 * (x * x).sum(1)
    assert float((ll[2] - ref).abs().max()) < 1e-5


@pytest.mark.parametrize("B,S,zx,dy,mode", [(300, 300, 32, 3, "sphere"), (37, 100, 5, 1, "diagonal"), (64, 257, 40, 7, "diagonal"),
                                            (16, 5, 32, 2, "sphere"), (1, 1, 8, 1, "sphere")])
def test_kde_mi_vs_fp64(ops, B, S, zx, dy, mode):
    """svae_kde_mi (MutInfoEstimator.forward, reference disentangle.py:278-317) against the estimator's own chunked torch evaluation
    in fp64 on the CPU -- the path the reference-generated `mcmi_tiny` fixture pins: the loss and its gradient with respect to the
    latent means, both variance modes, sizes that are not multiples of the 16-sample / 64-centre tiles."""
    from scrubvae_amd.model.disentangle import MutInfoEstimator
    g = torch.Generator().manual_seed(B + S + zx)
    x_s = torch.randn(S, zx, generator=g, dtype=torch.float64)
    y_s = torch.randn(S, dy, generator=g, dtype=torch.float64)
    Ls = torch.diag_embed(torch.rand(S, zx, generator=g, dtype=torch.float64) + 0.2) if mode == "diagonal" else None
    est = MutInfoEstimator(x_s, y_s, 0.7, var_mode=mode, model_var=Ls)
    x = (x_s[torch.randint(0, S, (B,), generator=g)] + 0.5 * torch.randn(B, zx, generator=g, dtype=torch.float64)).requires_grad_(True)
    y = torch.randn(B, dy, generator=g, dtype=torch.float64)
    ref = est(x, y)
    ref.backward()
    est_d = MutInfoEstimator(x_s.float().cuda(), y_s.float().cuda(), 0.7, var_mode=mode, model_var=None if Ls is None else Ls.float().cuda())
    xd = x.detach().float().cuda().requires_grad_(True)
    out = est_d(xd, y.float().cuda())
    out.backward()
    assert abs(float(out) - float(ref)) < 2e-5 * max(1.0, abs(float(ref)))
    # the gradient is a difference of two softmax averages of (x_s - x) / var (terms of order 1 / B): with few centres it nearly
    # cancels, so the floor of the scale is that of a term, not of the result
    assert float((xd.grad.double().cpu() - x.grad).abs().max()) < 2e-5 * max(float(x.grad.abs().max()), 1.0 / B)


def test_heads_diag(ops):
    B, z = 37, 8
    h = torch.randn(B, 2 * z, dtype=torch.float64, requires_grad=True)
    eps = torch.randn(B, z, dtype=torch.float64)
    mu, raw = h[:, :z], h[:, z:]
    L = O.cholesky_L(raw, z, True)
    zz = torch.matmul(L, eps[..., None]).squeeze(-1) + mu
    kl = O.prior_loss(mu, L)
    dz = torch.randn(B, z, dtype=torch.float64)
    (0.7 * kl + (zz * dz).sum()).backward()
    hd = torch.zeros(B, 16, device="cuda"); hd[:, :2 * z] = h.detach().float().cuda()
    mud, sd, zd = torch.empty(B, z, device="cuda"), torch.empty(B, z, device="cuda"), torch.zeros(B, 16, device="cuda")
    klp = torch.empty(ops.heads_blocks(B, z), device="cuda")
    ops.heads_diag_fwd(hd, 16, eps.float().cuda(), mud, sd, zd, 16, klp, B, z)
    assert relerr(zd.cpu()[:, :z], zz.detach()) < 1e-6
    assert relerr(sd.cpu(), L.diagonal(dim1=-2, dim2=-1).detach()) < 1e-6
    assert abs(float(klp.sum()) / B - float(kl)) < 1e-5 * abs(float(kl))
    dh = torch.zeros(B, 16, device="cuda")
    dzd = torch.zeros(B, 16, device="cuda"); dzd[:, :z] = dz.float().cuda()
    ops.heads_diag_bwd(hd, 16, eps.float().cuda(), sd, dzd, 16, None, None, 0.7 / B, dh, B, z)
    assert relerr(dh.cpu()[:, :2 * z], h.grad) < 2e-6


@pytest.mark.parametrize("J", [18, 23])
def test_pose_tail(ops, J):
    """tanh + unpack + FK + JPE/root sums + analytic backward vs autograd on the oracle."""
    from scrubvae_amd._lib import make_tree
    tree = O.skeleton_tree(J)
    B, W = 3, 50  # 150 rows: 2 full blocks + a ragged one
    rows = B * W
    C6 = 6 * J
    ld = (C6 + 3 + 15) // 16 * 16
    g = torch.Generator().manual_seed(J)
    y = torch.randn(rows, ld, generator=g, dtype=torch.float64)
    y[:, C6 + 3:] = 0
    y.requires_grad_(True)
    arena = torch.tensor([[-1.0, -2.0, -0.5], [3.0, 2.0, 1.5]], dtype=torch.float64)
    offsets = torch.randn(rows, J, 3, generator=g, dtype=torch.float64)
    target = torch.randn(rows, J, 3, generator=g, dtype=torch.float64)
    root = torch.randn(rows, 3, generator=g, dtype=torch.float64)
    xh = torch.tanh(y)
    x6d = xh[:, :C6].reshape(rows, J, 6)
    root_hat = O.inv_normalize_root(xh[:, C6:C6 + 3], arena)
    pose = O.fwd_kin(x6d, tree, offsets, torch.zeros(rows, 3, dtype=torch.float64), eps=1e-8)
    jpe = ((target - pose) ** 2).sum()
    rl = ((root_hat - root) ** 2).sum()
    (0.37 * jpe + 1.3 * rl).backward()

    dev = lambda t: t.detach().float().cuda().contiguous()
    nb = ops.tail_blocks(rows)
    x6o, ro = torch.empty(rows, C6, device="cuda"), torch.empty(rows, 3, device="cuda")
    lp = torch.empty(nb, 2, device="cuda")
    dy = torch.full((rows, ld), float("nan"), device="cuda")
    ops.pose_tail(dev(y), ld, dev(offsets), dev(target), dev(root), arena.flatten().tolist(), make_tree(J, tree),
                  0.37, 1.3, None, None, x6o, ro, lp, dy, rows)
    assert relerr(x6o.cpu(), x6d.detach().reshape(rows, C6)) < 2e-6
    assert relerr(ro.cpu(), root_hat.detach()) < 2e-6
    s = lp.double().sum(0).cpu()
    assert abs(float(s[0]) - float(jpe)) < 2e-5 * float(jpe)
    assert abs(float(s[1]) - float(rl)) < 2e-5 * float(rl)
    assert relerr(dy.cpu(), y.grad) < 5e-4  # ill-conditioned normalisations: fp32 floor ~1e-4
    # eval form: no gradient buffer
    ops.pose_tail(dev(y), ld, dev(offsets), dev(target), dev(root), arena.flatten().tolist(), make_tree(J, tree),
                  0.0, 0.0, None, None, x6o, ro, lp, None, rows)
    assert abs(float(lp.double().sum(0)[0]) - float(jpe)) < 2e-5 * float(jpe)


def test_rot_loss(ops):
    n = 1000
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, 6, generator=g, dtype=torch.float64)
    xh = (x + 0.3 * torch.randn(n, 6, generator=g, dtype=torch.float64)).requires_grad_(True)  # well-conditioned
    loss = O.stable_rotation_loss(x, xh)
    loss.backward()
    part = torch.empty(ops.rot_blocks(n), device="cuda")
    dxh = torch.empty(n, 6, device="cuda")
    ops.rot_loss(x.float().cuda(), xh.detach().float().cuda(), 1.0, part, dxh, n)
    assert abs(float(part.double().sum()) - float(loss)) < 1e-5 * float(loss)
    assert relerr(dxh.cpu(), xh.grad) < 1e-3


def test_adam_and_norm(ops):
    n = 4096 + 64
    g = torch.Generator().manual_seed(9)
    p0 = torch.randn(n, generator=g)
    p = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([p], lr=1e-3)
    pd, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for t in range(1, 4):
        gr = torch.randn(n, generator=g)
        p.grad = gr.clone()
        opt.step()
        ops.adam_step(pd, gr.cuda(), m, v, 1e-3, 0.9, 0.999, 1e-8, 0.01, t, True)
    assert relerr(pd.cpu(), p.detach()) < 1e-6
    part = torch.empty(ops.sumsq_blocks(n), device="cuda")
    ops.sumsq_partial(pd, part)
    out = torch.zeros(1, device="cuda")
    ops.reduce_rows(part, part.numel(), 1, 1.0, out)
    assert abs(float(out) - float((pd.double() ** 2).sum())) < 1e-5 * float(out)


def test_small_losses(ops):
    rows, Cn = 300, 3
    g = torch.Generator().manual_seed(2)
    pred = torch.randn(rows, Cn, generator=g, dtype=torch.float64, requires_grad=True)
    tgt = torch.randn(rows, Cn, generator=g, dtype=torch.float64)
    l = ((pred - tgt) ** 2).sum(); (0.5 * l).backward()
    pd = torch.zeros(rows, 16, device="cuda"); pd[:, :Cn] = pred.detach().float().cuda()
    part = torch.empty(ops.rowloss_blocks(rows), device="cuda")
    dp = torch.zeros(rows, 16, device="cuda")
    ops.mse_sum(pd, 16, tgt.float().cuda().contiguous(), Cn, rows, Cn, 0.5, part, dp)
    assert abs(float(part.double().sum()) - float(l)) < 1e-5 * float(l)
    assert relerr(dp.cpu()[:, :Cn], pred.grad) < 1e-6
    # CE on integer labels
    logits = torch.randn(rows, 4, generator=g, dtype=torch.float64, requires_grad=True)
    lab = torch.randint(0, 4, (rows,), generator=g)
    l = F.cross_entropy(logits, lab, reduction="sum"); l.backward()
    ld_ = torch.zeros(rows, 16, device="cuda"); ld_[:, :4] = logits.detach().float().cuda()
    dl = torch.zeros(rows, 16, device="cuda")
    ops.ce_sum(ld_, 16, lab.int().cuda(), rows, 4, 1.0, part, dl)
    assert abs(float(part.double().sum()) - float(l)) < 1e-5 * float(l)
    assert relerr(dl.cpu()[:, :4], logits.grad) < 2e-6
    # double softmax CE (adversarial-net quirk)
    logits = torch.randn(rows, 2, generator=g, dtype=torch.float64, requires_grad=True)
    yoh = F.one_hot((torch.arange(rows) >= rows // 2).long(), 2).double()
    l = F.cross_entropy(torch.softmax(logits, -1), yoh, reduction="sum"); (-0.25 * l).backward()
    ld_ = torch.zeros(rows, 16, device="cuda"); ld_[:, :2] = logits.detach().float().cuda()
    dl = torch.zeros(rows, 16, device="cuda")
    ops.double_softmax_ce_sum(ld_, 16, rows, -0.25, part, dl)
    assert abs(float(part.double().sum()) - float(l)) < 1e-5 * float(l)
    assert relerr(dl.cpu()[:, :2], logits.grad) < 2e-6


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("code", [1, 128128, 64128, 128064, 64064])
def test_wgrad_kernel_variants(ops, case, code):
    """Every weight-gradient kernel variant (tile overrides and the tap-fused kernel) gives the
    same result as torch's conv backward."""
    B, L, Cin, Cout, k, s, p, tr = case
    g = torch.Generator().manual_seed(7 + sum(case[:7]))
    x = torch.randn(B, Cin, L, generator=g, dtype=torch.float64)
    w = (torch.randn(*((Cin, Cout, k) if tr else (Cout, Cin, k)), generator=g, dtype=torch.float64)).requires_grad_(True)
    y = (F.conv_transpose1d if tr else F.conv1d)(x, w, None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    cv = ops.Conv(B, L, Cin, Cout, k, s, p, 1, tr)
    cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
    cv.desc.tile[2] = code
    cv._ws_bytes = None
    xd, dyd = to_nlc(x), to_nlc(dy)
    ws = torch.empty(cv.wgrad_workspace_bytes() // 4 + 4, device="cuda")
    dwd = torch.full(cv.weight_shape, float("nan"), device="cuda")
    cv.wgrad(xd, dyd, dwd, None, ws)
    dw = ops.conv_weight_from_tio(dwd.cpu(), Cin, Cout, tr)
    assert relerr(dw, w.grad) < 2e-6 * math.sqrt(B * y.shape[-1]) + 2e-6
    assert not torch.isnan(dwd).any()


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("code", [128128, 64128, 128064, 64064, 1128128, 1064128, 1128064, 1064064])
def test_gather_kernel_variants(ops, case, code):
    B, L, Cin, Cout, k, s, p, tr = case
    g = torch.Generator().manual_seed(9 + sum(case[:7]))
    x = torch.randn(B, Cin, L, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(*((Cin, Cout, k) if tr else (Cout, Cin, k)), generator=g, dtype=torch.float64) / math.sqrt(Cin * k)
    y = (F.conv_transpose1d if tr else F.conv1d)(x, w, None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    cv = ops.Conv(B, L, Cin, Cout, k, s, p, 1, tr)
    cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
    cv.desc.tile[0] = cv.desc.tile[1] = code
    wd = ops.conv_weight_to_tio(w.float(), tr).cuda().contiguous()
    yd = torch.full((B * cv.l_out, cv.c_out_p), float("nan"), device="cuda")
    cv.fwd(to_nlc(x.detach()), wd, None, yd)
    assert relerr(from_nlc(yd, B, cv.l_out, Cout), y.detach()) < 2e-6 * math.sqrt(Cin * k) + 2e-6
    dxd = torch.full((B * L, cv.c_in_p), float("nan"), device="cuda")
    cv.dgrad(to_nlc(dy), wd, dxd)
    assert relerr(from_nlc(dxd, B, L, Cin), x.grad) < 2e-6 * math.sqrt(Cout * k) + 2e-6


# ------------------------------------------------------------------ split-bf16 GEMM kernels
_SPLIT_TOL = {3: 2e-6, 2: 6e-5, 1: 1.2e-2, 22: 8e-6}  # per sqrt(K): pieces=3 is held to the fp32 kernels' bound; 22 = two fp16 pieces (forward)


def _split_cases():
    out = []
    for case in CONV_CASES:
        for code in (128128, 64128, 128064, 64064, 1128128, 1064128, 1128064, 1064064, 2128128, 2128064, 3128128, 3128064,
                     4128128, 4128064, 4064128, 5128128, 5128064, 5064128, 5064064,
                     6128128, 6128064, 6064128, 7128128, 7128064, 7064128, 7064064, 8128128, 8128064, 9128128, 9128064,
                     10128128, 10128064, 11128128, 11128064, 12128064, 13128064, 0):
            out.append((case, code, 3))
        for code in (64064, 2128064, 3128128, 4128128, 5064064, 8128128, 8128064, 9128128, 9128064,
                     10128128, 10128064, 11128128, 11128064, 12128128, 12128064, 13128128, 13128064, 14128128, 15128128,
                     16128128, 16128064, 17128128, 17128064, 18128128, 18128064, 19128128, 29128128):  # 2 pieces: the backward pass of bf16x6b3
            out.append((case, code, 2))
        out.append((case, 1128064, 1))
        for code in (128128, 1064064, 2128064, 3128128, 4128128, 5064064, 6128064, 7064128, 8128128, 8128064, 9128128, 9128064,
                     10128128, 11128128, 11128064, 12128128, 13128128, 14128128, 15128128,
                     16128128, 16128064, 17128128, 18128128, 18128064, 19128128, 29128128):  # two fp16 pieces / 3 products (the forward of f16x3b3)
            out.append((case, code, 22))
    return out


@pytest.mark.parametrize("case,code,pieces", _split_cases())
def test_split_gather_kernels(ops, case, code, pieces):
    """Forward / data-gradient on the bf16 matrix cores with 3 / 2 / 1 bf16 pieces per operand
    (csrc/gemm_bf16s.hip) against torch fp64; 3 pieces must meet the fp32 kernels' tolerance."""
    B, L, Cin, Cout, k, s, p, tr = case
    g = torch.Generator().manual_seed(11 + sum(case[:7]))
    x = torch.randn(B, Cin, L, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(*((Cin, Cout, k) if tr else (Cout, Cin, k)), generator=g, dtype=torch.float64) / math.sqrt(Cin * k)
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    y = (F.conv_transpose1d if tr else F.conv1d)(x, w, b, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    cv = ops.Conv(B, L, Cin, Cout, k, s, p, 1, tr, pieces=pieces)
    if pieces == 22:  # fp16 pieces exist for the forward only: the data-gradient of this mode runs two bf16 pieces
        cv.dgrad_pieces = cv.wgrad_pieces = 2
    cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
    cv.desc.tile[0] = cv.desc.tile[1] = code
    ops.bump_weight_epoch()
    wd = ops.conv_weight_to_tio(w.float(), tr).cuda().contiguous()
    bd = torch.zeros(cv.c_out_p); bd[:Cout] = b.float(); bd = bd.cuda()
    yd = torch.full((B * cv.l_out, cv.c_out_p), float("nan"), device="cuda")
    tol = _SPLIT_TOL[pieces]
    try:
        cv.fwd(to_nlc(x.detach()), wd, bd, yd)
        cv.dgrad(to_nlc(dy), wd, torch.empty(B * L, cv.c_in_p, device="cuda"))
    except RuntimeError as e:
        if code // 1000000 in (9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 29) and ("does not fit" in str(e) or "do not fit" in str(e)):
            pytest.skip("256-row halo image larger than LDS for this geometry (the tuner skips it the same way)")
        raise
    assert relerr(from_nlc(yd, B, cv.l_out, Cout), y.detach()) < tol * math.sqrt(Cin * k) + tol
    assert float(yd[:, Cout:].abs().max() if cv.c_out_p > Cout else 0) == 0.0
    # BatchNorm statistics from the epilogue (svae_conv_fwd_split_stats): per-tile (sum, sum of squares) of the values the launch
    # WRITES (with accumulate: previous content included), summed over the tiles == column sums of the output
    nt = cv.stats_tiles()
    assert nt > 0
    part = torch.full((nt, 2, cv.c_out_p), float("nan"), device="cuda")
    cv.fwd(to_nlc(x.detach()), wd, bd, yd, accumulate=True, stats=part)
    assert relerr(from_nlc(yd, B, cv.l_out, Cout), 2 * y.detach()) < tol * math.sqrt(Cin * k) + tol
    y2 = yd.double().cpu()
    sums = part.double().cpu().sum(0)
    assert not torch.isnan(part).any()
    assert relerr(sums[0], y2.sum(0)) < 1e-5 * math.sqrt(y2.shape[0]) and relerr(sums[1], (y2 * y2).sum(0)) < 1e-5
    dxd = torch.full((B * L, cv.c_in_p), float("nan"), device="cuda")
    cv.dgrad(to_nlc(dy), wd, dxd)
    tol_d = _SPLIT_TOL[2] if pieces == 22 else tol
    assert relerr(from_nlc(dxd, B, L, Cin), x.grad) < tol_d * math.sqrt(Cout * k) + tol_d
    assert not torch.isnan(dxd).any()


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("code,pieces", [(128128, 3), (64128, 3), (128064, 3), (64064, 3), (0, 3), (64064, 2), (128064, 1),
                                         (256256, 2), (1256256, 2), (256128, 2), (1128256, 2), (1256256, 3), (1256128, 3),
                                         (1128128, 3), (1064128, 3), (1128064, 3), (1064064, 3),
                                         (2256256, 2), (3256128, 2), (2128128, 2), (3064064, 2), (2064128, 3),
                                         (4064128, 2), (6064128, 2), (4128064, 2), (6128064, 2), (4128128, 2), (6128128, 2),
                                         (12064128, 2), (14128064, 2), (12128128, 2), (14128128, 2),
                                         (16064128, 2), (18128128, 2), (17064064, 3), (16128064, 2), (18128064, 2)])
def test_split_wgrad_kernels(ops, case, code, pieces):
    B, L, Cin, Cout, k, s, p, tr = case
    g = torch.Generator().manual_seed(13 + sum(case[:7]))
    x = torch.randn(B, Cin, L, generator=g, dtype=torch.float64)
    w = (torch.randn(*((Cin, Cout, k) if tr else (Cout, Cin, k)), generator=g, dtype=torch.float64)).requires_grad_(True)
    y = (F.conv_transpose1d if tr else F.conv1d)(x, w, None, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    cv = ops.Conv(B, L, Cin, Cout, k, s, p, 1, tr, pieces=pieces)
    cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
    cv.desc.tile[2] = code
    cv._ws_bytes = None
    xd, dyd = to_nlc(x), to_nlc(dy)
    ws = torch.empty(cv.wgrad_workspace_bytes() // 4 + 4, device="cuda")
    dwd = torch.full(cv.weight_shape, float("nan"), device="cuda")
    dbd = torch.full((cv.c_out_p,), float("nan"), device="cuda")
    try:
        cv.wgrad(xd, dyd, dwd, dbd, ws)
    except RuntimeError as e:
        if code >= 4000000 and ("all-taps kernel" in str(e) or "wgrad taps" in str(e) or "flat-tap variant" in str(e)):
            pytest.skip("geometry outside the all-taps kernel (odd-length strided transposed conv, > 6 taps): the tuner skips it the same way")
        raise
    tol = _SPLIT_TOL[pieces]
    bound = tol * math.sqrt(B * y.shape[-1]) + tol
    assert relerr(ops.conv_weight_from_tio(dwd.cpu(), Cin, Cout, tr), w.grad) < bound
    assert not torch.isnan(dwd).any()
    cv.wgrad(xd, dyd, dwd, dbd, ws, accumulate=True)
    assert relerr(ops.conv_weight_from_tio(dwd.cpu(), Cin, Cout, tr), 2 * w.grad) < bound


@pytest.mark.parametrize("geo", [(3, 4, 48, 40), (5, 7, 32, 64), (2, 25, 16, 32), (64, 13, 128, 64), (300, 4, 64, 128)])
@pytest.mark.parametrize("fcode,pieces", [(8128128, 22), (8128064, 3), (9128128, 22), (9128064, 2), (9128128, 3), (0, 22), (29128128, 22),
                                          (29128128, 2)])
@pytest.mark.parametrize("wcode", [4064128, 6128064, 12064128, 14128064, 0])
def test_conv_behind_upsample_fused(ops, geo, fcode, pieces, wcode):
    """nn.Upsample(scale_factor=2, mode="linear") -> nn.Conv1d(k + 1 taps) of the decoder's skip path (reference residual.py:153-170)
    with the upsample folded into the operand staging of the forward (halo kernels, tile codes 8 / 9) and of the all-taps weight
    gradient (svae_conv_desc.up2): against torch fp64, and against the unfused launches (svae_upsample2_fwd + the same kernels), which
    see the same blended fp32 values."""
    B, L, Cin, Cout = geo
    k, pad = 6, 2
    g = torch.Generator().manual_seed(17 + sum(geo))
    xh = torch.randn(B, Cin, L, generator=g, dtype=torch.float64)
    w = (torch.randn(Cout, Cin, k, generator=g, dtype=torch.float64) / math.sqrt(Cin * k)).requires_grad_(True)
    b = torch.randn(Cout, generator=g, dtype=torch.float64)
    up = F.interpolate(xh, scale_factor=2, mode="linear", align_corners=False)
    y = F.conv1d(up, w, b, padding=pad)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    with pytest.raises(ValueError):
        ops.Conv(B, 2 * L, Cin, Cout, 5, 1, 2, pieces=pieces, up2=True)  # 5 taps: not a geometry behind the upsampler
    cvu = ops.Conv(B, 2 * L, Cin, Cout, k, 1, pad, pieces=pieces, up2=True)
    cvp = ops.Conv(B, 2 * L, Cin, Cout, k, 1, pad, pieces=pieces)
    cvu.up2_wgrad = True  # the fused weight gradient too (the model's default reads the forward's by-product instead)
    for cv in (cvu, cvp):
        cv.wgrad_pieces = cv.dgrad_pieces = 2
        cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
        cv.desc.tile[0] = fcode if (fcode or cv is cvu) else 8128128
        cv.desc.tile[2] = wcode if (wcode or cv is cvu) else 4064128
        cv._ws_bytes = None
    ops.bump_weight_epoch()
    wd = ops.conv_weight_to_tio(w.detach().float()).cuda().contiguous()
    bd = torch.zeros(cvu.c_out_p); bd[:Cout] = b.float(); bd = bd.cuda()
    xd = to_nlc(xh)
    upd = torch.empty(B * 2 * L, cvu.c_in_p, device="cuda")
    ops.upsample2_fwd(xd, upd, B, L, cvu.c_in_p, cvu.c_in_p)
    yu = torch.full((B * cvu.l_out, cvu.c_out_p), float("nan"), device="cuda")
    yp = torch.full_like(yu, float("nan"))
    try:
        cvu.fwd(xd, wd, bd, yu)
    except RuntimeError as e:
        if fcode // 1000000 in (9, 29) and "does not fit" in str(e):
            pytest.skip("256-row halo image larger than LDS for this geometry")
        raise
    cvp.fwd(upd, wd, bd, yp)
    # by-product: the upsampled tensor itself, every row exactly once (NaN-filled before)
    upo = torch.full_like(upd, float("nan"))
    yo = torch.full_like(yu, float("nan"))
    cvu.fwd(xd, wd, bd, yo, up_out=upo)
    assert torch.equal(yo, yu)
    assert torch.equal(upo, upd), (int(torch.isnan(upo).sum()), float((upo - upd).abs().nan_to_num(0.0).max()))
    with pytest.raises(RuntimeError):
        cvp.fwd(upd, wd, bd, yp, up_out=upo)
    tol = _SPLIT_TOL[pieces]
    assert relerr(from_nlc(yu, B, cvu.l_out, Cout), y.detach()) < tol * math.sqrt(Cin * k) + tol
    assert relerr(yu, yp) < 1e-6  # the same blended values through the same kernel
    assert "true>" in cvu.kernel_name("fwd") and "false>" in cvp.kernel_name("fwd")
    # fused BatchNorm statistics and accumulate through the fused path as through the plain one
    nt = cvu.stats_tiles()
    part = torch.full((nt, 2, cvu.c_out_p), float("nan"), device="cuda")
    cvu.fwd(xd, wd, bd, yu, accumulate=True, stats=part)
    assert relerr(part.double().sum(0)[0], yu.double().sum(0)) < 1e-5 * math.sqrt(yu.shape[0])
    # weight gradient
    dyd = to_nlc(dy)
    wsu = torch.empty(cvu.wgrad_workspace_bytes() // 4 + 4, device="cuda")
    wsp = torch.empty(cvp.wgrad_workspace_bytes() // 4 + 4, device="cuda")
    dwu = torch.full(cvu.weight_shape, float("nan"), device="cuda")
    dwp = torch.full(cvu.weight_shape, float("nan"), device="cuda")
    cvu.wgrad(xd, dyd, dwu, None, wsu)
    cvp.wgrad(upd, dyd, dwp, None, wsp)
    bound = _SPLIT_TOL[2] * math.sqrt(B * y.shape[-1]) + _SPLIT_TOL[2]
    assert relerr(ops.conv_weight_from_tio(dwu.cpu(), Cin, Cout), w.grad) < bound
    assert relerr(dwu, dwp) < 1e-6
    cvu.wgrad(xd, dyd, dwu, None, wsu, accumulate=True)
    assert relerr(ops.conv_weight_from_tio(dwu.cpu(), Cin, Cout), 2 * w.grad) < bound
    # the data gradient is the plain geometry's (with respect to the upsampled input)
    gu = torch.full((B * 2 * L, cvu.c_in_p), float("nan"), device="cuda")
    cvu.dgrad(dyd, wd, gu)
    upr = up.detach().requires_grad_(True)
    F.conv1d(upr, w.detach(), b, padding=pad).backward(dy)
    assert relerr(from_nlc(gu, B, 2 * L, Cin), upr.grad) < _SPLIT_TOL[2] * math.sqrt(Cout * k) + _SPLIT_TOL[2]
    # kernels without the fused path refuse the flag instead of reading the half-length tensor as a full one
    cvu.desc.tile[0] = 16128128
    with pytest.raises(RuntimeError):
        cvu.fwd(xd, wd, bd, yu)
    cvu.desc.tile[0] = fcode
    cvu.desc.tile[2] = 2128128
    cvu._ws_bytes = None
    with pytest.raises(RuntimeError):
        cvu.wgrad(xd, dyd, dwu, None, torch.empty(cvu.wgrad_workspace_bytes() // 4 + 4, device="cuda"))
    # ... while the plain weight gradient of the same Conv takes the full-length tensor on any kernel
    cvu.up2_wgrad = False
    cvu._ws_bytes = None
    cvu.__dict__.pop("_names", None)
    cvu.wgrad(upd, dyd, dwu, None, torch.empty(cvu.wgrad_workspace_bytes() // 4 + 4, device="cuda"))
    assert relerr(ops.conv_weight_from_tio(dwu.cpu(), Cin, Cout), w.grad) < bound


def test_split_weight_copies_follow_the_epoch(ops):
    """The bf16 weight pieces are refreshed when the weight epoch is bumped, not before."""
    cv = ops.Conv(4, 16, 32, 32, 5, 1, 2, pieces=3)
    cv.__dict__["_tuned"] = {"fwd", "dgrad", "wgrad"}
    g = torch.Generator().manual_seed(3)
    x = torch.randn(4 * 16, 32, generator=g).cuda()
    w = torch.randn(5, 32, 32, generator=g).cuda()
    y0 = torch.empty(4 * 16, 32, device="cuda"); y1 = torch.empty_like(y0); y2 = torch.empty_like(y0)
    ops.bump_weight_epoch()
    cv.fwd(x, w, None, y0)
    w.mul_(2.0)
    cv.fwd(x, w, None, y1)          # same epoch: stale pieces by contract
    ops.bump_weight_epoch()
    cv.fwd(x, w, None, y2)
    assert torch.equal(y0, y1)
    assert relerr(y2, 2 * y0) < 1e-6


def test_split_weights_batched_matches_single(ops):
    """One batched launch over several convs writes the same piece planes as the per-conv kernel."""
    g = torch.Generator().manual_seed(5)
    specs = [(2, 8, 48, 32, 5, 1, 2, False), (2, 8, 64, 144, 6, 1, 2, False), (3, 4, 32, 16, 5, 2, 2, True), (4, 1, 80, 32, 1, 1, 0, False)]
    convs, ws = [], []
    for B, L, cin, cout, k, s, p, tr in specs:
        cv = ops.Conv(B, L, cin, cout, k, s, p, 1, tr, pieces=3)
        w = torch.randn(*cv.weight_shape, generator=g).cuda()
        convs.append(cv); ws.append(w)
    ops.bump_weight_epoch()
    single = [cv.split_weights(w).clone() for cv, w in zip(convs, ws)]
    for cv in convs:
        cv._wsplit.zero_()
    ops.bump_weight_epoch()
    ops.split_weights_batched(list(zip(convs, ws)))
    torch.cuda.synchronize()
    for cv, ref in zip(convs, single):
        assert cv._split_epoch == ops.WEIGHT_EPOCH
        n = ref.numel() - 64
        assert torch.equal(cv._wsplit[:n], ref[:n])


# ------------------------------------------------------------------ fused MLP ensembles (csrc/ensemble.hip)
class _StubLin:
    """The attributes FusedEnsembleRunner reads from a LinearP: TIO weight [1][in_p][out_p], bias [out_p], their grads."""

    def __init__(self, inf, outf, g, trainable=True):
        p16 = lambda n: (n + 15) // 16 * 16
        self.in_f, self.out_f, self.in_lib, self.out_lib = inf, outf, p16(inf), p16(outf)
        self.w64 = torch.randn(inf, outf, generator=g, dtype=torch.float64) / math.sqrt(inf)
        self.b64 = torch.randn(outf, generator=g, dtype=torch.float64) * 0.5
        w = torch.zeros(1, self.in_lib, self.out_lib)
        w[0, :inf, :outf] = self.w64.float()
        b = torch.zeros(self.out_lib)
        b[:outf] = self.b64.float()
        self.weight, self.bias = w.cuda(), b.cuda()
        self.weight.grad = torch.full_like(self.weight, 7.0) if trainable else None  # stale values: the kernel must overwrite
        self.bias.grad = torch.full_like(self.bias, 7.0) if trainable else None


class _StubEns:
    def __init__(self, ind, outd, g, trainable=True):
        self.in_dim, self.out_dim = ind, outd
        dims = [[ind, ind, ind, outd], [ind, ind, outd], [ind, ind, ind // 2, outd], [ind, 2 * ind, 2 * ind, outd]]
        self.m = [[_StubLin(a, b, g, trainable) for a, b in zip(d, d[1:])] for d in dims]

    def members(self):
        return self.m


@pytest.mark.parametrize("case", [(8, 3, 37, 1), (32, 2, 200, 1), (37, 2, 70, 2), (13, 2, 5, 2), (32, 4, 64, 1)])
def test_fused_ensemble_fwd_bwd(ops, case, monkeypatch):
    """MLPEnsemble (disentangle.py:583-632) forward, input gradient (with the reversal coefficient, both halves of the
    adversarial net's doubled batch summed) and parameter gradients on the fused kernels vs fp64 torch autograd."""
    from scrubvae_amd.model import disentangle as D
    monkeypatch.setattr(D, "LinearP", _StubLin)  # the runner picks the Linear layers of a member by isinstance
    ind, outd, B, halves = case
    g = torch.Generator().manual_seed(sum(case))
    z = ind if halves == 1 else ind - 5
    ens = _StubEns(ind, outd, g, trainable=(halves == 1))
    r = D.FusedEnsembleRunner(ens, B, "cuda", halves)
    assert r.fits()
    zp = (z + 15) // 16 * 16
    mu = torch.zeros(B, zp, dtype=torch.float64)
    mu[:, :z] = torch.randn(B, z, generator=g, dtype=torch.float64)
    var = torch.randn(B, 5, generator=g, dtype=torch.float64)
    perm = torch.randperm(B, generator=g)
    mu_c = mu.float().cuda()
    if halves == 1:
        outs = r.forward(mu_c, z)
        x64 = mu[:, :z].clone().requires_grad_(True)
        xin = x64
    else:
        outs = r.forward(mu_c, z, src1=var.float().cuda().contiguous(), perm=perm.cuda(), shuf_col=3)
        x64 = mu[:, :z].clone().requires_grad_(True)
        v2 = var.clone()
        v2[:, 3] = var[perm, 3]
        xin = torch.cat([torch.cat([x64, x64], 0), torch.cat([var, v2], 0)], 1)
    ws = [[(l.w64.clone().requires_grad_(True), l.b64.clone().requires_grad_(True)) for l in mem] for mem in ens.m]
    ref_outs = []
    for mem in ws:
        h = xin
        for i, (w, b) in enumerate(mem):
            h = h @ w + b
            if i < len(mem) - 1:
                h = torch.relu(h)
        ref_outs.append(h)
    for o, ro in zip(outs, ref_outs):
        assert relerr(o.cpu()[:, :outd], ro.detach()) < 1e-5
        assert float(o[:, outd:].abs().max()) == 0.0  # padded output columns stay zero
    d_outs = [torch.randn(B * halves, outd, generator=g, dtype=torch.float64) for _ in range(4)]
    for t, d in zip(r.d_outs, d_outs):
        t.zero_()
        t[:, :outd] = d.float().cuda()
    torch.autograd.backward(ref_outs, d_outs)
    d_mu = torch.full((B, zp), 0.25).cuda()
    raw = r.backward(d_mu, -0.7, param_grads=(halves == 1), accumulate=False, want_raw=True)
    torch.cuda.synchronize()
    assert relerr(d_mu.cpu()[:, :z] - 0.25, -0.7 * x64.grad) < 2e-5
    assert zp == z or float((d_mu[:, z:] - 0.25).abs().max()) == 0.0
    if halves == 1:
        assert relerr(raw.cpu()[:, :z], x64.grad) < 2e-5
        for mem, refm in zip(ens.m, ws):
            for l, (w, b) in zip(mem, refm):
                assert relerr(l.weight.grad.cpu()[0, : l.in_f, : l.out_f], w.grad) < 2e-5
                assert relerr(l.bias.grad.cpu()[: l.out_f], b.grad) < 2e-5
                assert float(l.weight.grad[0, l.in_f:, :].abs().max() if l.in_lib > l.in_f else 0.0) == 0.0
        # accumulate: a second backward adds onto the first
        g0 = ens.m[3][1].weight.grad.clone()
        r.backward(None, 0.0, param_grads=True, accumulate=True)
        assert relerr(ens.m[3][1].weight.grad.cpu(), 2 * g0.cpu()) < 1e-6
    else:
        assert all(l.weight.grad is None for mem in ens.m for l in mem)


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_fused_ensemble_losses(ops, kind):
    """svae_ens_loss: the four members' losses / seed gradients in one launch vs the per-member kernels' definitions
    (losses.py:267-309): summed squared error, CrossEntropy(sum), CrossEntropy applied to the softmax output."""
    g = torch.Generator().manual_seed(kind)
    rows, C, ld = 300, (3 if kind == 0 else (4 if kind == 1 else 2)), 16
    outs64 = [torch.randn(rows, C, generator=g, dtype=torch.float64) for _ in range(4)]
    outs = []
    for o in outs64:
        t = torch.zeros(rows, ld)
        t[:, :C] = o.float()
        outs.append(t.cuda())
    lw, gs = [0.5, 0.25, 2.0, 1.0], [0.1, 0.2, 0.3, 0.4]
    tgt = torch.randn(rows, C, generator=g, dtype=torch.float64)
    labels = torch.randint(0, C, (rows,), generator=g)
    dp = [torch.zeros(rows, ld).cuda() for _ in range(4)]
    nb = ops.rowloss_blocks(rows)
    part = torch.zeros(4 * nb).cuda()
    ops.ens_loss(kind, outs, dp, lw, gs, tgt.float().cuda().contiguous() if kind == 0 else None, C,
                 labels.int().cuda() if kind == 1 else None, rows, C, ld, part)
    total = 0.0
    for m, o in enumerate(outs64):
        x = o.clone().requires_grad_(True)
        if kind == 0:
            l = ((x - tgt) ** 2).sum()
        elif kind == 1:
            l = F.cross_entropy(x, labels, reduction="sum")
        else:
            y = F.one_hot((torch.arange(rows) >= rows // 2).long(), 2).double()
            l = F.cross_entropy(torch.softmax(x, -1), y, reduction="sum")
        l.backward()
        total += lw[m] * float(l)
        assert relerr(dp[m].cpu()[:, :C], gs[m] * x.grad) < 2e-5
    assert abs(float(part.double().sum()) - total) <= 2e-5 * abs(total)


@pytest.mark.parametrize("B,z,lo,hi", [(64, 8, 1.0, 4.0), (257, 32, 1.0, 12.0), (33, 5, 6.5, 40.0)])
def test_heads_beta_vs_torch(ops, B, z, lo, hi):
    """model.prior = "beta": alpha / beta / mu, KL(Beta || Beta(1, 1)) and the backward -- the implicit reparameterisation gradient of
    the draw (torch's _dirichlet_grad), the KL gradient (trigamma) and the mode's -- against torch.distributions on the CPU in fp64,
    over shapes that reach every branch of the piecewise gradient approximation (series near 0 / 1, both shapes > 6, rational)."""
    g = torch.Generator().manual_seed(B + z)
    # raw head outputs chosen so that alpha, beta = softplus(raw) + 1 cover [lo, hi]
    tgt_a = lo + (hi - lo) * torch.rand(B, z, generator=g, dtype=torch.float64)
    tgt_b = lo + (hi - lo) * torch.rand(B, z, generator=g, dtype=torch.float64)
    inv = lambda t: torch.log(torch.expm1((t - 1).clamp_min(1e-3)))
    ra, rb = inv(tgt_a).float().double().requires_grad_(True), inv(tgt_b).float().double().requires_grad_(True)
    x = (torch.rand(B, z, generator=g, dtype=torch.float64) * 0.98 + 0.01).float().double()
    x[0, :] = torch.linspace(1e-4, 1 - 1e-4, z, dtype=torch.float64).float().double()  # both ends of (0, 1)
    a, b = F.softplus(ra) + 1, F.softplus(rb) + 1
    mu = (a - 1 + 1e-8) / (a + b - 2 + 2e-8) * 2 - 1
    zz = O._BetaRSample.apply(a, b, x) * 2 - 1
    q = torch.distributions.Beta(a, b)
    kl = torch.distributions.kl_divergence(q, torch.distributions.Beta(torch.ones_like(a), torch.ones_like(b))).sum()
    gz, gm = torch.randn(B, z, generator=g, dtype=torch.float64), torch.randn(B, z, generator=g, dtype=torch.float64)
    kls = 0.37
    ((zz * gz).sum() + (mu * gm).sum() + kls * kl).backward()

    zp = (z + 15) // 16 * 16
    h = torch.zeros(B, 2 * zp)
    h[:, :z], h[:, zp: zp + z] = ra.detach().float(), rb.detach().float()
    h = h.cuda()
    al, be, mud = (torch.zeros(B, zp, device="cuda") for _ in range(3))
    klp = torch.empty(ops.heads_blocks(B, z), device="cuda")
    ops.heads_beta_fwd(h, 2 * zp, al, be, mud, zp, klp, B, z, zp)
    assert relerr(al[:, :z].cpu(), a.detach()) < 2e-6 and relerr(be[:, :z].cpu(), b.detach()) < 2e-6
    assert relerr(mud[:, :z].cpu(), mu.detach()) < 5e-6
    assert abs(float(klp.double().sum()) - float(kl)) < 1e-5 * abs(float(kl)) + 1e-5
    dz = torch.zeros(B, zp); dz[:, :z] = gz.float(); dmu = torch.zeros(B, zp); dmu[:, :z] = gm.float()
    dh = torch.full((B, 2 * zp), float("nan"), device="cuda")
    ops.heads_beta_bwd(h, 2 * zp, x.float().cuda().contiguous(), al, be, zp, dz.cuda(), zp, dmu.cuda(), kls, dh, B, z, zp)
    # the piecewise approximation is evaluated in fp64 here and (for its series branches) in fp32 by torch on fp32 inputs; against the
    # fp64 evaluation of the same algorithm:
    assert relerr(dh[:, :z].cpu(), ra.grad) < 2e-5, relerr(dh[:, :z].cpu(), ra.grad)
    assert relerr(dh[:, zp: zp + z].cpu(), rb.grad) < 2e-5, relerr(dh[:, zp: zp + z].cpu(), rb.grad)
