"""GPU parity tests of the individual HIP operators, called through the C ABI, against plain fp32/fp64 PyTorch on CPU.

Integer-valued operands make the MFMA GEMM / conv checks BIT-EXACT in both dtypes (every product and partial sum is
exactly representable), which pins fragment layouts, transposes, tails and the epilogue order independent of rounding.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def G_exp():
    import gpu_ops
    return gpu_ops.exp_lib()


DTYPES = [torch.float32, torch.bfloat16]


def _ints(shape, lo, hi, g, dtype):
    return torch.randint(lo, hi + 1, shape, generator=g).to(dtype)


def _cuda(*ts):
    return [None if t is None else t.cuda() for t in ts]


# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("tile", [64, 128, 12864])
@pytest.mark.parametrize("layout", ["NT", "NN", "TT", "TN"])
@pytest.mark.parametrize("shape", [(128, 128, 128), (100, 97, 72), (257, 40, 300), (64, 520, 64), (33, 8, 8)])
def test_gemm_exact_integer(dtype, tile, layout, shape):
    import gpu_ops as G
    M, N, K = shape
    ce = 8 if dtype == torch.bfloat16 else 4
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    tA, tB = layout[0] == "T", layout[1] == "N"      # "NT" = k-contiguous A and k-contiguous B (Linear forward)
    pad = lambda n: (n + ce - 1) // ce * ce           # noqa: E731
    A = _ints((M, K), -3, 3, g, torch.float32)
    Bm = _ints((N, K), -3, 3, g, torch.float32)
    ref = A.double() @ Bm.double().t()
    if tA:
        As = torch.zeros(K, pad(M)); As[:, :M] = A.t()
    else:
        As = torch.zeros(M, pad(K)); As[:, :K] = A
    if tB:
        Bs = torch.zeros(K, pad(N)); Bs[:, :N] = Bm.t()
    else:
        Bs = torch.zeros(N, pad(K)); Bs[:, :K] = Bm
    Ad, Bd = As.to(dtype).cuda(), Bs.to(dtype).cuda()
    C = G.gemm(Ad, Bd, M, N, K, transA=tA, transB=tB, force_tile=tile, ldc=pad(N))
    torch.cuda.synchronize()
    got = C[:, :N].float().cpu().double()
    ref = ref.float().to(dtype).double()       # one output rounding (exact for fp32; RNE to bf16 otherwise)
    assert torch.equal(got, ref), (got - ref).abs().max()
    assert float(C[:, N:].float().abs().sum()) == 0.0          # pad columns untouched


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_epilogue_order(dtype):
    """bias -> relu -> (maskY) -> +R -> accumulate, and fp32 output from bf16 operands (weight-gradient form)."""
    import gpu_ops as G
    M, N, K = 70, 48, 64
    g = torch.Generator().manual_seed(1)
    A = _ints((M, K), -2, 2, g, torch.float32)
    Bm = _ints((N, K), -2, 2, g, torch.float32)
    bias = _ints((N,), -4, 4, g, torch.float32)
    R = _ints((M, N), -5, 5, g, torch.float32)
    Y = _ints((M, N), 0, 1, g, torch.float32)
    C0 = _ints((M, N), -3, 3, g, torch.float32)
    Ad, Bd, Rd, Yd = A.to(dtype).cuda(), Bm.to(dtype).cuda(), R.to(dtype).cuda(), Y.to(dtype).cuda()
    # linear forward with bias + relu + residual
    got = G.gemm(Ad, Bd, M, N, K, bias=bias.cuda(), relu=True, R=Rd).float().cpu()
    ref = (torch.relu(A @ Bm.t() + bias) + R).to(dtype).float()
    assert torch.equal(got, ref)
    # relu/dropout backward mask, scaled by 2 (exact), accumulated into an existing C
    Cd = C0.to(dtype).cuda()
    got = G.gemm(Ad, Bd, M, N, K, maskY=Yd, mask_scale=2.0, C=Cd, accumulate=True).float().cpu()
    ref = ((A @ Bm.t()) * (Y != 0) * 2.0 + C0).to(dtype).float()
    assert torch.equal(got, ref)
    # fp32 output
    got = G.gemm(Ad, Bd, M, N, K, out_f32=True)
    assert got.dtype == torch.float32 and torch.equal(got.cpu(), A @ Bm.t())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(256, 256, 2560), (96, 300, 1000), (512, 64, 333)])
def test_gemm_split_k_weight_gradient(dtype, shape):
    """dW[N,K] += dY[M,N]^T X[M,K] with the token dimension split over workgroups (fp32 atomics): exact on integers."""
    import gpu_ops as G
    Nout, Kin, Mtok = shape
    ce = 8 if dtype == torch.bfloat16 else 4
    pad = lambda n: (n + ce - 1) // ce * ce           # noqa: E731
    g = torch.Generator().manual_seed(Mtok)
    dY = torch.zeros(Mtok, pad(Nout)); dY[:, :Nout] = _ints((Mtok, Nout), -2, 2, g, torch.float32)
    X = torch.zeros(Mtok, pad(Kin)); X[:, :Kin] = _ints((Mtok, Kin), -2, 2, g, torch.float32)
    C0 = _ints((Nout, Kin), -3, 3, g, torch.float32)
    C = C0.clone().cuda()
    G.gemm(dY.to(dtype).cuda(), X.to(dtype).cuda(), Nout, Kin, Mtok, transA=True, transB=True, out_f32=True, C=C, split_k=32)
    ref = dY[:, :Nout].double().t() @ X[:, :Kin].double() + C0.double()
    assert torch.equal(C.cpu().double(), ref)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("split_k", [0, 32])
@pytest.mark.parametrize("shape", [(256, 256, 2688), (96, 300, 1000), (1024, 256, 2688), (8, 64, 37)])
def test_linear_wgrad_with_bias_gradient(dtype, split_k, shape):
    """nn.Linear parameter gradients in one launch: dW += dY^T X and db += colsum(dY) (from the dY staging registers); exact on
    integers, for the 64- and 128-row tiles, ragged N / rows, with and without split-K."""
    import gpu_ops as G
    from bltvqg_amd._lib import check, ptr, stream_ptr, load
    Nout, Kin, Mtok = shape
    ce = 8 if dtype == torch.bfloat16 else 4
    pad = lambda n: (n + ce - 1) // ce * ce           # noqa: E731
    g = torch.Generator().manual_seed(Mtok + Nout)
    dY = torch.zeros(Mtok, pad(Nout)); dY[:, :Nout] = _ints((Mtok, Nout), -2, 2, g, torch.float32)
    X = torch.zeros(Mtok, pad(Kin)); X[:, :Kin] = _ints((Mtok, Kin), -2, 2, g, torch.float32)
    W0 = _ints((Nout, Kin), -3, 3, g, torch.float32)
    b0 = _ints((Nout,), -3, 3, g, torch.float32)
    dW, db = W0.clone().cuda(), b0.clone().cuda()
    dYd, Xd = dY.to(dtype).cuda(), X.to(dtype).cuda()
    check(load().bltvqg_linear_wgrad(G.DT[dtype], ptr(dYd), dYd.stride(0), ptr(Xd), Xd.stride(0), ptr(dW), Kin, ptr(db), Mtok, Nout, Kin,
                                     split_k, stream_ptr()), "linear_wgrad")
    torch.cuda.synchronize()
    assert torch.equal(dW.cpu().double(), dY[:, :Nout].double().t() @ X[:, :Kin].double() + W0.double())
    assert torch.equal(db.cpu().double(), dY[:, :Nout].double().sum(0) + b0.double())


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_random_accuracy(dtype):
    import gpu_ops as G
    M, N, K = 512, 384, 512
    g = torch.Generator().manual_seed(2)
    A = torch.randn(M, K, generator=g)
    Bm = torch.randn(N, K, generator=g) / math.sqrt(K)
    Ad, Bd = A.to(dtype).cuda(), Bm.to(dtype).cuda()
    got = G.gemm(Ad, Bd, M, N, K).float().cpu()
    ref = (Ad.float().cpu().double() @ Bd.float().cpu().double().t()).float()
    tol = 2e-6 if dtype == torch.float32 else 8e-3      # bf16: output rounding only (inputs already rounded)
    assert (got - ref).abs().max() <= tol * ref.abs().max()


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_dropout_matches_exported_mask(dtype):
    import gpu_ops as G
    M, N, K = 96, 40, 32
    A = torch.ones(M, K).to(dtype).cuda()
    Bm = torch.ones(N, K).to(dtype).cuda()
    p = 0.25
    got = G.gemm(A, Bm, M, N, K, drop_p=p, seed=1234, stream_id=77).float().cpu()
    mask = G.dropout_mask(1234, 77, M, N, (N + 7) // 8 * 8, p).cpu().float()
    assert torch.allclose(got, mask * K / (1 - p), rtol=1e-2 if dtype == torch.bfloat16 else 1e-6)
    keep = float(mask.mean())
    assert abs(keep - (1 - p)) < 0.03


# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("geom", [(3, 64, 7, 2, 3, 38, 34), (3, 64, 7, 2, 3, 64, 96), (3, 64, 7, 2, 3, 47, 31), (64, 64, 3, 1, 1, 14, 14), (64, 128, 3, 2, 1, 15, 13),
                                  (64, 128, 1, 2, 0, 14, 14), (256, 512, 3, 1, 1, 4, 4)])
def test_conv2d_exact_integer_and_stats(dtype, geom):
    import gpu_ops as G
    cin, cout, k, stride, pad, Hi, Wi = geom
    N = 3
    g = torch.Generator().manual_seed(cin + cout + k)
    x = _ints((N, cin, Hi, Wi), -2, 2, g, torch.float32)
    w = _ints((cout, cin, k, k), -1, 1, g, torch.float32)
    ref = F.conv2d(x.double(), w.double(), None, stride, pad).permute(0, 2, 3, 1).contiguous()
    if cin == 3:      # ResNet stem: zero-bordered NHWC4 image, weights packed [Cout, 7, 8, 4]
        Ho, Wo = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
        Hp = max(Hi + 6, 2 * (Ho - 1) + 7)
        Wp = (max(Wi + 6, 2 * (Wo - 1) + 8) + 1) // 2 * 2
        xp = G.img_pack(x.cuda(), dtype, 4, 3, 3, Hp, Wp)
        wp = G.conv_pack_w(w.cuda(), dtype, 4, 8)
        y, ssum, ssq = G.conv_stem(xp, wp, N, Hi, Wi, cout, stats=True)
    else:
        xp = G.img_pack(x.cuda(), dtype, cin)
        wp = G.conv_pack_w(w.cuda(), dtype, cin)
        y, ssum, ssq = G.conv2d(xp, wp, N, Hi, Wi, cin, cout, k, stride, pad, stats=True)
    torch.cuda.synchronize()
    got = y.float().cpu().double()
    assert got.shape == ref.shape
    assert torch.equal(got, ref.float().to(dtype).double()), (got - ref).abs().max()
    # statistics come from the fp32 accumulators (before the output rounding)
    assert torch.equal(ssum.double().sum(0).cpu(), ref.reshape(-1, cout).sum(0))
    assert torch.equal(ssq.double().sum(0).cpu(), (ref.reshape(-1, cout) ** 2).sum(0))


@pytest.mark.parametrize("geom", [(2, 224, 224), (5, 32, 56), (1, 16, 28), (3, 48, 84)])
def test_stem_with_pooled_extrema_equals_conv_bn_relu_maxpool(geom):
    """bltvqg_conv_stem_pool + bn_apply_pp(relu) == conv_stem -> bn -> relu -> maxpool(3, 2, 1) (encoder_cnn.py:17,33): the pooling
    window's max (gamma >= 0) / min (gamma < 0) of the raw convolution output commutes with the monotone BatchNorm + ReLU.  Real-valued
    data, gammas of both signs and an exact zero: the pooled tensors are bit-identical; the statistics equal the direct kernel's sums."""
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    N, Hi, Wi = geom
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(N + Hi + Wi)
    x = torch.randn(N, 3, Hi, Wi, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.1
    gamma = torch.randn(64, generator=g)
    gamma[5] = 0.0
    gamma[6] = -0.0
    beta = torch.randn(64, generator=g)
    Ho, Wo = Hi // 2, Wi // 2
    Hp, Wp = Hi + 6, Wi + 6
    assert lib.bltvqg_conv_stem_pool_ok(G.DT[dtype], Hi, Wi, Hp, Wp, 64) == 1
    xp = G.img_pack(x.cuda(), dtype, 4, 3, 3, Hp, Wp)
    wp = G.conv_pack_w(w.cuda(), dtype, 4, 8)
    # two-pass reference on the GPU: stem -> (scale, shift) -> bn + relu + maxpool into a PP tensor
    y, ssum, ssq = G.conv_stem(xp, wp, N, Hi, Wi, 64, stats=True)
    cnt = N * Ho * Wo
    mean = ssum.double().sum(0) / cnt
    var = ssq.double().sum(0) / cnt - mean * mean
    scale = (gamma.double().cuda() / torch.sqrt(var + 1e-5)).float()
    shift = (beta.double().cuda() - mean * scale.double()).float()
    Hq, Wq = Ho // 2, Wo // 2
    pbuf, pbody, pview = _pp_alloc(N, Hq, Wq, 64, dtype)
    check(lib.bltvqg_bn_relu_maxpool_pp(G.DT[dtype], ptr(y), ptr(scale), ptr(shift), ptr(pbody), N, Ho, Wo, 64, stream_ptr()), "bn_relu_maxpool_pp")
    # fused: extrema + statistics, then the affine + ReLU on the pooled tensor
    qbuf, qbody, qview = _pp_alloc(N, Hq, Wq, 64, dtype, fill=9.0)
    rows = lib.bltvqg_conv_stem_pool_stat_rows(N, Hi, Wi)
    s1 = torch.zeros(rows, 64, device="cuda"); s2 = torch.zeros(rows, 64, device="cuda")
    check(lib.bltvqg_conv_stem_pool(ptr(xp), ptr(wp), ptr(gamma.cuda()), ptr(qbody), N, Hi, Wi, Hp, Wp, ptr(s1), ptr(s2), stream_ptr()), "conv_stem_pool")
    check(lib.bltvqg_bn_apply_pp(G.DT[dtype], ptr(qbody), ptr(scale), ptr(shift), None, ptr(qbody), N, Hq, Wq, 64, 1, stream_ptr()), "bn_apply_pp")
    torch.cuda.synchronize()
    assert torch.equal(qview[:, :Hq, :Wq].float().cpu(), pview[:, :Hq, :Wq].float().cpu())
    assert float(qview[:, Hq].abs().max()) == 0.0 and float(qview[:, :, Wq].abs().max()) == 0.0
    a, b = s1.double().sum(0).cpu(), ssum.double().sum(0).cpu()
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-3), float((a - b).abs().max())
    a, b = s2.double().sum(0).cpu(), ssq.double().sum(0).cpu()
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-3), float((a - b).abs().max())
    # and against torch end to end (bf16 tolerance)
    conv = F.conv2d(x.to(dtype).float(), w.to(dtype).float(), None, 2, 3)
    ref = F.max_pool2d(torch.relu(conv * scale.cpu().view(1, -1, 1, 1) + shift.cpu().view(1, -1, 1, 1)), 3, 2, 1).permute(0, 2, 3, 1)
    assert (qview[:, :Hq, :Wq].float().cpu() - ref).abs().max() < 3e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES)
def test_batchnorm2d_train_pipeline(dtype):
    """conv statistics -> finalize (scale/shift + running stats) -> apply(+residual, relu) / relu+maxpool / avgpool."""
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    N, C, Hh, Ww = 4, 64, 12, 10
    g = torch.Generator().manual_seed(3)
    x = torch.randn(N, C, Hh, Ww, generator=g) * 2 + 0.5
    res = torch.randn(N, C, Hh, Ww, generator=g)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    rm, rv = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y_ref = F.batch_norm(x, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    xh = x.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    xf = xh.float().reshape(-1, C)
    nparts = 6
    parts = xf.chunk(nparts, 0)
    psum = torch.stack([p.sum(0) for p in parts]).contiguous()
    psq = torch.stack([(p * p).sum(0) for p in parts]).contiguous()
    scale = torch.empty(C, device="cuda"); shift = torch.empty(C, device="cuda")
    rmd, rvd = rm.cuda(), rv.cuda()
    scratch = torch.zeros(lib.bltvqg_bn_scratch_doubles(C), dtype=torch.float64, device="cuda")   # ticket counters start at zero
    check(lib.bltvqg_bn_finalize(ptr(psum), ptr(psq), nparts, C, xf.shape[0], ptr(gamma.cuda()), ptr(beta.cuda()), 1e-5, 0.1, ptr(rmd), ptr(rvd),
                                 ptr(scale), ptr(shift), ptr(scratch), stream_ptr()), "bn_finalize")
    tol = 1e-4 if dtype == torch.float32 else 3e-2
    assert torch.allclose(rmd.cpu(), rm_ref, atol=tol) and torch.allclose(rvd.cpu(), rv_ref, atol=tol)
    y = torch.empty_like(xh)
    resh = res.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    check(lib.bltvqg_bn_apply(G.DT[dtype], ptr(xh), ptr(scale), ptr(shift), ptr(resh), ptr(y), xf.shape[0], C, 1, stream_ptr()), "bn_apply")
    ref = torch.relu(y_ref + res).permute(0, 2, 3, 1)
    assert torch.allclose(y.float().cpu(), ref, atol=tol * 4, rtol=tol)
    # relu + maxpool 3x3/2 pad 1
    Ho, Wo = (Hh - 1) // 2 + 1, (Ww - 1) // 2 + 1
    yp = torch.empty(N, Ho, Wo, C, dtype=dtype, device="cuda")
    check(lib.bltvqg_bn_relu_maxpool(G.DT[dtype], ptr(xh), ptr(scale), ptr(shift), ptr(yp), N, Hh, Ww, C, stream_ptr()), "maxpool")
    refp = F.max_pool2d(torch.relu(y_ref), 3, 2, 1).permute(0, 2, 3, 1)
    assert torch.allclose(yp.float().cpu(), refp, atol=tol * 4, rtol=tol)
    # global average pool
    ya = torch.empty(N, C, dtype=dtype, device="cuda")
    check(lib.bltvqg_avgpool(G.DT[dtype], ptr(xh), ptr(ya), N, Hh * Ww, C, stream_ptr()), "avgpool")
    assert torch.allclose(ya.float().cpu(), xh.float().cpu().mean(dim=(1, 2)), atol=tol)


@pytest.mark.parametrize("dtype", DTYPES)
def test_batchnorm1d_fwd_bwd(dtype):
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    B, C = 24, 72
    g = torch.Generator().manual_seed(4)
    x = (torch.randn(B, C, generator=g) * 1.5 + 0.3).to(dtype)
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    dy = torch.randn(B, C, generator=g).to(dtype)
    rm, rv = torch.zeros(C), torch.ones(C)
    xr = x.float().clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True); br = beta.clone().requires_grad_(True)
    y_ref = F.batch_norm(xr, rm, rv, gr, br, True, 0.01, 1e-5)
    y_ref.backward(dy.float())
    xd = x.cuda(); y = torch.empty_like(xd)
    mean = torch.empty(C, device="cuda"); rstd = torch.empty(C, device="cuda")
    rmd, rvd = torch.zeros(C, device="cuda"), torch.ones(C, device="cuda")
    check(lib.bltvqg_bn1d_fwd(G.DT[dtype], ptr(xd), ptr(gamma.cuda()), ptr(beta.cuda()), ptr(y), ptr(mean), ptr(rstd), ptr(rmd), ptr(rvd), B, C,
                              1e-5, 0.01, stream_ptr()), "bn1d_fwd")
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.allclose(y.float().cpu(), y_ref.detach(), atol=tol * 4, rtol=tol)
    assert torch.allclose(rmd.cpu(), rm, atol=1e-5) and torch.allclose(rvd.cpu(), rv, atol=1e-5)
    dx = torch.empty_like(xd); dg = torch.empty(C, device="cuda"); db = torch.empty(C, device="cuda")
    check(lib.bltvqg_bn1d_bwd(G.DT[dtype], ptr(dy.cuda()), ptr(xd), ptr(gamma.cuda()), ptr(mean), ptr(rstd), ptr(dx), ptr(dg), ptr(db), B, C,
                              stream_ptr()), "bn1d_bwd")
    assert torch.allclose(dx.float().cpu(), xr.grad, atol=tol * 4, rtol=tol * 4)
    assert torch.allclose(dg.cpu(), gr.grad, atol=tol * 10, rtol=tol) and torch.allclose(db.cpu(), br.grad, atol=tol * 10, rtol=tol)


# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape", [(37, 64), (5120, 512), (9, 1024), (130, 256)])
def test_layernorm_fwd_bwd(dtype, shape):
    import gpu_ops as G
    rows, cols = shape
    g = torch.Generator().manual_seed(rows + cols)
    x = (torch.randn(rows, cols, generator=g) * 2 + 1).to(dtype)
    gamma, beta = torch.rand(cols, generator=g) + 0.5, torch.randn(cols, generator=g)
    dy = torch.randn(rows, cols, generator=g).to(dtype)
    dres = torch.randn(rows, cols, generator=g).to(dtype)
    xr = x.float().clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = F.layer_norm(xr, (cols,), gr, br, 1e-5)
    y_ref.backward(dy.float())
    y, mean, rstd = G.layernorm_fwd(x.cuda(), gamma.cuda(), beta.cuda())
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.allclose(y.float().cpu(), y_ref.detach(), atol=tol * 4, rtol=tol)
    dx, dg, db = G.layernorm_bwd(dy.cuda(), x.cuda(), gamma.cuda(), mean, rstd, dres.cuda())
    assert torch.allclose(dx.float().cpu(), xr.grad + dres.float(), atol=tol * 8, rtol=tol * 4)
    gs = max(1.0, float(gr.grad.abs().max()))
    assert (dg.cpu() - gr.grad).abs().max() < tol * 4 * gs and (db.cpu() - br.grad).abs().max() < tol * 4 * gs
    # in-place residual form used by the engine: dres and dx are the same buffer
    buf = dres.cuda().clone()
    from gpu_ops import check, ptr, stream_ptr
    dg2 = torch.zeros(cols, device="cuda"); db2 = torch.zeros(cols, device="cuda")
    check(G.lib().bltvqg_layernorm_bwd(G.DT[dtype], ptr(dy.cuda()), ptr(x.cuda()), ptr(gamma.cuda()), ptr(mean), ptr(rstd), ptr(buf), ptr(buf),
                                       ptr(dg2), ptr(db2), rows, cols, stream_ptr()), "ln_bwd")
    assert torch.equal(buf, dx)


# --------------------------------------------------------------------------------------------------------------
def _attn_ref(Q, K, V, key_ids, B, h, Tq, Tk, d, causal, scale, keep=None, p=0.0):
    """Q [B*Tq, h*d] etc. fp32 tensors with requires_grad; mirrors transformer_layers.py:494-526."""
    q = Q.view(B, Tq, h, d).permute(0, 2, 1, 3) * scale
    k = K.view(B, Tk, h, d).permute(0, 2, 1, 3)
    v = V.view(B, Tk, h, d).permute(0, 2, 1, 3)
    logits = q @ k.transpose(-1, -2)
    mask = key_ids.eq(0).view(B, 1, 1, Tk).expand(B, 1, Tq, Tk)
    if causal:
        mask = mask | torch.triu(torch.ones(Tq, Tk, dtype=torch.bool), 1).view(1, 1, Tq, Tk)
    logits = logits.masked_fill(mask, -1e18)
    w = torch.softmax(logits, -1)
    if keep is not None:
        w = w * keep / (1 - p)
    return (w @ v).permute(0, 2, 1, 3).reshape(B * Tq, h * d)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("case", [dict(B=3, h=4, Tq=5, Tk=5, d=16, causal=False), dict(B=2, h=8, Tq=20, Tk=20, d=64, causal=True),
                                  dict(B=4, h=4, Tq=20, Tk=5, d=32, causal=False), dict(B=2, h=2, Tq=21, Tk=21, d=64, causal=False)])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_attention_fwd_bwd(dtype, case, p):
    import gpu_ops as G
    B, h, Tq, Tk, d, causal = case["B"], case["h"], case["Tq"], case["Tk"], case["d"], case["causal"]
    Hd = h * d
    g = torch.Generator().manual_seed(B * 100 + Tq)
    packed = Tq == Tk
    if packed:      # self-attention reads Q, K, V out of one [M, 3H] projection buffer
        qkv = torch.randn(B * Tq, 3 * Hd, generator=g).to(dtype)
        Q, K, V = qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:]
    else:
        Q = torch.randn(B * Tq, Hd, generator=g).to(dtype)
        kv = torch.randn(B * Tk, 2 * Hd, generator=g).to(dtype)
        K, V = kv[:, :Hd], kv[:, Hd:]
    ids = torch.randint(1, 50, (B, Tk), generator=g, dtype=torch.int32)
    ids[0, Tk - 2:] = 0
    if not causal:
        ids[1, :] = 0          # a fully masked key row -> uniform attention, no NaN (masked value is -1e18, not -inf)
    dO = torch.randn(B * Tq, Hd, generator=g).to(dtype)
    scale = d ** -0.5
    seed, sid = 99, 1003
    keep = None
    if p > 0:
        keep = G.dropout_mask(seed, sid, B * h * Tq, Tk, Tk, p).cpu().float().view(B, h, Tq, Tk)
    Qr, Kr, Vr = [t.float().clone().requires_grad_(True) for t in (Q, K, V)]
    ref = _attn_ref(Qr, Kr, Vr, ids, B, h, Tq, Tk, d, causal, scale, keep, p)
    ref.backward(dO.float())
    if packed:
        qd = qkv.cuda()
        Qd, Kd, Vd = qd[:, :Hd], qd[:, Hd:2 * Hd], qd[:, 2 * Hd:]
    else:
        Qd = Q.cuda(); kd = kv.cuda(); Kd, Vd = kd[:, :Hd], kd[:, Hd:]
    O = G.attn_fwd(Qd, Kd, Vd, ids.cuda(), B, h, Tq, Tk, d, causal, scale, p, seed, sid)
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.isfinite(O.float()).all()
    assert torch.allclose(O.float().cpu(), ref.detach(), atol=tol * 4, rtol=tol)
    dQ, dK, dV = G.attn_bwd(Qd, Kd, Vd, dO.cuda(), ids.cuda(), B, h, Tq, Tk, d, causal, scale, p, seed, sid)
    for got, want in ((dQ, Qr.grad), (dK, Kr.grad), (dV, Vr.grad)):
        assert torch.allclose(got.float().cpu(), want, atol=tol * 8, rtol=tol * 4)


@pytest.mark.parametrize("case", [dict(B=5, h=3, Tq=7, Tk=21, d=64, causal=0), dict(B=2, h=2, Tq=32, Tk=32, d=128, causal=1),
                                  dict(B=3, h=1, Tq=1, Tk=1, d=32, causal=0), dict(B=9, h=8, Tq=20, Tk=21, d=64, causal=0),
                                  dict(B=4, h=4, Tq=17, Tk=17, d=64, causal=2), dict(B=256, h=8, Tq=20, Tk=20, d=64, causal=1)])
@pytest.mark.parametrize("p", [0.0, 0.3])
def test_attention_mfma_form_matches_reference_and_valu_form(case, p):
    """bf16 attention runs as one wave per (batch, head) on MFMA tiles (attn.hip, `attn_*_mfma_kernel`); debug key 16 selects the
    VALU kernels it replaced.  Both against the fp32 torch restatement of transformer_layers.py:494-526 with the SAME dropout mask,
    and against each other (the dropout stream is addressed identically, so the two forms drop the same weights)."""
    import gpu_ops as G
    B, h, Tq, Tk, d, causal = case["B"], case["h"], case["Tq"], case["Tk"], case["d"], case["causal"]
    Hd = h * d
    g = torch.Generator().manual_seed(B * 7 + Tq * 3 + Tk)
    Q = torch.randn(B * Tq, Hd, generator=g).bfloat16()
    kv = torch.randn(B * Tk, 2 * Hd, generator=g).bfloat16()
    K, V = kv[:, :Hd], kv[:, Hd:]
    ids = torch.randint(1, 50, (B, Tk), generator=g, dtype=torch.int32)
    if Tk > 2:
        ids[0, Tk - 2:] = 0
    if causal == 0 and B > 1:
        ids[1, :] = 0
    dO = torch.randn(B * Tq, Hd, generator=g).bfloat16()
    scale = d ** -0.5
    seed, sid = 1234567, 2011
    keep = None
    if p > 0:
        keep = G.dropout_mask(seed, sid, B * h * Tq, Tk, Tk, p).cpu().float().view(B, h, Tq, Tk)
    Qd = Q.cuda(); kd = kv.cuda(); Kd, Vd = kd[:, :Hd], kd[:, Hd:]
    res = {}
    for form in (0, 1):
        G.lib().bltvqg_debug_set(16, form)
        try:
            O = G.attn_fwd(Qd, Kd, Vd, ids.cuda(), B, h, Tq, Tk, d, causal, scale, p, seed, sid)
            dQ, dK, dV = G.attn_bwd(Qd, Kd, Vd, dO.cuda(), ids.cuda(), B, h, Tq, Tk, d, causal, scale, p, seed, sid)
        finally:
            G.lib().bltvqg_debug_set(16, 0)
        res[form] = [t.float().cpu() for t in (O, dQ, dK, dV)]
        assert all(torch.isfinite(t).all() for t in res[form])
    for a_, b_ in zip(res[0], res[1]):
        assert torch.allclose(a_, b_, atol=6e-2, rtol=4e-2), float((a_ - b_).abs().max())
    if causal != 2:       # causal == 2 is the greedy decoder's prefix form (no torch restatement here; compared against the VALU form above)
        Qr, Kr, Vr = [t.float().clone().requires_grad_(True) for t in (Q, K, V)]
        ref = _attn_ref(Qr, Kr, Vr, ids, B, h, Tq, Tk, d, bool(causal), scale, keep, p)
        ref.backward(dO.float())
        tol = 2e-2
        assert torch.allclose(res[0][0], ref.detach(), atol=tol * 4, rtol=tol)
        for got, want in zip(res[0][1:], (Qr.grad, Kr.grad, Vr.grad)):
            assert torch.allclose(got, want, atol=tol * 8, rtol=tol * 4), float((got - want).abs().max())


@pytest.mark.parametrize("case", [dict(B=5, h=4, Tq=20, Tk=20, causal=1), dict(B=9, h=8, Tq=20, Tk=5, causal=0), dict(B=256, h=8, Tq=21, Tk=21, causal=0),
                                  dict(B=3, h=8, Tq=32, Tk=32, causal=1), dict(B=4, h=2, Tq=1, Tk=3, causal=0)])
@pytest.mark.parametrize("p", [0.0, 0.2])
@pytest.mark.parametrize("residual", [True, False])
def test_fused_attention_output_projection_is_bit_identical_to_the_two_launch_path(case, p, residual):
    """bltvqg_attn_out_fwd (attention core + output Linear + residual, one workgroup per batch element) against bltvqg_attn_fwd followed by
    the planned-tile GEMM with the residual epilogue (transformer_layers.py:494-532 + the sub-layer's `x +`): the context AND the sub-layer
    output must be BIT-identical — same MFMA instruction, same ascending k order, same Philox stream for the attention dropout."""
    import gpu_ops as G
    B, h, Tq, Tk, causal = case["B"], case["h"], case["Tq"], case["Tk"], case["causal"]
    d, Hd = 64, case["h"] * 64
    g = torch.Generator().manual_seed(B * 11 + Tq * 5 + Tk)
    Q = torch.randn(B * Tq, Hd, generator=g).bfloat16().cuda()
    kv = torch.randn(B * Tk, 2 * Hd, generator=g).bfloat16().cuda()
    K, V = kv[:, :Hd], kv[:, Hd:]
    Wo = (torch.randn(Hd, Hd, generator=g) * Hd ** -0.5).bfloat16().cuda()
    R = torch.randn(B * Tq, Hd, generator=g).bfloat16().cuda() if residual else None
    ids = torch.randint(1, 50, (B, Tk), generator=g, dtype=torch.int32)
    if Tk > 2:
        ids[0, Tk - 2:] = 0
    ids = ids.cuda()
    scale = d ** -0.5
    O2 = G.attn_fwd(Q, K, V, ids, B, h, Tq, Tk, d, causal, scale, p, 99, 31)
    Y2 = G.gemm(O2, Wo, B * Tq, Hd, Hd, R=R)
    O1, Y1 = G.attn_out_fwd(Q, K, V, Wo, R, ids, B, h, Tq, Tk, d, causal, scale, p, 99, 31)
    assert torch.isfinite(Y1.float()).all()
    assert torch.equal(O1, O2)
    assert torch.equal(Y1, Y2), float((Y1.float() - Y2.float()).abs().max())


# --------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("V", [97, 8000])
def test_cross_entropy_and_bow(dtype, V):
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    B, T = 6, 20
    M = B * T
    ld = (V + 7) // 8 * 8
    g = torch.Generator().manual_seed(V)
    logits = (torch.randn(M, V, generator=g) * 3).to(dtype)
    target = torch.randint(1, V, (B, T), generator=g)
    target[:, 15:] = 0
    target[2, 3:] = 0
    count = torch.tensor([float((target != 0).sum())])
    lr = logits.float().clone().requires_grad_(True)
    loss_ref = F.cross_entropy(lr, target.reshape(-1), ignore_index=0)
    loss_ref.backward()
    buf = torch.zeros(M, ld, dtype=dtype, device="cuda"); buf[:, :V] = logits.cuda()
    buf[:, V:] = 7.0       # garbage in the pad columns must be zeroed by the gradient write
    loss = torch.zeros(1, device="cuda")
    check(lib.bltvqg_ce_fwd_bwd(G.DT[dtype], ptr(buf), ld, ptr(target.reshape(-1).int().cuda()), M, V, ptr(count.cuda()), 1.0, ptr(loss), 1,
                                stream_ptr()), "ce")
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert abs(float(loss) - float(loss_ref)) < tol * 10
    assert torch.allclose(buf[:, :V].float().cpu(), lr.grad, atol=tol / 5, rtol=tol * 5)
    assert float(buf[:, V:].float().abs().sum()) == 0.0
    # bag-of-words CE (train_iq.py:92-94)
    z = (torch.randn(B, V, generator=g) * 2).to(dtype)
    zr = z.float().clone().requires_grad_(True)
    rep = zr.unsqueeze(1).repeat(1, T, 1)
    aux_ref = F.cross_entropy(rep.reshape(-1, V), target.reshape(-1), ignore_index=0)
    aux_ref.backward()
    zb = torch.zeros(B, ld, dtype=dtype, device="cuda"); zb[:, :V] = z.cuda()
    dz = torch.empty_like(zb)
    aux = torch.zeros(1, device="cuda")
    check(lib.bltvqg_bow_ce_fwd_bwd(G.DT[dtype], ptr(zb), ld, ptr(target.reshape(-1).int().cuda()), B, T, V, ptr(count.cuda()), 1.0, ptr(aux),
                                    ptr(dz), stream_ptr()), "bow")
    assert abs(float(aux) - float(aux_ref)) < tol * 10
    assert torch.allclose(dz[:, :V].float().cpu(), zr.grad, atol=tol / 2, rtol=tol * 5)


@pytest.mark.parametrize("dtype", DTYPES)
def test_latent_and_mse(dtype):
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    from oracle.iq_oracle import gaussian_kld
    lib = G.lib()
    B, Z = 10, 64
    g = torch.Generator().manual_seed(8)
    mlvp = (torch.randn(B, 2 * Z, generator=g) * 0.7).to(dtype)
    mlvq = (torch.randn(B, 2 * Z, generator=g) * 0.7).to(dtype)
    eps = torch.randn(B, Z, generator=g)
    dz = torch.randn(B, Z, generator=g).to(dtype)
    pr, qr = mlvp.float().clone().requires_grad_(True), mlvq.float().clone().requires_grad_(True)
    kld_ref = gaussian_kld(qr[:, :Z], qr[:, Z:], pr[:, :Z], pr[:, Z:]).mean()
    z_ref = eps * torch.exp(0.5 * qr[:, Z:]) + qr[:, :Z]
    gk = 0.37
    (gk * kld_ref + (z_ref * dz.float()).sum()).backward()
    z = torch.empty(B, Z, dtype=dtype, device="cuda"); kld = torch.zeros(1, device="cuda")
    check(lib.bltvqg_latent_fwd(G.DT[dtype], ptr(mlvp.cuda()), ptr(mlvq.cuda()), ptr(eps.cuda()), ptr(z), ptr(kld), B, Z, 2 * Z, stream_ptr()), "latent_fwd")
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert torch.allclose(z.float().cpu(), z_ref.detach(), atol=tol * 4, rtol=tol)
    assert abs(float(kld) - float(kld_ref)) < max(tol, 1e-4) * max(1.0, abs(float(kld_ref)))
    dp = torch.empty(B, 2 * Z, dtype=dtype, device="cuda"); dq = torch.empty(B, 2 * Z, dtype=dtype, device="cuda")
    check(lib.bltvqg_latent_bwd(G.DT[dtype], ptr(mlvp.cuda()), ptr(mlvq.cuda()), ptr(eps.cuda()), ptr(dz.cuda()), gk, ptr(dp), ptr(dq), B, Z, 2 * Z,
                                stream_ptr()), "latent_bwd")
    assert torch.allclose(dp.float().cpu(), pr.grad, atol=tol * 2, rtol=tol * 4)
    assert torch.allclose(dq.float().cpu(), qr.grad, atol=tol * 2, rtol=tol * 4)
    # MSE with gradient to both arguments (train_iq.py:84)
    a, b = torch.randn(B, 72, generator=g).to(dtype), torch.randn(B, 72, generator=g).to(dtype)
    ar, br = a.float().clone().requires_grad_(True), b.float().clone().requires_grad_(True)
    l_ref = F.mse_loss(ar, br); (0.1 * l_ref).backward()
    da, db = torch.empty_like(a, device="cuda"), torch.empty_like(b, device="cuda")
    l = torch.zeros(1, device="cuda")
    check(lib.bltvqg_mse_fwd_bwd(G.DT[dtype], ptr(a.cuda()), ptr(b.cuda()), B * 72, 0.1, ptr(l), ptr(da), ptr(db), stream_ptr()), "mse")
    assert abs(float(l) - float(l_ref)) < 1e-5 * max(1.0, float(l_ref))
    assert torch.allclose(da.float().cpu(), ar.grad, atol=tol / 10, rtol=tol) and torch.allclose(db.float().cpu(), br.grad, atol=tol / 10, rtol=tol)


@pytest.mark.parametrize("dtype", DTYPES)
def test_embedding_gather_scatter(dtype):
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    V, E, rows, ld = 50, 20, 300, 32
    g = torch.Generator().manual_seed(9)
    table = torch.randn(V, E, generator=g)
    ids = torch.randint(0, V, (rows,), generator=g, dtype=torch.int32)
    out = torch.full((rows, ld), 5.0, dtype=dtype, device="cuda")
    check(lib.bltvqg_embed_gather(G.DT[dtype], ptr(table.cuda()), ptr(ids.cuda()), ptr(out), rows, E, ld, stream_ptr()), "gather")
    assert torch.equal(out[:, :E].float().cpu(), table[ids.long()].to(dtype).float())
    assert float(out[:, E:].float().abs().sum()) == 0.0
    d = _ints((rows, ld), -2, 2, g, torch.float32).to(dtype)
    dtab = torch.zeros(V, E, device="cuda")
    check(lib.bltvqg_embed_scatter(G.DT[dtype], ptr(d.cuda()), ld, ptr(ids.cuda()), ptr(dtab), rows, E, 0, stream_ptr()), "scatter")
    ref = torch.zeros(V, E)
    ref.index_add_(0, ids.long(), d[:, :E].float())
    ref[0] = 0          # padding_idx row receives no gradient (iq.py:72)
    assert torch.equal(dtab.cpu(), ref)


def test_adam_with_clip_matches_torch():
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    n = 10007
    g = torch.Generator().manual_seed(10)
    p0 = torch.randn(n, generator=g)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-3)
    p = p0.cuda(); m = torch.zeros(n, device="cuda"); v = torch.zeros(n, device="cuda")
    for step in range(1, 4):
        grad = torch.randn(n, generator=g) * (3.0 if step == 2 else 0.01)
        pr.grad = grad.clone()
        torch.nn.utils.clip_grad_norm_([pr], 5.0)
        lr = 1e-3 * step
        for gp in opt.param_groups:
            gp["lr"] = lr
        opt.step()
        gd = grad.cuda()
        nsq = torch.zeros(1, device="cuda")
        check(lib.bltvqg_sumsq(ptr(gd), n, ptr(nsq), stream_ptr()), "sumsq")
        assert abs(float(nsq) - float((grad.double() ** 2).sum())) < 1e-4 * float((grad.double() ** 2).sum())
        check(lib.bltvqg_adam_step(ptr(p), ptr(gd), ptr(m), ptr(v), n, ptr(nsq), 5.0, lr, 0.9, 0.999, 1e-8, step, stream_ptr()), "adam")
        assert torch.allclose(p.cpu(), pr.detach(), atol=2e-6, rtol=1e-5)


# ---------------------------------------------------------------------------------------------------------------
# padded-pitch (PP) activations: [N][H+1][W+1][C] with zero pad pixels + guards (include/bltvqg_hip.h)
# ---------------------------------------------------------------------------------------------------------------
def _pp_alloc(N, H, W, C, dtype, fill=0.0):
    import gpu_ops as G
    lib = G.lib()
    gf, gt, n = lib.bltvqg_pp_guard_front(), lib.bltvqg_pp_guard_tail(), lib.bltvqg_pp_pixels(N, H, W)
    buf = torch.zeros(gf + n + gt, C, dtype=dtype, device="cuda")
    body = buf[gf:gf + n]
    if fill:
        body.fill_(fill)
    return buf, body, body.view(N, H + 1, W + 1, C)


def _pp_pack(x_nhwc, dtype):
    N, H, W, C = x_nhwc.shape
    buf, body, view = _pp_alloc(N, H, W, C, dtype)
    view[:, :H, :W] = x_nhwc.to(dtype).cuda()
    return buf, body


@pytest.mark.parametrize("geom", [(64, 64, 56, 56, 2), (64, 64, 56, 56, 13), (64, 64, 24, 40, 37), (128, 128, 28, 28, 3), (256, 256, 14, 14, 2), (512, 512, 7, 7, 5),
                                  (64, 128, 9, 13, 1), (128, 64, 20, 33, 2), (192, 64, 5, 62, 1)])
def test_conv3x3_pp_exact_integer_and_stats(geom):
    """The LDS-patch 3x3 convolution (csrc/conv_pp.hip) against F.conv2d on integer data: exact outputs at every real pixel, exact
    BatchNorm partial sums with the pad positions masked out; the guards and the tile overhang stay untouched."""
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    cin, cout, H, W, N = geom
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(cin + cout + H)
    x = _ints((N, cin, H, W), -2, 2, g, torch.float32)
    w = _ints((cout, cin, 3, 3), -1, 1, g, torch.float32)
    ref = F.conv2d(x.double(), w.double(), None, 1, 1).permute(0, 2, 3, 1).contiguous()
    xbuf, xbody = _pp_pack(x.permute(0, 2, 3, 1).contiguous(), dtype)
    wp = G.conv_pack_w(w.cuda(), dtype, cin)
    ybuf, ybody, yview = _pp_alloc(N, H, W, cout, dtype, fill=7.0)
    rows = lib.bltvqg_conv3x3_pp_stat_rows(N, H, W)
    # debug key 19: 0 = default plan (256-position tiles for Cout = 64, 128-position ones otherwise), 1 = 128 positions everywhere,
    # 2 = 256 positions also for Cout % 128 == 0
    for form in ((0, 2) if cout % 128 == 0 else (0, 1)):
        ybody.fill_(7.0)
        ssum = torch.zeros(rows, cout, device="cuda")
        ssq = torch.zeros(rows, cout, device="cuda")
        lib.bltvqg_debug_set(19, form)
        try:
            check(lib.bltvqg_conv3x3_pp(ptr(xbody), ptr(wp), ptr(ybody), N, H, W, cin, cout, ptr(ssum), ptr(ssq), stream_ptr()), "conv3x3_pp")
            torch.cuda.synchronize()
        finally:
            lib.bltvqg_debug_set(19, 0)
        got = yview[:, :H, :W].float().cpu().double()
        assert torch.equal(got, ref.float().to(dtype).double()), (form, (got - ref).abs().max())
        assert torch.equal(ssum.double().sum(0).cpu(), ref.reshape(-1, cout).sum(0))
        assert torch.equal(ssq.double().sum(0).cpu(), (ref.reshape(-1, cout) ** 2).sum(0))
    gf = lib.bltvqg_pp_guard_front()
    assert float(ybuf[:gf].abs().max()) == 0.0 and float(ybuf[gf + ybody.shape[0]:].abs().max()) == 0.0      # guards untouched
    # without statistics (eval-mode BatchNorm path)
    ybody.fill_(7.0)
    check(lib.bltvqg_conv3x3_pp(ptr(xbody), ptr(wp), ptr(ybody), N, H, W, cin, cout, None, None, stream_ptr()), "conv3x3_pp")
    torch.cuda.synchronize()
    assert torch.equal(yview[:, :H, :W].float().cpu().double(), ref.float().to(dtype).double())


@pytest.mark.parametrize("geom", [(64, 64, 56, 56, 2), (64, 64, 24, 40, 37), (128, 128, 28, 28, 3), (256, 256, 14, 14, 2), (512, 512, 7, 7, 5),
                                  (64, 128, 9, 13, 1), (192, 64, 5, 62, 1), (512, 512, 7, 7, 64)])
def test_conv3x3_pp_with_input_batchnorm_relu_fused(geom):
    """bltvqg_conv3x3_pp_bn_relu_in == bltvqg_conv3x3_pp o bltvqg_bn_apply_pp(relu): the BatchNorm + ReLU of the previous convolution
    applied on the staged patch in LDS.  The raw input carries GARBAGE (NaN / huge values) at its pad positions and in both guards, as
    a convolution leaves them: they must read as the zero padding.  Bit-identical outputs and statistics (same arithmetic per element)."""
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    cin, cout, H, W, N = geom
    dtype = torch.bfloat16
    g = torch.Generator().manual_seed(cin + cout + H + N)
    x = torch.randn(N, H, W, cin, generator=g) * 2
    w = torch.randn(cout, cin, 3, 3, generator=g) * 0.05
    scale = torch.randn(cin, generator=g)            # both signs, and one exact zero
    scale[3] = 0.0
    shift = torch.randn(cin, generator=g)
    xbuf, xbody, xview = _pp_alloc(N, H, W, cin, dtype, fill=float("nan"))
    xbuf[:lib.bltvqg_pp_guard_front()] = 3.0e38
    xbuf[lib.bltvqg_pp_guard_front() + xbody.shape[0]:] = float("nan")
    xview[:, :H, :W] = x.to(dtype).cuda()
    wp = G.conv_pack_w(w.cuda(), dtype, cin)
    rows = lib.bltvqg_conv3x3_pp_stat_rows(N, H, W)
    res = []
    for fused in (True, False):
        ybuf, ybody, yview = _pp_alloc(N, H, W, cout, dtype, fill=7.0)
        ssum = torch.zeros(rows, cout, device="cuda")
        ssq = torch.zeros(rows, cout, device="cuda")
        if fused:
            check(lib.bltvqg_conv3x3_pp_bn_relu_in(ptr(xbody), ptr(scale.cuda()), ptr(shift.cuda()), ptr(wp), ptr(ybody), N, H, W, cin, cout, ptr(ssum),
                                                   ptr(ssq), stream_ptr()), "conv3x3_pp_bn_relu_in")
        else:
            abuf, abody, aview = _pp_alloc(N, H, W, cin, dtype)
            check(lib.bltvqg_bn_apply_pp(G.DT[dtype], ptr(xbody), ptr(scale.cuda()), ptr(shift.cuda()), None, ptr(abody), N, H, W, cin, 1, stream_ptr()),
                  "bn_apply_pp")
            check(lib.bltvqg_conv3x3_pp(ptr(abody), ptr(wp), ptr(ybody), N, H, W, cin, cout, ptr(ssum), ptr(ssq), stream_ptr()), "conv3x3_pp")
        torch.cuda.synchronize()
        res.append((yview[:, :H, :W].float().cpu(), ssum.cpu(), ssq.cpu()))
    assert torch.isfinite(res[0][0]).all()
    assert torch.equal(res[0][0], res[1][0]), float((res[0][0] - res[1][0]).abs().max())
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
    # and against torch on the bf16-rounded activation
    act = torch.relu(x.to(dtype).float() * scale + shift).to(dtype).float()
    ref = F.conv2d(act.permute(0, 3, 1, 2).double(), w.to(dtype).double(), None, 1, 1).permute(0, 2, 3, 1)
    assert (res[0][0].double() - ref).abs().max() < 2e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("geom", [(64, 128, 3, 2, 1, 20, 20, 1, 1), (64, 128, 1, 2, 0, 20, 20, 1, 1), (64, 64, 3, 1, 1, 12, 10, 1, 1),
                                  (64, 64, 3, 1, 1, 12, 10, 0, 1), (64, 64, 3, 2, 1, 11, 13, 1, 0), (128, 256, 3, 2, 1, 28, 28, 1, 1)])
def test_conv2d_pp_layouts(dtype, geom):
    """The implicit-GEMM convolution reading / writing padded-pitch buffers (stride-2 and 1x1 convolutions of the ResNet stack, every
    convolution in fp32 mode): exact outputs, zeros at the pad positions of a PP output, exact statistics."""
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    cin, cout, k, stride, pad, Hi, Wi, in_pp, out_pp = geom
    N = 3
    g = torch.Generator().manual_seed(cin + cout + k + Hi)
    x = _ints((N, cin, Hi, Wi), -2, 2, g, torch.float32)
    w = _ints((cout, cin, k, k), -1, 1, g, torch.float32)
    ref = F.conv2d(x.double(), w.double(), None, stride, pad).permute(0, 2, 3, 1).contiguous()
    Ho, Wo = ref.shape[1], ref.shape[2]
    xh = x.permute(0, 2, 3, 1).contiguous()
    if in_pp:
        xbuf, xin = _pp_pack(xh, dtype)
    else:
        xin = xh.to(dtype).cuda()
    wp = G.conv_pack_w(w.cuda(), dtype, cin)
    if out_pp:
        ybuf, yout, yview = _pp_alloc(N, Ho, Wo, cout, dtype, fill=7.0)
    else:
        yout = torch.full((N, Ho, Wo, cout), 7.0, dtype=dtype, device="cuda")
        yview = yout
    rows = lib.bltvqg_conv2d_pp_stat_rows(G.DT[dtype], N, Hi, Wi, cin, cout, k, k, stride, pad, out_pp)
    ssum = torch.zeros(rows, cout, device="cuda")
    ssq = torch.zeros(rows, cout, device="cuda")
    check(lib.bltvqg_conv2d_pp(G.DT[dtype], ptr(xin), ptr(wp), ptr(yout), N, Hi, Wi, cin, cout, k, k, stride, pad, in_pp, out_pp, ptr(ssum),
                               ptr(ssq), stream_ptr()), "conv2d_pp")
    torch.cuda.synchronize()
    got = yview[:, :Ho, :Wo].float().cpu().double()
    assert torch.equal(got, ref.float().to(dtype).double()), (got - ref).abs().max()
    if out_pp:
        assert float(yview[:, Ho].abs().max()) == 0.0 and float(yview[:, :, Wo].abs().max()) == 0.0
    assert torch.equal(ssum.double().sum(0).cpu(), ref.reshape(-1, cout).sum(0))
    assert torch.equal(ssq.double().sum(0).cpu(), (ref.reshape(-1, cout) ** 2).sum(0))


@pytest.mark.parametrize("dtype", DTYPES)
def test_padded_pitch_elementwise_ops(dtype):
    """bn_apply_pp (zeros at pad positions whatever the input holds there), bn_relu_maxpool_pp (PP output), avgpool_pp."""
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    N, H, W, C = 3, 6, 5, 64
    g = torch.Generator().manual_seed(11)
    x = torch.randn(N, H, W, C, generator=g)
    r = torch.randn(N, H, W, C, generator=g)
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g)
    xbuf, xbody, xview = _pp_alloc(N, H, W, C, dtype, fill=3.0)          # garbage at the pad positions, as a convolution leaves them
    xview[:, :H, :W] = x.to(dtype).cuda()
    rbuf, rbody = _pp_pack(r, dtype)
    check(lib.bltvqg_bn_apply_pp(G.DT[dtype], ptr(xbody), ptr(scale.cuda()), ptr(shift.cuda()), ptr(rbody), ptr(xbody), N, H, W, C, 1,
                                 stream_ptr()), "bn_apply_pp")
    torch.cuda.synchronize()
    ref = torch.relu(x.to(dtype).float() * scale + shift + r.to(dtype).float())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert (xview[:, :H, :W].float().cpu() - ref).abs().max() < tol * max(1.0, float(ref.abs().max()))
    assert float(xview[:, H].abs().max()) == 0.0 and float(xview[:, :, W].abs().max()) == 0.0
    # avgpool over the PP tensor == mean over the real pixels
    out = torch.zeros(N, C, dtype=dtype, device="cuda")
    check(lib.bltvqg_avgpool_pp(G.DT[dtype], ptr(xbody), ptr(out), N, H, W, C, stream_ptr()), "avgpool_pp")
    torch.cuda.synchronize()
    pm = xview[:, :H, :W].float().mean(dim=(1, 2)).cpu()
    assert (out.float().cpu() - pm).abs().max() < tol * max(1.0, float(pm.abs().max()))
    # relu(bn(x)) -> maxpool 3x3/2 pad 1 into a PP buffer
    Hi, Wi = 11, 8
    z = torch.randn(N, C, Hi, Wi, generator=g)
    zh = z.permute(0, 2, 3, 1).contiguous().to(dtype).cuda()
    Ho, Wo = (Hi - 1) // 2 + 1, (Wi - 1) // 2 + 1
    pbuf, pbody, pview = _pp_alloc(N, Ho, Wo, C, dtype)
    check(lib.bltvqg_bn_relu_maxpool_pp(G.DT[dtype], ptr(zh), ptr(scale.cuda()), ptr(shift.cuda()), ptr(pbody), N, Hi, Wi, C, stream_ptr()),
          "bn_relu_maxpool_pp")
    torch.cuda.synchronize()
    zr = torch.relu(zh.float().cpu() * scale + shift).permute(0, 3, 1, 2)
    ref = F.max_pool2d(zr, 3, 2, 1).permute(0, 2, 3, 1)
    assert (pview[:, :Ho, :Wo].float().cpu() - ref).abs().max() < tol * max(1.0, float(ref.abs().max()))
    assert float(pview[:, Ho].abs().max()) == 0.0 and float(pview[:, :, Wo].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(2688, 256, 256, True, True), (640, 256, 512, True, False), (77, 200, 96, False, True), (64, 256, 1024, False, False)])
def test_linear_with_fused_layernorm(shape):
    """Linear (+bias, +residual) with the following LayerNorm in the GEMM epilogue: C must equal the plain GEMM bit for bit on integer
    operands, the normalised output / statistics must match LayerNorm of that bf16 C."""
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    M, N, K, use_bias, use_res = shape
    g = torch.Generator().manual_seed(M + N)
    X = _ints((M, K), -2, 2, g, torch.float32)
    W = _ints((N, K), -1, 1, g, torch.float32)
    bias = _ints((N,), -3, 3, g, torch.float32) if use_bias else None
    R = _ints((M, N), -4, 4, g, torch.float32) if use_res else None
    gamma, beta = torch.rand(N, generator=g) + 0.5, torch.randn(N, generator=g)
    Xd, Wd = X.bfloat16().cuda(), W.bfloat16().cuda()
    Rd = R.bfloat16().cuda() if use_res else None
    bd = bias.cuda() if use_bias else None
    C = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    out = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
    gd, btd = gamma.cuda(), beta.cuda()
    check(G_exp().bltvqg_linear_layernorm(ptr(Xd), K, ptr(Wd), K, ptr(bd), 0, 0.0, 0, 0, None, 0, ptr(Rd), N, ptr(C), N, ptr(gd), ptr(btd), 1e-5,
                                      ptr(out), ptr(mean), ptr(rstd), M, N, K, stream_ptr()), "linear_layernorm")
    torch.cuda.synchronize()
    ref = X @ W.t()
    if use_bias:
        ref = ref + bias
    if use_res:
        ref = ref + R
    refb = ref.bfloat16()
    assert torch.equal(C.cpu(), refb)
    ln = F.layer_norm(refb.float(), (N,), gamma, beta, 1e-5)
    assert (out.float().cpu() - ln).abs().max() < 2e-2 * max(1.0, float(ln.abs().max()))
    mu = refb.float().mean(1)
    var = refb.float().var(1, unbiased=False)
    assert (mean.cpu() - mu).abs().max() < 1e-4 * max(1.0, float(mu.abs().max()))
    assert ((rstd.cpu() - (var + 1e-5).rsqrt()).abs() / (var + 1e-5).rsqrt()).max() < 1e-4


@pytest.mark.parametrize("shape", [(2688, 768, 256, False), (640, 512, 256, True), (77, 256, 200, True), (130, 96, 64, False)])
def test_layernorm_folded_into_consumer_linear(shape):
    """bltvqg_layernorm_linear: LayerNorm on the A tile in LDS.  The normalised activations / statistics must match a LayerNorm launch
    and the product must equal the plain GEMM of those (bf16-rounded) activations."""
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    lib = G.lib()
    M, N, K, use_bias = shape
    g = torch.Generator().manual_seed(M + K)
    X = (torch.randn(M, K, generator=g) * 1.5 + 0.3).bfloat16()
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).bfloat16()
    bias = torch.randn(N, generator=g) if use_bias else None
    gamma, beta = torch.rand(K, generator=g) + 0.5, torch.randn(K, generator=g)
    Xd, Wd = X.cuda(), W.cuda()
    bd = bias.cuda() if use_bias else None
    Xn = torch.zeros(M, K, dtype=torch.bfloat16, device="cuda")
    C = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.zeros(M, device="cuda"), torch.zeros(M, device="cuda")
    check(G_exp().bltvqg_layernorm_linear(ptr(Xd), K, ptr(gamma.cuda()), ptr(beta.cuda()), 1e-5, ptr(Xn), ptr(mean), ptr(rstd), ptr(Wd), K, ptr(bd),
                                      0, 0.0, 0, 0, None, 0, ptr(C), N, M, N, K, stream_ptr()), "layernorm_linear")
    torch.cuda.synchronize()
    ln = F.layer_norm(X.float(), (K,), gamma, beta, 1e-5)
    assert (Xn.float().cpu() - ln).abs().max() < 2e-2 * max(1.0, float(ln.abs().max()))
    mu, var = X.float().mean(1), X.float().var(1, unbiased=False)
    assert (mean.cpu() - mu).abs().max() < 1e-4 and ((rstd.cpu() - (var + 1e-5).rsqrt()).abs() / (var + 1e-5).rsqrt()).max() < 1e-4
    # the GEMM consumed exactly the Xn it wrote
    ref = Xn.float().cpu() @ W.float().t()
    if use_bias:
        ref = ref + bias
    assert (C.float().cpu() - ref).abs().max() < 2e-2 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("geom", [(3, 224, 224, 230, 232), (5, 32, 56, 38, 62), (2, 30, 50, 36, 56), (1, 8, 260, 14, 266), (7, 12, 16, 20, 24)])
@pytest.mark.parametrize("form", [0, 1])
def test_img_pack_stem_layout(geom, form):
    """fp32 NCHW image -> the stem's zero-bordered NHWC4 bf16 image (image at (3, 3); channel 3 and every border pixel zero): the
    16-byte-vectorised form (one wave per padded row), the per-pixel form (debug key 27 = 1; also taken when W % 4 != 0 or the row is wider
    than 256 pixels) — both exact against torch (one bf16 rounding per value).  Reference: the image tensor EncoderCNN.forward hands to
    torchvision's conv1 (models/encoder_cnn.py:33)."""
    import gpu_ops as G
    N, H, W, Hp, Wp = geom
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(N, 3, H, W, generator=g)
    G.lib().bltvqg_debug_set(27, form)
    try:
        out = G.img_pack(x.cuda(), torch.bfloat16, 4, 3, 3, Hp, Wp)
        torch.cuda.synchronize()
    finally:
        G.lib().bltvqg_debug_set(27, 0)
    ref = torch.zeros(N, Hp, Wp, 4, dtype=torch.bfloat16)
    ref[:, 3:3 + H, 3:3 + W, :3] = x.permute(0, 2, 3, 1).bfloat16()
    assert torch.equal(out.cpu(), ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("T", [8, 21, 51])
def test_attn_fwd_rows_equals_the_prefix_redecode_row(dtype, T):
    """bltvqg_attn_fwd_rows (incremental greedy decoding): query row t against the first t + 1 key / value rows of [B, T, .] projections
    — the earlier steps' rows are the cache — is BIT-IDENTICAL to row t of the full pass with causal = 2 ("future keys do not exist",
    the reference's prefix re-decode, models/iq.py:134-141), pad keys included.  T = 51 (max_decode_length 50) takes the VALU kernel
    in bf16 too; cross-attention form: q_rows = T, k_rows = 0."""
    import gpu_ops as G
    from gpu_ops import check, ptr, stream_ptr
    B, heads, d = 5, 4, 64
    Hd = heads * d
    g = torch.Generator().manual_seed(T)
    qkv = torch.randn(B * T, 3 * Hd, generator=g).to(dtype).cuda()
    ids = torch.randint(0, 4, (B, T), generator=g, dtype=torch.int32)      # zeros = pad keys
    ids[:, 0] = 1
    ids = ids.cuda()
    Q, K, V = qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:]
    scale = 1.0 / d ** 0.5
    full = G.attn_fwd(Q, K, V, ids, B, heads, T, T, d, 2, scale)
    torch.cuda.synchronize()
    out = torch.zeros_like(full)
    for t in range(T):
        check(G.lib().bltvqg_attn_fwd_rows(G.DT[dtype], Q[t:].data_ptr(), 3 * Hd, T, ptr(K), 3 * Hd, ptr(V), 3 * Hd, T, out[t:].data_ptr(), Hd, ptr(ids),
                                           B, heads, 1, t + 1, d, 0, scale, stream_ptr()), "attn_fwd_rows")
    torch.cuda.synchronize()
    assert torch.equal(out, full)
    assert float(full.float().abs().sum()) > 0
    # cross-attention of one query row per batch element over a short, separately laid out key set
    Sk = 5
    kv = torch.randn(B * Sk, 2 * Hd, generator=g).to(dtype).cuda()
    kid = torch.tensor([[1, 2, 0, 3, 0]] * B, dtype=torch.int32).cuda()
    fullc = G.attn_fwd(Q, kv[:, :Hd], kv[:, Hd:], kid, B, heads, T, Sk, d, 0, scale)
    outc = torch.zeros_like(fullc)
    t = T - 2
    check(G.lib().bltvqg_attn_fwd_rows(G.DT[dtype], Q[t:].data_ptr(), 3 * Hd, T, kv.data_ptr(), 2 * Hd, kv[:, Hd:].data_ptr(), 2 * Hd, 0, outc[t:].data_ptr(),
                                       Hd, ptr(kid), B, heads, 1, Sk, d, 0, scale, stream_ptr()), "attn_fwd_rows")
    torch.cuda.synchronize()
    assert torch.equal(outc.view(B, T, Hd)[:, t], fullc.view(B, T, Hd)[:, t])
    assert float(outc.view(B, T, Hd)[:, :t].float().abs().sum()) == 0
    # a row-subset view is forward-only and must cover the rows it names
    assert G.lib().bltvqg_attn_fwd_rows(G.DT[dtype], ptr(Q), 3 * Hd, 1, ptr(K), 3 * Hd, ptr(V), 3 * Hd, T, ptr(out), Hd, ptr(ids), B, heads, 2, T, d, 0,
                                        scale, stream_ptr()) != 0
