"""LayerNorm folded into the Linear that consumes it (gemm2.hip FOLD variant, misc.hip::ln_fold_prepare, GemmArgs::out_stat) through
the C ABI.

Reference call sites: every pre-LayerNorm of the stacks and the one Linear behind it — models/transformer_layers.py:260-262 (q|k|v),
:271-273 / :356-358 (FFN layer 0, :400-408 with its ReLU + dropout), :340-342 (cross-attention query).

  * row statistics (out_stat): EXACT on integer operands — sums and sums of squares of the stored bf16 rows, every compiled tile, tails;
  * fold preparation: W' = bf16(W gamma), s = sum of the ROUNDED W', c = sum beta W + bias against float64;
  * the folded Linear against (a) the algebraic identity in float64 on the same rounded operands (tight) and (b) torch LayerNorm + Linear in
    fp32 (the bf16 tolerance the engine states for its activations).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

TILES = [(0, 0), (64, 64), (64, 128), (128, 64), (128, 128), (160, 64), (160, 128), (160, 192), (160, 256), (192, 64), (192, 128), (192, 192),
         (192, 256), (128, 256), (256, 128), (256, 64), (224, 256), (96, 64), (32, 64)]


def _ints(shape, lo, hi, g):
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def _rowstat(A, W, M, N, K, tile, bias=None, relu=False, R=None, C2=None):
    import gpu_ops as G
    from bltvqg_amd._lib import check, stream_ptr
    C = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    parts = int(G.lib().bltvqg_gemm_rowstat_parts(M, N, K, tile[1]))
    slots = parts + 2
    st = torch.full((M, slots, 2), 7.0, dtype=torch.float32, device="cuda")      # slots beyond `parts` must stay untouched
    check(G.lib().bltvqg_gemm_rowstat(G.ptr(A), A.stride(0), G.ptr(W), W.stride(0), G.ptr(C), C.stride(0), M, N, K, G.ptr(bias), int(relu), 0.0, 0, 0,
                                      G.ptr(C2), 0 if C2 is None else C2.stride(0), G.ptr(R), 0 if R is None else R.stride(0), G.ptr(st), slots, tile[0],
                                      tile[1], stream_ptr()), "gemm_rowstat")
    torch.cuda.synchronize()
    assert float((st[:, parts:, :] - 7.0).abs().max()) == 0.0
    return C, st[:, :parts, :].double().sum(1).float()


@pytest.mark.parametrize("tile", TILES)
def test_row_statistics_of_the_stored_rows_are_exact(tile):
    g = torch.Generator().manual_seed(11 + tile[0] + tile[1])
    for (M, N, K) in ((333, 200, 72), (70, 512, 64), (517, 96, 136)):
        A = _ints((M, K), -2, 2, g).bfloat16().cuda()
        W = _ints((N, K), -1, 1, g).bfloat16().cuda()
        R = _ints((M, N), -3, 3, g).bfloat16().cuda()
        bias = _ints((N,), -2, 2, g).cuda()
        C, st = _rowstat(A, W, M, N, K, tile, bias=bias, R=R)
        torch.cuda.synchronize()
        ref = A.double().cpu() @ W.double().cpu().t() + bias.double().cpu() + R.double().cpu()
        assert float(ref.abs().max()) < 256          # integers this small are exact in bf16
        assert torch.equal(C.double().cpu(), ref), (tile, M, N, K)
        assert torch.equal(st[:, 0].double().cpu(), ref.sum(1)), (tile, M, N, K)
        assert torch.equal(st[:, 1].double().cpu(), (ref * ref).sum(1)), (tile, M, N, K)


def test_row_statistics_are_those_of_the_rounded_output():
    """Non-integer data: the statistics must describe the row AS STORED (bf16), which is what the consumer's MFMA multiplies."""
    g = torch.Generator().manual_seed(3)
    M, N, K = 640, 512, 512
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    R = torch.randn(M, N, generator=g).bfloat16().cuda()
    C, st = _rowstat(A, W, M, N, K, (0, 0), R=R)
    torch.cuda.synchronize()
    x = C.double().cpu()
    assert float((st[:, 0].double().cpu() - x.sum(1)).abs().max()) < 1e-3
    assert float(((st[:, 1].double().cpu() - (x * x).sum(1)) / (x * x).sum(1)).abs().max()) < 1e-5


def test_row_statistics_do_not_depend_on_the_tile_shape():
    """One slot per 64 columns whatever the tile: every compiled tile leaves the SAME slot array, bit for bit (what makes an incremental
    pass over B rows reproduce the full pass over B * T rows, engine.hip::dec_step_fwd)."""
    g = torch.Generator().manual_seed(6)
    M, N, K = 333, 328, 136
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.1).bfloat16().cuda()
    R = torch.randn(M, N, generator=g).bfloat16().cuda()
    import gpu_ops as G
    from bltvqg_amd._lib import check, stream_ptr
    ref = None
    for tile in TILES:
        C = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        st = torch.zeros(M, 6, 2, dtype=torch.float32, device="cuda")
        assert int(G.lib().bltvqg_gemm_rowstat_parts(M, N, K, tile[1])) == 6
        check(G.lib().bltvqg_gemm_rowstat(G.ptr(A), K, G.ptr(W), K, G.ptr(C), N, M, N, K, None, 0, 0.0, 0, 0, None, 0, G.ptr(R), N, G.ptr(st), 6, tile[0],
                                          tile[1], stream_ptr()), "gemm_rowstat")
        torch.cuda.synchronize()
        if ref is None:
            ref = (C.clone(), st.clone())
            x = C.float()
            for gi in range(6):      # slot g = columns 64 g .. 64 g + 63 of the stored row
                cols = x[:, 64 * gi:64 * gi + 64].double()
                assert float((st[:, gi, 0].double() - cols.sum(1)).abs().max()) < 1e-3
                assert float((st[:, gi, 1].double() - (cols * cols).sum(1)).abs().max()) < 1e-2
        assert torch.equal(C, ref[0]), tile
        assert torch.equal(st, ref[1]), tile


def _prepare(W, gamma, beta, bias):
    import gpu_ops as G
    from bltvqg_amd._lib import check, stream_ptr
    N, K = W.shape
    Wf = torch.zeros(N, K, dtype=torch.bfloat16, device="cuda")
    fs = torch.zeros(N, dtype=torch.float32, device="cuda")
    fc = torch.zeros(N, dtype=torch.float32, device="cuda")
    check(G.lib().bltvqg_ln_fold_prepare(G.ptr(W), N, K, G.ptr(gamma), G.ptr(beta), G.ptr(bias), G.ptr(Wf), G.ptr(fs), G.ptr(fc), stream_ptr()),
          "ln_fold_prepare")
    return Wf, fs, fc


@pytest.mark.parametrize("with_bias", [False, True])
def test_fold_preparation(with_bias):
    g = torch.Generator().manual_seed(9)
    N, K = 203, 512
    W = (torch.randn(N, K, generator=g) * 0.1).cuda()
    gamma = (1.0 + 0.2 * torch.randn(K, generator=g)).cuda()
    beta = (0.1 * torch.randn(K, generator=g)).cuda()
    bias = torch.randn(N, generator=g).cuda() if with_bias else None
    Wf, fs, fc = _prepare(W, gamma, beta, bias)
    torch.cuda.synchronize()
    ref_wf = (W * gamma).bfloat16()
    assert torch.equal(Wf, ref_wf)
    assert float((fs.double() - ref_wf.double().sum(1)).abs().max()) < 1e-4
    ref_c = (W.double() * beta.double()).sum(1) + (bias.double() if with_bias else 0.0)
    assert float((fc.double() - ref_c).abs().max()) < 1e-4


def _folded(X, Wf, fs, fc, st, tile, relu=False, drop_p=0.0, seed=0, sid=0, eps=1e-5, parts=3):
    import gpu_ops as G
    from bltvqg_amd._lib import check, stream_ptr
    M, K = X.shape
    N = Wf.shape[0]
    Y = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    mean = torch.zeros(M, dtype=torch.float32, device="cuda")
    rstd = torch.zeros(M, dtype=torch.float32, device="cuda")
    # the row sums as `parts` partial sums in the first slots of a wider slot array (as a producer GEMM with that many column tiles leaves them)
    slots = parts + 1
    w = torch.rand(M, parts, 1, device="cuda", dtype=torch.float64) + 0.1
    sp = torch.full((M, slots, 2), 1e9, dtype=torch.float32, device="cuda")
    sp[:, :parts, :] = (st.double()[:, None, :] * w / w.sum(1, keepdim=True)).float()
    check(G.lib().bltvqg_linear_ln_folded(G.ptr(X), X.stride(0), G.ptr(Wf), Wf.stride(0), G.ptr(Y), Y.stride(0), M, N, K, G.ptr(fs), G.ptr(fc), G.ptr(sp),
                                          slots, parts, G.ptr(mean), G.ptr(rstd), eps, int(relu), float(drop_p), int(seed), int(sid), tile[0], tile[1],
                                          stream_ptr()), "linear_ln_folded")
    return Y, mean, rstd


def test_row_statistics_are_bit_reproducible():
    g = torch.Generator().manual_seed(4)
    M, N, K = 5120, 512, 512
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
    R = torch.randn(M, N, generator=g).bfloat16().cuda()
    ref = None
    for _ in range(20):
        _, st = _rowstat(A, W, M, N, K, (0, 0), R=R)
        if ref is None:
            ref = st.clone()
        assert torch.equal(st, ref)


@pytest.mark.parametrize("tile", [(0, 0), (160, 64), (160, 256), (128, 256), (192, 192), (64, 64), (32, 64)])
@pytest.mark.parametrize("relu", [False, True])
def test_folded_linear_matches_layernorm_then_linear(tile, relu):
    g = torch.Generator().manual_seed(21 + tile[0])
    M, N, K = 421, 328, 512
    # a residual-stream-like input: per-row offset and scale, so that mean * s_n is NOT small against the result
    X = ((torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g) * 3.0)) + torch.randn(M, 1, generator=g) * 2.0).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.06).cuda()
    gamma = (1.0 + 0.2 * torch.randn(K, generator=g)).cuda()
    beta = (0.1 * torch.randn(K, generator=g)).cuda()
    bias = (0.5 * torch.randn(N, generator=g)).cuda()
    Wf, fs, fc = _prepare(W, gamma, beta, bias)
    xd = X.double()
    st = torch.stack([xd.sum(1), (xd * xd).sum(1)], 1).float().contiguous()
    Y, mean, rstd = _folded(X, Wf, fs, fc, st, tile, relu=relu)
    torch.cuda.synchronize()
    # (a) the identity on the operands the kernel sees, float64
    mu = st[:, 0].double() / K
    rs = 1.0 / torch.sqrt(torch.clamp(st[:, 1].double() / K - mu * mu, min=0.0) + 1e-5)
    ref = rs[:, None] * (xd @ Wf.double().t() - mu[:, None] * fs.double()[None, :]) + fc.double()[None, :]
    if relu:
        ref = torch.relu(ref)
    err = float((Y.double() - ref).abs().max() / ref.abs().max())
    assert err < 6e-3, err                      # one bf16 rounding of the result
    assert float((mean.double() - mu).abs().max()) < 2e-5 and float(((rstd.double() - rs) / rs).abs().max()) < 2e-4
    # (b) against torch: LayerNorm (fp32) -> Linear
    xn = torch.nn.functional.layer_norm(X.float(), (K,), gamma, beta, 1e-5)
    ref2 = xn @ W.t() + bias
    if relu:
        ref2 = torch.relu(ref2)
    rel = float((Y.float() - ref2).norm() / ref2.norm())
    assert rel < 8e-3, rel
    assert float((mean - X.float().mean(1)).abs().max()) < 1e-4


def test_folded_linear_dropout_uses_the_exported_mask():
    import gpu_ops as G
    from bltvqg_amd._lib import check, stream_ptr
    g = torch.Generator().manual_seed(2)
    M, N, K = 300, 256, 256
    X = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * 0.06).cuda()
    gamma, beta = torch.ones(K).cuda(), torch.zeros(K).cuda()
    Wf, fs, fc = _prepare(W, gamma, beta, None)
    xd = X.double()
    st = torch.stack([xd.sum(1), (xd * xd).sum(1)], 1).float().contiguous()
    Y0, _, _ = _folded(X, Wf, fs, fc, st, (0, 0), relu=True, parts=1)
    Y1, _, _ = _folded(X, Wf, fs, fc, st, (0, 0), relu=True, drop_p=0.25, seed=77, sid=5, parts=1)
    mask = torch.zeros(M, N, dtype=torch.uint8, device="cuda")
    check(G.lib().bltvqg_dropout_mask(77, 5, M, N, (N + 7) // 8 * 8, 0.25, G.ptr(mask), stream_ptr()), "dropout_mask")
    torch.cuda.synchronize()
    ref = torch.where(mask.bool(), Y0.float() / 0.75, torch.zeros_like(Y0.float()))
    assert float((Y1.float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max())
    assert 0.70 < float(mask.float().mean()) < 0.80
