"""The round-2 GEMM kernels (csrc/gemm2.hip) through the C ABI, bit-exact on integer-valued operands (products and sums of small
integers are exact in bf16 x bf16 -> fp32, so any indexing / swizzle / tail / epilogue-order mistake shows as a wrong integer):
  * bltvqg_gemm_ex: the planned-tile Linear forward / input-gradient kernel, every compiled tile shape, M / N / K tails, and every
    epilogue term (bias, position table, ReLU, dropout with the exported Philox mask, ReLU/dropout-backward mask, second output,
    residual, accumulate) against a float64 torch reference AND against the round-1 kernel;
  * bltvqg_linear_wgrad_group: many weight gradients in one launch (token-major operands, transposed LDS reads, bias gradients from the
    ones-MFMA), stored and split-K forms, ragged sizes.
Reference call sites: models/transformer_layers.py:400-408,453-456,489-491,530 (Linear forward) and their autograd backward."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

TILES = [(64, 64), (64, 128), (128, 64), (128, 128), (160, 64), (160, 128), (160, 192), (160, 256), (192, 64), (192, 128), (192, 192), (192, 256),
         (128, 256), (256, 128), (256, 64), (224, 256), (96, 64), (32, 64)]


def _ints(shape, lo, hi, g):
    return torch.randint(lo, hi + 1, shape, generator=g).float()


def _gemm_ex(A, W, M, N, K, tile, bias=None, rowtab=None, rowidx=None, relu=False, drop_p=0.0, seed=0, stream_id=0, maskY=None, mask_scale=1.0, C2=None,
             R=None, C=None, accumulate=False):
    import gpu_ops as G
    from bltvqg_amd._lib import check, stream_ptr
    if C is None:
        C = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    check(G.lib().bltvqg_gemm_ex(G.ptr(A), A.stride(0), G.ptr(W), W.stride(0), G.ptr(C), C.stride(0), M, N, K, G.ptr(bias), G.ptr(rowtab), G.ptr(rowidx),
                                 0 if rowtab is None else rowtab.stride(0), int(relu), float(drop_p), int(seed), int(stream_id), G.ptr(maskY),
                                 0 if maskY is None else maskY.stride(0), float(mask_scale), G.ptr(C2), 0 if C2 is None else C2.stride(0), G.ptr(R),
                                 0 if R is None else R.stride(0), int(accumulate), tile[0], tile[1], stream_ptr()), "gemm_ex")
    return C


@pytest.mark.parametrize("tile", TILES)
def test_nt2_every_tile_exact_with_tails(tile):
    g = torch.Generator().manual_seed(tile[0] * 7 + tile[1])
    # M, N not multiples of the tile (nor of 16), K with a partial last K-step and a partial 8-chunk
    for (M, N, K) in ((max(2 * tile[0] + 37, 300), tile[1] + 24, 200), (300, 72, 64), (517, 2 * tile[1] + 8, 328)):
        A = _ints((M, K), -2, 2, g).bfloat16().cuda()
        W = _ints((N, K), -2, 2, g).bfloat16().cuda()
        ref = (A.double().cpu() @ W.double().cpu().t())
        assert float(ref.abs().max()) < 2 ** 15
        got = _gemm_ex(A, W, M, N, K, tile)
        torch.cuda.synchronize()
        assert torch.equal(got.float().cpu().double(), ref.float().bfloat16().double()), (tile, M, N, K, float((got.float().cpu().double() - ref).abs().max()))


@pytest.mark.parametrize("tile", [(0, 0), (160, 64), (160, 256), (192, 192), (128, 256), (64, 64)])
def test_nt2_epilogue_terms_exact_and_equal_to_round1(tile):
    import gpu_ops as G
    g = torch.Generator().manual_seed(5)
    M, N, K = 421, 200, 136
    A = _ints((M, K), -2, 2, g).bfloat16().cuda()
    W = _ints((N, K), -1, 1, g).bfloat16().cuda()
    bias = _ints((N,), -3, 3, g).cuda()
    tab = _ints((7, N), -2, 2, g).cuda()
    ridx = torch.randint(0, 7, (M,), generator=g).int().cuda()
    R = _ints((M, N), -4, 4, g).bfloat16().cuda()
    mask = (_ints((M, N), 0, 2, g)).bfloat16().cuda()            # zeros and non-zeros
    old = _ints((M, N), -3, 3, g).bfloat16().cuda()
    base = A.double().cpu() @ W.double().cpu().t()
    p = 0.25
    keep = G.dropout_mask(77, 9, M, N, (N + 7) // 8 * 8, p).cpu().double()
    cases = {
        "bias+tab": (dict(bias=bias, rowtab=tab, rowidx=ridx), base + bias.double().cpu() + tab.double().cpu()[ridx.long().cpu()]),
        "relu+res": (dict(bias=bias, relu=True, R=R), torch.relu(base + bias.double().cpu()) + R.double().cpu()),
        "mask": (dict(maskY=mask, mask_scale=2.0), torch.where(mask.double().cpu() != 0, base * 2.0, torch.zeros_like(base))),
        "mask+res": (dict(maskY=mask, mask_scale=0.5, R=R), torch.where(mask.double().cpu() != 0, base * 0.5, torch.zeros_like(base)) + R.double().cpu()),
        "dropout": (dict(relu=True, drop_p=p, seed=77, stream_id=9), torch.relu(base) * keep / (1 - p)),
    }
    for name, (kw, ref) in cases.items():
        got = _gemm_ex(A, W, M, N, K, tile, **kw)
        r1 = _gemm_ex(A, W, M, N, K, (-1, -1), **kw)
        torch.cuda.synchronize()
        assert torch.equal(got, r1), (name, tile)
        want = ref.float().bfloat16().double()
        if name == "dropout":
            assert torch.allclose(got.float().cpu().double(), want, rtol=1e-2), name      # 1/(1-p) is not a bf16 number
        else:
            assert torch.equal(got.float().cpu().double(), want), (name, tile, float((got.float().cpu().double() - want).abs().max()))
    # second output (the pre-residual value, what the FFN keeps as its ReLU/dropout mask) and accumulate
    C2 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    C = old.clone()
    got = _gemm_ex(A, W, M, N, K, tile, bias=bias, relu=True, C2=C2, R=R, C=C, accumulate=True)
    torch.cuda.synchronize()
    pre = torch.relu(base + bias.double().cpu())
    assert torch.equal(C2.float().cpu().double(), pre.float().bfloat16().double())
    assert torch.equal(got.float().cpu().double(), (pre + R.double().cpu() + old.double().cpu()).float().bfloat16().double())


def _wgrad_group(problems, table_bytes=1 << 16):
    import gpu_ops as G
    from bltvqg_amd._lib import check, stream_ptr
    n = len(problems)
    PA, IA = ctypes.c_void_p * n, ctypes.c_int32 * n
    outs = []
    dY, X, dW, db, ldy, ldx, ldw, rows, Ns, Ks = [], [], [], [], [], [], [], [], [], []
    for (y, x, with_bias, N, K) in problems:
        w = torch.zeros(N, K, dtype=torch.float32, device="cuda")
        b = torch.zeros(N, dtype=torch.float32, device="cuda") if with_bias else None
        outs.append((w, b))
        dY.append(y.data_ptr()); X.append(x.data_ptr()); dW.append(w.data_ptr()); db.append(b.data_ptr() if with_bias else None)
        ldy.append(y.stride(0)); ldx.append(x.stride(0)); ldw.append(K); rows.append(y.shape[0]); Ns.append(N); Ks.append(K)
    table = torch.zeros(table_bytes, dtype=torch.uint8, device="cuda")
    check(G.lib().bltvqg_linear_wgrad_group(n, PA(*dY), IA(*ldy), PA(*X), IA(*ldx), PA(*dW), IA(*ldw), PA(*db), IA(*rows), IA(*Ns), IA(*Ks), G.ptr(table),
                                            table_bytes, stream_ptr()), "wgrad_group")
    torch.cuda.synchronize()
    return outs


@pytest.mark.parametrize("tile_rows", [128, 256, 512, 1024])      # 512 = the 256 x 256 (square) tile form
def test_wgrad_group_exact_stored_and_split(tile_rows):
    import gpu_ops as G
    G.lib().bltvqg_debug_set(13, tile_rows)          # force the 128- / 256-row tile variant (the planner picks by launch size)
    try:
        _wgrad_group_cases()
    finally:
        G.lib().bltvqg_debug_set(13, 0)


def _wgrad_group_cases():
    g = torch.Generator().manual_seed(11)
    # a "stack" of problems: enough tiles for the stored form (no split-K): ragged rows, widths that are not tile multiples, a padded ld
    big = []
    for (rows, N, K, ldx_pad, with_bias) in ((1000, 512, 256, 0, True), (777, 384, 512, 0, False), (1000, 1536, 512, 0, False), (640, 200, 304, 16, True),
                                             (2100, 512, 2048, 0, True), (333, 128, 128, 0, True)):
        y = _ints((rows, N), -2, 2, g).bfloat16().cuda()
        xfull = torch.zeros(rows, K + ldx_pad, dtype=torch.bfloat16, device="cuda")
        xfull[:, :K] = _ints((rows, K), -2, 2, g).bfloat16().cuda()
        big.append((y, xfull[:, :K] if ldx_pad == 0 else xfull, with_bias, N, K))
    outs = _wgrad_group([(y, x, wb, N, K) for (y, x, wb, N, K) in big])
    for (y, x, wb, N, K), (w, b) in zip(big, outs):
        ref = y.double().cpu().t() @ x[:, :K].double().cpu()
        assert float(ref.abs().max()) < 2 ** 23
        assert torch.equal(w.cpu().double(), ref), (N, K, float((w.cpu().double() - ref).abs().max()))
        if wb:
            assert torch.equal(b.cpu().double(), y.double().cpu().sum(0))
    # a launch that is short of tiles slices the contraction and adds atomically (integers: still exact)
    y = _ints((4096, 128), -1, 1, g).bfloat16().cuda()
    x = _ints((4096, 256), -1, 1, g).bfloat16().cuda()
    (w, b), = _wgrad_group([(y, x, True, 128, 256)])
    assert torch.equal(w.cpu().double(), y.double().cpu().t() @ x.double().cpu())
    assert torch.equal(b.cpu().double(), y.double().cpu().sum(0))


@pytest.mark.parametrize("tile_rows", [128, 256, 512, 1024])      # 512 = the 256 x 256 (square) tile form
def test_wgrad_group_strided_views_like_the_engine(tile_rows):
    """The engine's fused q|k|v gradient [rows, 3H] against xn [rows, H], and the embedding's [rows, 320]-pitched operand with K = 300."""
    import gpu_ops as G
    G.lib().bltvqg_debug_set(13, tile_rows)
    try:
        _wgrad_strided_cases()
    finally:
        G.lib().bltvqg_debug_set(13, 0)


def _wgrad_strided_cases():
    g = torch.Generator().manual_seed(3)
    rows, H = 1260, 256
    gqkv = _ints((rows, 3 * H), -2, 2, g).bfloat16().cuda()
    xn = _ints((rows, H), -2, 2, g).bfloat16().cuda()
    emb = torch.zeros(rows, 320, dtype=torch.bfloat16, device="cuda")
    emb[:, :300] = _ints((rows, 300), -2, 2, g).bfloat16().cuda()
    dx = _ints((rows, H), -2, 2, g).bfloat16().cuda()
    outs = _wgrad_group([(gqkv, xn, False, 3 * H, H), (dx, emb, True, H, 300), (gqkv[:, H:], xn, False, 2 * H, H)])
    assert torch.equal(outs[0][0].cpu().double(), gqkv.double().cpu().t() @ xn.double().cpu())
    assert torch.equal(outs[1][0].cpu().double(), dx.double().cpu().t() @ emb[:, :300].double().cpu())
    assert torch.equal(outs[1][1].cpu().double(), dx.double().cpu().sum(0))
    assert torch.equal(outs[2][0].cpu().double(), gqkv[:, H:].double().cpu().t() @ xn.double().cpu())


def test_gemm_rotate_computes_every_copy_and_the_chain():
    """bltvqg_gemm_rotate (timing aid with cold operands): launch i uses copy i % copies of A / B / C; chain = 1 feeds launch i-1's output to
    launch i.  Every copy must hold the plain GEMM's result (integer operands: exact)."""
    import gpu_ops as G
    from bltvqg_amd._lib import ptr, stream_ptr, check
    lib = G.exp_lib()      # (experiments build: include/bltvqg_hip_experiments.h)
    M, N, K = 512, 128, 128
    g = torch.Generator().manual_seed(3)
    A = torch.randint(-3, 4, (3, M, K), generator=g).float().bfloat16().cuda()
    B = torch.randint(-2, 3, (2, N, K), generator=g).float().bfloat16().cuda()
    C = torch.zeros(6, M, N, dtype=torch.bfloat16, device="cuda")
    check(lib.bltvqg_gemm_rotate(1, ptr(A), K, 3, M * K * 2, ptr(B), K, 2, N * K * 2, ptr(C), N, 6, M * N * 2, M, N, K, 0, 6, stream_ptr()), "rotate")
    torch.cuda.synchronize()
    for i in range(6):
        want = (A[i % 3].float() @ B[i % 2].float().t())
        assert torch.equal(C[i].float(), want.bfloat16().float()), i
    # chain: N == K; launch 0 reads A[0], launch i reads C[(i - 1) % 2]
    Bs = torch.eye(K).bfloat16().cuda().unsqueeze(0).contiguous()             # identity weights: every link of the chain reproduces A[0]
    C2 = torch.zeros(2, M, K, dtype=torch.bfloat16, device="cuda")
    check(lib.bltvqg_gemm_rotate(1, ptr(A), K, 1, 0, ptr(Bs), K, 1, 0, ptr(C2), K, 2, M * K * 2, M, K, K, 1, 5, stream_ptr()), "chain")
    torch.cuda.synchronize()
    assert torch.equal(C2[0], A[0]) and torch.equal(C2[1], A[0])
