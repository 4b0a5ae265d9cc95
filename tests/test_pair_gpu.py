"""Two Linear problems in one planned-tile launch (gemm2.hip::gemm_nt2_pair_kernel, bltvqg_linear_pair) through the C ABI of the EXPERIMENTS
library (include/bltvqg_hip_experiments.h: built, bit-identical, measured in the train step, not adopted — DESIGN.md 9).

Reference call sites: the encoder stack and the posterior encoder stack (models/iq.py:31-34, run back to back in IQ.forward, iq.py:66-78)
execute the same Linear positions (models/transformer_layers.py:260-275: q|k|v, attention output, the two FFN layers) on different rows
with different weights.

  * a paired launch at a forced tile shape is BIT-IDENTICAL to the two separate launches at that shape, for every epilogue form the
    stacks use: bias + ReLU + dropout + second output (forward FFN layer 0), bias + residual + row statistics (attention output / FFN
    layer 1), mask + scale (ReLU/dropout backward), the folded-LayerNorm consumer;
  * the planner's own choice for the pair against float64 on the same bf16 operands;
  * mismatched epilogue terms are refused.
"""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu

TILES = [(64, 64), (128, 128), (160, 64), (160, 128), (160, 256), (192, 192), (128, 256), (256, 128), (256, 64), (224, 256), (32, 64)]
SHAPES = [((1280, 5376), 256, 256), ((64, 5376), 512, 256), ((333, 517), 200, 72), ((70, 1), 96, 136)]


def _desc(**kw):
    from bltvqg_amd._lib import LinearDesc
    d = LinearDesc()
    for k, v in kw.items():
        if isinstance(v, torch.Tensor):
            setattr(d, k, v.data_ptr())
        elif v is not None:
            setattr(d, k, v)
    return d


def _pair(d1, d2, N, K, tile, relu=0, drop_p=0.0, seed=0, mask_scale=1.0, slots=0, parts=0, eps=1e-5):
    import gpu_ops as G
    from bltvqg_amd._lib import stream_ptr
    return G.exp_lib().bltvqg_linear_pair(ctypes.byref(d1), ctypes.byref(d2), N, K, int(relu), float(drop_p), int(seed), float(mask_scale), slots, parts,
                                      float(eps), tile[0], tile[1], stream_ptr())


def _operands(M, N, K, g, scale=0.05):
    A = torch.randn(M, K, generator=g).bfloat16().cuda()
    W = (torch.randn(N, K, generator=g) * scale).bfloat16().cuda()
    b = torch.randn(N, generator=g).cuda()
    R = torch.randn(M, N, generator=g).bfloat16().cuda()
    return A, W, b, R


@pytest.mark.parametrize("tile", TILES)
def test_pair_is_bit_identical_to_two_launches_forward_forms(tile):
    import gpu_ops as G
    from bltvqg_amd._lib import check, stream_ptr
    g = torch.Generator().manual_seed(5 + tile[0] + tile[1])
    for (Ms, N, K) in SHAPES:
        ops = [_operands(M, N, K, g) for M in Ms]
        # --- bias + ReLU + dropout + second output (FFN layer 0 forward, transformer_layers.py:400-408) ---
        single, paired = [], []
        for i, (A, W, b, R) in enumerate(ops):
            C = torch.zeros(Ms[i], N, dtype=torch.bfloat16, device="cuda")
            C2 = torch.zeros_like(C)
            check(G.lib().bltvqg_gemm_ex(G.ptr(A), K, G.ptr(W), K, G.ptr(C), N, Ms[i], N, K, G.ptr(b), None, None, 0, 1, 0.1, 77, 11 + i, None, 0, 1.0,
                                         G.ptr(C2), N, None, 0, 0, tile[0], tile[1], stream_ptr()), "gemm_ex")
            single.append((C, C2))
            paired.append((torch.zeros_like(C), torch.zeros_like(C)))
        ds = [_desc(A=ops[i][0], lda=K, W=ops[i][1], ldw=K, C=paired[i][0], ldc=N, M=Ms[i], bias=ops[i][2], C2=paired[i][1], ldc2=N, stream_id=11 + i)
              for i in range(2)]
        check(_pair(ds[0], ds[1], N, K, tile, relu=1, drop_p=0.1, seed=77), "linear_pair")
        torch.cuda.synchronize()
        for i in range(2):
            assert torch.equal(single[i][0], paired[i][0]), (tile, Ms, N, K, i)
            assert torch.equal(single[i][1], paired[i][1]), (tile, Ms, N, K, i)
            assert float(single[i][0].float().abs().sum()) > 0
        # --- bias + residual + row statistics (attention output / FFN layer 1: transformer_layers.py:266-268,275) ---
        if tile[1] == 192:
            continue
        parts = int(G.lib().bltvqg_gemm_rowstat_parts(1, N, K, tile[1]))
        slots = parts + 1
        single, paired = [], []
        for i, (A, W, b, R) in enumerate(ops):
            C = torch.zeros(Ms[i], N, dtype=torch.bfloat16, device="cuda")
            st = torch.full((Ms[i], slots, 2), 3.0, device="cuda")
            check(G.lib().bltvqg_gemm_rowstat(G.ptr(A), K, G.ptr(W), K, G.ptr(C), N, Ms[i], N, K, G.ptr(b), 0, 0.0, 0, 0, None, 0, G.ptr(R), N, G.ptr(st),
                                              slots, tile[0], tile[1], stream_ptr()), "gemm_rowstat")
            single.append((C, st))
            paired.append((torch.zeros_like(C), torch.full_like(st, 3.0)))
        ds = [_desc(A=ops[i][0], lda=K, W=ops[i][1], ldw=K, C=paired[i][0], ldc=N, M=Ms[i], bias=ops[i][2], R=ops[i][3], ldr=N, out_stat=paired[i][1])
              for i in range(2)]
        check(_pair(ds[0], ds[1], N, K, tile, slots=slots), "linear_pair")
        torch.cuda.synchronize()
        for i in range(2):
            assert torch.equal(single[i][0], paired[i][0]), (tile, Ms, N, K, i)
            assert torch.equal(single[i][1], paired[i][1]), (tile, Ms, N, K, i)


@pytest.mark.parametrize("tile", [(160, 64), (128, 256), (256, 64), (64, 64)])
def test_pair_is_bit_identical_backward_mask_form(tile):
    """dX through the ReLU/dropout mask of the forward (maskY != 0) * mask_scale, with strided views (the top layer on row 0 only)."""
    import gpu_ops as G
    from bltvqg_amd._lib import check, stream_ptr
    g = torch.Generator().manual_seed(8)
    Ms, N, K, S = (640, 64), 512, 256, 5
    single, paired, ops = [], [], []
    for i, M in enumerate(Ms):
        ld = S if i == 1 else 1          # problem 2: every S-th row of wider buffers
        A = torch.randn(M * ld, K, generator=g).bfloat16().cuda()
        W = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
        Y = torch.relu(torch.randn(M * ld, N, generator=g)).bfloat16().cuda()
        ops.append((A, W, Y, ld))
        C = torch.zeros(M * ld, N, dtype=torch.bfloat16, device="cuda")
        check(G.lib().bltvqg_gemm_ex(G.ptr(A), K * ld, G.ptr(W), K, G.ptr(C), N * ld, M, N, K, None, None, None, 0, 0, 0.0, 0, 0, G.ptr(Y), N * ld, 1.25,
                                     None, 0, None, 0, 0, tile[0], tile[1], stream_ptr()), "gemm_ex")
        single.append(C)
        paired.append(torch.zeros_like(C))
    ds = [_desc(A=ops[i][0], lda=K * ops[i][3], W=ops[i][1], ldw=K, C=paired[i], ldc=N * ops[i][3], M=Ms[i], maskY=ops[i][2], ldm=N * ops[i][3])
          for i in range(2)]
    check(_pair(ds[0], ds[1], N, K, tile, mask_scale=1.25), "linear_pair")
    torch.cuda.synchronize()
    for i in range(2):
        assert torch.equal(single[i], paired[i]), (tile, i)
        assert float(single[i].float().abs().sum()) > 0
    assert float(paired[1].view(Ms[1], S, N)[:, 1:].float().abs().sum()) == 0      # rows between the strided ones stay untouched


@pytest.mark.parametrize("tile", [(160, 64), (160, 256), (128, 256), (224, 256), (32, 64)])
def test_pair_is_bit_identical_folded_layernorm_form(tile):
    import gpu_ops as G
    from bltvqg_amd._lib import check, stream_ptr
    g = torch.Generator().manual_seed(13)
    Ms, N, K, parts = (421, 1333), 328, 256, 3
    slots = parts + 1
    single, paired, ops = [], [], []
    for i, M in enumerate(Ms):
        X = (torch.randn(M, K, generator=g) * 2.0 + torch.randn(M, 1, generator=g)).bfloat16().cuda()
        Wf = (torch.randn(N, K, generator=g) * 0.05).bfloat16().cuda()
        fs = Wf.float().sum(1).contiguous()
        fc = torch.randn(N, generator=g).cuda()
        st = torch.zeros(M, slots, 2, device="cuda")
        x = X.float()
        st[:, 0, 0] = x[:, :100].sum(1); st[:, 1, 0] = x[:, 100:200].sum(1); st[:, 2, 0] = x[:, 200:].sum(1)
        st[:, 0, 1] = (x[:, :100] ** 2).sum(1); st[:, 1, 1] = (x[:, 100:200] ** 2).sum(1); st[:, 2, 1] = (x[:, 200:] ** 2).sum(1)
        ops.append((X, Wf, fs, fc, st))
        Y = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
        mean = torch.zeros(M, device="cuda"); rstd = torch.zeros(M, device="cuda")
        check(G.lib().bltvqg_linear_ln_folded(G.ptr(X), K, G.ptr(Wf), K, G.ptr(Y), N, M, N, K, G.ptr(fs), G.ptr(fc), G.ptr(st), slots, parts, G.ptr(mean),
                                              G.ptr(rstd), 1e-5, 1, 0.1, 5, 40 + i, tile[0], tile[1], stream_ptr()), "linear_ln_folded")
        single.append((Y, mean, rstd))
        paired.append((torch.zeros_like(Y), torch.zeros_like(mean), torch.zeros_like(rstd)))
    ds = [_desc(A=ops[i][0], lda=K, W=ops[i][1], ldw=K, C=paired[i][0], ldc=N, M=Ms[i], fold_s=ops[i][2], fold_c=ops[i][3], row_stat=ops[i][4],
                mean=paired[i][1], rstd=paired[i][2], stream_id=40 + i) for i in range(2)]
    check(_pair(ds[0], ds[1], N, K, tile, relu=1, drop_p=0.1, seed=5, slots=slots, parts=parts), "linear_pair")
    torch.cuda.synchronize()
    for i in range(2):
        for a, b in zip(single[i], paired[i]):
            assert torch.equal(a, b), (tile, i)
        # and the statistics are the LayerNorm's
        x = ops[i][0].double()
        assert float((paired[i][1].double() - x.mean(1)).abs().max()) < 1e-4
        assert float((paired[i][2].double() * torch.sqrt(x.var(1, unbiased=False) + 1e-5) - 1).abs().max()) < 1e-4


def test_pair_planned_tile_against_float64():
    g = torch.Generator().manual_seed(2)
    for (Ms, N, K) in (((1280, 5376), 768, 256), ((1280, 5376), 256, 512), ((64, 5376), 256, 256)):
        ops = [_operands(M, N, K, g) for M in Ms]
        out = [torch.zeros(M, N, dtype=torch.bfloat16, device="cuda") for M in Ms]
        ds = [_desc(A=ops[i][0], lda=K, W=ops[i][1], ldw=K, C=out[i], ldc=N, M=Ms[i], bias=ops[i][2], R=ops[i][3], ldr=N) for i in range(2)]
        from bltvqg_amd._lib import check
        check(_pair(ds[0], ds[1], N, K, (0, 0)), "linear_pair")
        torch.cuda.synchronize()
        for i, (A, W, b, R) in enumerate(ops):
            ref = A.double() @ W.double().t() + b.double() + R.double()
            err = float((out[i].double() - ref).norm() / ref.norm())
            assert err < 4e-3, (Ms, N, K, i, err)


def test_pair_refuses_mismatched_epilogue_terms():
    import gpu_ops as G
    g = torch.Generator().manual_seed(1)
    M, N, K = 300, 128, 64
    A, W, b, R = _operands(M, N, K, g)
    C1 = torch.zeros(M, N, dtype=torch.bfloat16, device="cuda")
    C2 = torch.zeros_like(C1)
    d1 = _desc(A=A, lda=K, W=W, ldw=K, C=C1, ldc=N, M=M, bias=b)
    d2 = _desc(A=A, lda=K, W=W, ldw=K, C=C2, ldc=N, M=M)                # no bias
    assert _pair(d1, d2, N, K, (0, 0)) != 0
    assert b"same epilogue" in G.exp_lib().bltvqg_last_error_string()
    d3 = _desc(A=A, lda=K + 4, W=W, ldw=K, C=C2, ldc=N, M=M, bias=b)     # pitch not a multiple of 8
    assert _pair(d1, d3, N, K, (0, 0)) != 0
    d4 = _desc(A=A, lda=K, W=W, ldw=K, C=C2, ldc=N, M=0, bias=b)
    assert _pair(d1, d4, N, K, (0, 0)) != 0
    torch.cuda.synchronize()
    assert float(C2.float().abs().sum()) == 0
