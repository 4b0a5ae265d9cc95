"""Test-side convenience wrappers: call the C ABI of libbltvqg_hip.so with torch CUDA tensors."""
import ctypes

import torch

from bltvqg_amd import _lib
from bltvqg_amd._lib import check, stream_ptr

_KEEP = []


def ptr(t):
    """Device pointer of a tensor; keeps the tensor alive so that a temporary such as ``ptr(x.cuda())`` is not returned to
    the caching allocator (and overwritten by the next temporary) before the asynchronous kernel has consumed it."""
    if t is None:
        return None
    _KEEP.append(t)
    if len(_KEEP) > 512:
        del _KEEP[:256]
    return _lib.ptr(t)

DT = {torch.float32: 0, torch.bfloat16: 1}
TD = {0: torch.float32, 1: torch.bfloat16}


def lib():
    return _lib.load()


def exp_lib():
    """The experiments build (include/bltvqg_hip_experiments.h); the calling test is skipped when it has not been built."""
    import pytest
    e = _lib.load_experiments()
    if e is None:
        pytest.skip("experiments build absent (make -C blt-vqg_amd/csrc experiments)")
    return e


def gemm(A, B, M, N, K, transA=False, transB=False, bias=None, relu=False, drop_p=0.0, seed=0, stream_id=0, maskY=None,
         mask_scale=1.0, R=None, C=None, accumulate=False, out_f32=False, force_tile=0, ldc=None, split_k=0):
    dt = DT[A.dtype]
    out_dtype = torch.float32 if (out_f32 or dt == 0) else torch.bfloat16
    if C is None:
        ldc = ldc or N
        C = torch.zeros(M, ldc, dtype=out_dtype, device=A.device)
    else:
        ldc = C.stride(0)
    check(lib().bltvqg_gemm(dt, ptr(A), A.stride(0), int(transA), ptr(B), B.stride(0), int(transB), ptr(C), ldc, M, N, K, ptr(bias),
                            int(relu), float(drop_p), int(seed), int(stream_id), ptr(maskY), 0 if maskY is None else maskY.stride(0),
                            float(mask_scale), ptr(R), 0 if R is None else R.stride(0), int(accumulate), int(out_f32), int(force_tile),
                            int(split_k), stream_ptr()), "gemm")
    return C


def conv2d(x_nhwc, w_packed, N, Hi, Wi, Cin, Cout, K, stride, pad, stats=False):
    dt = DT[x_nhwc.dtype]
    Ho = (Hi + 2 * pad - K) // stride + 1
    Wo = (Wi + 2 * pad - K) // stride + 1
    y = torch.zeros(N, Ho, Wo, Cout, dtype=x_nhwc.dtype, device=x_nhwc.device)
    ssum = ssq = None
    if stats:
        rows = lib().bltvqg_conv2d_stat_rows(N, Hi, Wi, Cout, K, K, stride, pad)
        ssum = torch.zeros(rows, Cout, dtype=torch.float32, device=x_nhwc.device)
        ssq = torch.zeros(rows, Cout, dtype=torch.float32, device=x_nhwc.device)
    check(lib().bltvqg_conv2d(dt, ptr(x_nhwc), ptr(w_packed), ptr(y), N, Hi, Wi, Cin, Cout, K, K, stride, pad, ptr(ssum), ptr(ssq),
                              stream_ptr()), "conv2d")
    return y, ssum, ssq


def img_pack(images, dtype, cpad=8, pad_top=0, pad_left=0, Hp=None, Wp=None):
    N, C, H, W = images.shape
    Hp, Wp = Hp or H, Wp or W
    out = torch.empty(N, Hp, Wp, cpad, dtype=dtype, device=images.device)
    check(lib().bltvqg_img_pack(DT[dtype], ptr(images), ptr(out), N, C, H, W, cpad, pad_top, pad_left, Hp, Wp, stream_ptr()), "img_pack")
    return out


def conv_stem(x_padded, w_packed, N, H, W, Cout, stats=False):
    dt = DT[x_padded.dtype]
    Hp, Wp = x_padded.shape[1], x_padded.shape[2]
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.zeros(N, Ho, Wo, Cout, dtype=x_padded.dtype, device=x_padded.device)
    ssum = ssq = None
    if stats:
        rows = lib().bltvqg_conv_stem_stat_rows(N, H, W, Cout)
        ssum = torch.zeros(rows, Cout, dtype=torch.float32, device=y.device)
        ssq = torch.zeros(rows, Cout, dtype=torch.float32, device=y.device)
    check(lib().bltvqg_conv_stem(dt, ptr(x_padded), ptr(w_packed), ptr(y), N, H, W, Hp, Wp, Cout, ptr(ssum), ptr(ssq), stream_ptr()), "conv_stem")
    return y, ssum, ssq


def conv_pack_w(w, dtype, cpad, kwpad=None):
    Cout, Cin, KH, KW = w.shape
    kwpad = kwpad or KW
    out = torch.empty(Cout, KH, kwpad, cpad, dtype=dtype, device=w.device)
    check(lib().bltvqg_conv_pack_w(DT[dtype], ptr(w), ptr(out), Cout, Cin, KH, KW, cpad, kwpad, stream_ptr()), "conv_pack_w")
    return out


def layernorm_fwd(x, g, b, eps=1e-5):
    rows, cols = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().bltvqg_layernorm_fwd(DT[x.dtype], ptr(x), ptr(g), ptr(b), ptr(y), ptr(mean), ptr(rstd), rows, cols, eps, stream_ptr()), "ln_fwd")
    return y, mean, rstd


def layernorm_bwd(dy, x, g, mean, rstd, dres=None):
    rows, cols = x.shape
    dx = torch.empty_like(x)
    dg = torch.zeros(cols, dtype=torch.float32, device=x.device)
    db = torch.zeros(cols, dtype=torch.float32, device=x.device)
    check(lib().bltvqg_layernorm_bwd(DT[x.dtype], ptr(dy), ptr(x), ptr(g), ptr(mean), ptr(rstd), ptr(dres), ptr(dx), ptr(dg), ptr(db), rows,
                                     cols, stream_ptr()), "ln_bwd")
    return dx, dg, db


def attn_fwd(Q, K, V, key_ids, B, heads, Tq, Tk, d, causal, scale, drop_p=0.0, seed=0, stream_id=0):
    O = torch.zeros(B * Tq, heads * d, dtype=Q.dtype, device=Q.device)
    check(lib().bltvqg_attn_fwd(DT[Q.dtype], ptr(Q), Q.stride(0), ptr(K), K.stride(0), ptr(V), V.stride(0), ptr(O), O.stride(0), ptr(key_ids),
                                B, heads, Tq, Tk, d, int(causal), float(scale), float(drop_p), int(seed), int(stream_id), stream_ptr()), "attn_fwd")
    return O


def attn_out_fwd(Q, K, V, Wo, R, key_ids, B, heads, Tq, Tk, d, causal, scale, drop_p=0.0, seed=0, stream_id=0):
    """fused attention + output Linear + residual: returns (O, Y)"""
    O = torch.zeros(B * Tq, heads * d, dtype=Q.dtype, device=Q.device)
    Y = torch.zeros(B * Tq, heads * d, dtype=Q.dtype, device=Q.device)
    check(exp_lib().bltvqg_attn_out_fwd(ptr(Q), Q.stride(0), ptr(K), K.stride(0), ptr(V), V.stride(0), ptr(O), O.stride(0), ptr(Wo), Wo.stride(0),
                                    ptr(R), 0 if R is None else R.stride(0), ptr(Y), Y.stride(0), ptr(key_ids), B, heads, Tq, Tk, d, int(causal),
                                    float(scale), float(drop_p), int(seed), int(stream_id), stream_ptr()), "attn_out_fwd")
    return O, Y


def attn_bwd(Q, K, V, dO, key_ids, B, heads, Tq, Tk, d, causal, scale, drop_p=0.0, seed=0, stream_id=0):
    dQ, dK, dV = torch.zeros_like(Q), torch.zeros_like(K), torch.zeros_like(V)
    check(lib().bltvqg_attn_bwd(DT[Q.dtype], ptr(Q), Q.stride(0), ptr(K), K.stride(0), ptr(V), V.stride(0), ptr(dO), dO.stride(0),
                                ptr(dQ), dQ.stride(0), ptr(dK), dK.stride(0), ptr(dV), dV.stride(0), ptr(key_ids), B, heads, Tq, Tk, d,
                                int(causal), float(scale), float(drop_p), int(seed), int(stream_id), stream_ptr()), "attn_bwd")
    return dQ, dK, dV


def dropout_mask(seed, stream_id, rows, cols, ld_index, p, device="cuda"):
    out = torch.empty(rows, cols, dtype=torch.uint8, device=device)
    check(lib().bltvqg_dropout_mask(int(seed), int(stream_id), rows, cols, ld_index, float(p), ptr(out), stream_ptr()), "dropout_mask")
    return out
