"""gemm_repeat of a few shapes (for a rocprofv3 --pmc pass): python3 scratch/mb_pmc.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bltvqg_amd import _lib
from bltvqg_amd._lib import ptr, stream_ptr, check
lib = _lib.load()
for (M, N, K) in [(5120, 2048, 512), (5120, 512, 512), (5120, 512, 2048)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    R = torch.randn(M, N, device="cuda").bfloat16(); bias = torch.randn(N, device="cuda"); C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    check(lib.bltvqg_gemm_repeat(1, ptr(A), K, ptr(W), K, ptr(C), N, M, N, K, ptr(bias), 1, ptr(R), N, 30, stream_ptr()), "rep")
    torch.cuda.synchronize()
