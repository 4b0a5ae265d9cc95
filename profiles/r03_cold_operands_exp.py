"""What a planned-tile Linear GEMM pays for COLD operands (bltvqg_gemm_rotate): hot = same A, B, C every launch (bltvqg_gemm_repeat);
B rotated over n copies (another layer's weights each launch: n x |B| larger than the 4 MB L2 / the 256 MB Infinity Cache); A rotated;
chain = each launch reads what the previous one wrote."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bltvqg_amd import _lib
from bltvqg_amd._lib import ptr, stream_ptr, check
lib = _lib.load()
REPS = 240
def t(fn):
    fn(20); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record(); fn(REPS); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / REPS * 1e3)
    return best
for (M, N, K) in [(5120, 512, 512), (5120, 2048, 512), (5120, 512, 2048)]:
    na, nb = 48, 600
    A = torch.randn(na, M, K, device="cuda").bfloat16()
    B = (torch.randn(nb, N, K, device="cuda") * 0.05).bfloat16()
    C = torch.zeros(max(na, 2), M, N, device="cuda", dtype=torch.bfloat16)
    sa, sb, sc = M * K * 2, N * K * 2, M * N * 2
    def run(ac, bc, cc, chain):
        return lambda n: check(_lib.load_experiments().bltvqg_gemm_rotate(1, ptr(A), K, ac, sa, ptr(B), K, bc, sb, ptr(C), N, cc, sc, M, N, K, chain, n, stream_ptr()), "rot")
    res = [("hot", t(run(1, 1, 1, 0))), ("B x8 (L2-sized set)", t(run(1, 8, 1, 0))), ("B x64 (fits the Infinity Cache)", t(run(1, 64, 1, 0))),
           ("B x600 (HBM)", t(run(1, nb, 1, 0))), ("A x48 + C x48", t(run(na, 1, na, 0))), ("A x48 + B x600 + C x48", t(run(na, nb, na, 0)))]
    if N == K:
        res.append(("chain (A = previous C), B hot", t(run(1, 1, 2, 1))))
        res.append(("chain, B x600", t(run(1, nb, 2, 1))))
    print("M=%d N=%d K=%d: " % (M, N, K) + " | ".join("%s %.1f us" % r for r in res), flush=True)
    del A, B, C
