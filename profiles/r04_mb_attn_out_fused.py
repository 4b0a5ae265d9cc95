"""attention + output projection: two launches (attn_fwd, planned-tile GEMM with residual) against the fused one-launch experiment
(bltvqg_attn_out_fwd, experiments build): per-launch device time from a rocprofv3 kernel trace."""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
REPS = 12
if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    from collections import defaultdict
    d = defaultdict(list)
    for r in csv.DictReader(open(sys.argv[2])):
        n = r["Kernel_Name"]
        for key in ("attn_out_fwd_kernel", "attn_fwd_mfma_kernel", "gemm_nt2_kernel"):
            if key in n: d[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in d.items():
        v = sorted(v[2:]); print("%-24s %3d launches, median %6.2f us, min %6.2f us" % (k, len(v), v[len(v) // 2] / 1e3, v[0] / 1e3))
    sys.exit(0)
import torch
import gpu_ops as G
B, h, T, d = 256, 8, 20, 64
Hd = h * d
g = torch.Generator().manual_seed(1)
Q = torch.randn(B * T, Hd, generator=g).bfloat16().cuda(); kv = torch.randn(B * T, 2 * Hd, generator=g).bfloat16().cuda()
K, V = kv[:, :Hd], kv[:, Hd:]
Wo = (torch.randn(Hd, Hd, generator=g) * Hd ** -0.5).bfloat16().cuda(); R = torch.randn(B * T, Hd, generator=g).bfloat16().cuda()
ids = torch.randint(1, 50, (B, T), generator=g, dtype=torch.int32).cuda()
for _ in range(REPS):
    O2 = G.attn_fwd(Q, K, V, ids, B, h, T, T, d, 1, d ** -0.5, 0.1, 99, 31)
    Y2 = G.gemm(O2, Wo, B * T, Hd, Hd, R=R)
for _ in range(REPS):
    O1, Y1 = G.attn_out_fwd(Q, K, V, Wo, R, ids, B, h, T, T, d, 1, d ** -0.5, 0.1, 99, 31)
torch.cuda.synchronize()
assert torch.equal(Y1, Y2)
print("ok")
