"""Per-launch durations of the planned-tile GEMM for the step's shapes and epilogue forms:
   rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 scratch/mb4.py ; python3 scratch/mb4.py --parse DIR/.../*_kernel_trace.csv"""
import csv, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
# (M, N, K, form): plain | res (R) | ffn1 (bias+relu+drop) | ffn2 (bias+relu+drop+R) | mask (maskY: dgrad through relu) | bias
CASES = [(5120, 512, 512, "plain"), (5120, 512, 512, "res"), (5120, 1536, 512, "plain"), (5120, 2048, 512, "ffn1"), (5120, 512, 2048, "ffn2"),
         (5120, 2048, 512, "mask"), (5120, 512, 2048, "plain"), (5376, 512, 512, "res"), (5376, 1536, 512, "plain"), (5376, 2048, 512, "ffn1"),
         (5376, 512, 2048, "ffn2"), (1280, 512, 512, "res"), (1280, 1536, 512, "plain"), (1280, 2048, 512, "ffn1"), (1280, 512, 2048, "ffn2"),
         (5120, 8000, 512, "bias"), (5120, 512, 8000, "plain"), (256, 1024, 1024, "bias")]
REPS = 14

if len(sys.argv) > 2 and sys.argv[1] == "--parse":
    rows = []
    with open(sys.argv[2]) as fh:
        for r in csv.DictReader(fh):
            n = r["Kernel_Name"]
            if "gemm_nt2_kernel" in n or "gemm_dma_kernel" in n or "gemm_kernel" in n:
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), n))
    rows.sort()
    assert len(rows) == REPS * len(CASES), (len(rows), REPS * len(CASES))
    for i, (M, N, K, form) in enumerate(CASES):
        d = sorted(x[1] for x in rows[i * REPS + 2:(i + 1) * REPS])
        us = d[len(d) // 2] / 1e3
        print("%5d x %4d x %4d %-5s %7.2f us (min %6.2f) %6.0f TF/s  [%s]" % (M, N, K, form, us, d[0] / 1e3, 2.0 * M * N * K / us / 1e6, rows[i * REPS + 2][2][40:90]))
    sys.exit(0)

import torch
from gpu_ops import gemm
torch.manual_seed(0)
for (M, N, K, form) in CASES:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    R = torch.randn(M, N, device="cuda").bfloat16(); bias = torch.randn(N, device="cuda")
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    kw = {}
    if form == "res": kw = dict(R=R)
    elif form == "bias": kw = dict(bias=bias)
    elif form == "ffn1": kw = dict(bias=bias, relu=True, drop_p=0.1, seed=5, stream_id=3)
    elif form == "ffn2": kw = dict(bias=bias, relu=True, drop_p=0.1, seed=5, stream_id=3, R=R)
    elif form == "mask": kw = dict(maskY=R, mask_scale=1.1)
    for _ in range(REPS):
        gemm(A, W, M, N, K, C=C, **kw)
    torch.cuda.synchronize()
print("done")
