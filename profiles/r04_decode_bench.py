import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (profiles/ and scratch/ are one level below the root)
sys.path.insert(0, ROOT)
import torch
import bltvqg_amd.synthetic as synthetic
from bltvqg_amd.engine import StepEngine, make_config
from bltvqg_amd.trainer import init_reference_style
V = 8000
for (B, H, F, L, NH, tag, dtype) in ((128, 256, 512, 2, 4, "small", 1), (256, 512, 2048, 6, 8, "big", 1), (64, 256, 512, 2, 4, "small-f32", 0)):
    for T in (21, 51):
        c = make_config(B, H, F, H, 300, L, NH, V, len_target=T, dtype=dtype, attention_dropout=0.0, relu_dropout=0.0)
        e = StepEngine(c); e.allocate(); init_reference_style(e, seed=0)
        b = synthetic.make_batch(B, V, H, seed=1)
        img, ans, eps = b["images"].cuda(), b["answers"].cuda(), b["eps"].cuda()
        ref = None
        for key in (1, 0, 1, 0):
            e.lib.bltvqg_debug_set(29, key)
            for _ in range(2): out = e.decode_greedy(img, ans, eps, True)
            torch.cuda.synchronize(); t0 = time.time(); n = 6
            for _ in range(n): out = e.decode_greedy(img, ans, eps, True)
            torch.cuda.synchronize(); ms = (time.time() - t0) / n * 1e3
            if ref is None: ref = [x.clone() for x in out]
            same = bool(torch.equal(ref[0], out[0])), bool(torch.equal(ref[1], out[1])), float((ref[2] - out[2]).abs().max())
            print("%s B=%d T=%d full-redecode=%d: %.2f ms per batch -> %.0f questions/s (tokens same %s, top6 ids same %s, max |dp| %.2e)" % (tag, B, T, key, ms, B / ms * 1e3, same[0], same[1], same[2]), flush=True)
        e.lib.bltvqg_debug_set(29, 0)
        del e
