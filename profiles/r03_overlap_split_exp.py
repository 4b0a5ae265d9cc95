"""Round-3 experiment 3: how much of the conv stack to run ahead, and where (big cfg, B=256, bf16, unmasked)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bltvqg_amd.synthetic as synthetic
from bltvqg_amd.engine import StepEngine, make_config
from bltvqg_amd.trainer import init_reference_style

dev = torch.device("cuda", 0)
B = int(os.environ.get("B", 256))
c = make_config(B, 512, 2048, 512, 300, 6, 8, 8000, dtype=1)
eng = StepEngine(c, dev); eng.allocate(); init_reference_style(eng, seed=0); eng.trust_shadows(True)
batch = synthetic.make_batch(B, 8000, 512, seed=1234, image_hw=224)
d = {k: batch[k].to(dev) for k in ("images", "answers", "posteriors", "questions")}
gen = torch.Generator(device=dev).manual_seed(99)

def loop(n, mode):
    for i in range(n):
        eps = torch.randn(B, 512, device=dev, generator=gen)
        if mode == "inline":
            eng.forward(d["images"], d["answers"], d["posteriors"], d["questions"], eps, True, 1000 + i)
            eng.loss_backward(0.5)
        else:
            if eng.prefetch_pending() == 0: eng.prefetch_images(d["images"])
            if mode == "before": eng.prefetch_images(d["images"])
            eng.forward(None, d["answers"], d["posteriors"], d["questions"], eps, True, 1000 + i)
            if mode == "after_fwd": eng.prefetch_images(d["images"])
            eng.loss_backward(0.5)
            if mode == "after_bwd": eng.prefetch_images(d["images"])
        eng.optimizer_step(1e-4, 5.0, overlap=True)
    eng.optimizer_wait()

def drain():
    while eng.prefetch_pending() > 0:
        eps = torch.randn(B, 512, device=dev, generator=gen)
        eng.forward(None, d["answers"], d["posteriors"], d["questions"], eps, True, 1); eng.loss_backward(0.5)
    torch.cuda.synchronize()

def timeit(label, mode, n=30):
    loop(6, mode); torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        t0 = time.perf_counter(); loop(n, mode); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    print("%-40s %.3f ms/step" % (label, best), flush=True)
    drain()

timeit("inline", "inline")
for split in (10, 1, 2, 3, 5, 7, 9):
    eng.set_prefetch_split(split)
    for mode in (("before",) if split == 10 else ()) + ("after_fwd", "after_bwd"):
        timeit("split %2d %s" % (split, mode), mode)
timeit("inline", "inline")
