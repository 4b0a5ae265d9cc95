"""Per-kernel cost of running the chain on a CU partition: python scratch/mask_prof.py <conv_cus_per_xcd|0> (under rocprofv3 --stats)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bltvqg_amd.synthetic as synthetic
from bltvqg_amd.engine import StepEngine, make_config
from bltvqg_amd.trainer import init_reference_style
k = int(sys.argv[1])
dev = torch.device("cuda", 0)
B = 256
c = make_config(B, 512, 2048, 512, 300, 6, 8, 8000, dtype=1)
eng = StepEngine(c, dev); eng.allocate(); init_reference_style(eng, seed=0); eng.trust_shadows(True)
batch = synthetic.make_batch(B, 8000, 512, seed=1234, image_hw=224)
d = {kk: batch[kk].to(dev) for kk in ("images", "answers", "posteriors", "questions")}
gen = torch.Generator(device=dev).manual_seed(99)
if k:
    chain = StepEngine.cu_mask(0, 32 - k)
    eng.set_cu_masks(chain=chain, side=chain, conv=StepEngine.cu_mask(32 - k, 32), chain_cus=8 * (32 - k))
s = eng.chain_stream() if k else torch.cuda.current_stream()
with torch.cuda.stream(s):
    for i in range(10):
        eps = torch.randn(B, 512, device=dev, generator=gen)
        # conv stack and chain strictly one after the other: per-kernel durations without cross-stream contention
        eng.prefetch_images(d["images"]); torch.cuda.synchronize()
        eng.forward(None, d["answers"], d["posteriors"], d["questions"], eps, True, 1000 + i)
        eng.loss_backward(0.5)
        eng.optimizer_step(1e-4, 5.0, overlap=True)
        eng.optimizer_wait(); torch.cuda.synchronize()
