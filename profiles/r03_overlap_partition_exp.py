"""Round-3 experiment 2: conv stack alone and chain alone as a function of the CU partition (big cfg, B=256, bf16)."""
import ctypes, os, sys, time
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
masked = False

def S():
    return eng.chain_stream() if masked else torch.cuda.current_stream()

def chain_step(i):
    eps = torch.randn(B, 512, device=dev, generator=gen)
    eng.forward(None, d["answers"], d["posteriors"], d["questions"], eps, True, 1000 + i)
    eng.loss_backward(0.5)
    eng.optimizer_step(1e-4, 5.0, overlap=True)

def measure(label, n=12):
    with torch.cuda.stream(S()):
        # conv stack alone (nothing else running)
        tc = []
        for i in range(n):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            eng.prefetch_images(d["images"]); torch.cuda.synchronize(); tc.append(time.perf_counter() - t0)
            # chain alone (its conv stack is done)
            t0 = time.perf_counter(); chain_step(i); eng.optimizer_wait(); torch.cuda.synchronize(); tc[-1] = (tc[-1], time.perf_counter() - t0)
        cnn = sorted(x[0] for x in tc[2:])[len(tc[2:]) // 2] * 1e3
        ch = sorted(x[1] for x in tc[2:])[len(tc[2:]) // 2] * 1e3
        # overlapped steady state
        eng.prefetch_images(d["images"])
        for i in range(5):
            eng.prefetch_images(d["images"]); chain_step(i)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(30):
            eng.prefetch_images(d["images"]); chain_step(i)
        eng.optimizer_wait(); torch.cuda.synchronize()
        ov = (time.perf_counter() - t0) / 30 * 1e3
        # drain
        eps = torch.randn(B, 512, device=dev, generator=gen)
        eng.forward(None, d["answers"], d["posteriors"], d["questions"], eps, True, 1); eng.loss_backward(0.5); torch.cuda.synchronize()
    print("%-44s conv alone %.3f  chain alone %.3f  overlapped %.3f ms" % (label, cnn, ch, ov), flush=True)

measure("unmasked")
for k, plan in ((8, None), (7, None), (8, 256), (9, None)):
    conv = StepEngine.cu_mask(32 - k, 32); chain = StepEngine.cu_mask(0, 32 - k)
    masked = True
    eng.set_cu_masks(chain=chain, side=chain, conv=conv, chain_cus=plan or 8 * (32 - k))
    measure("conv %2d/XCD chain %2d/XCD plan_cus %3d" % (k, 32 - k, plan or 8 * (32 - k)))
