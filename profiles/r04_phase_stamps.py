"""Steady-state phase stamps (engine debug key 12) without per-step synchronisation, conv stack one batch ahead; + host enqueue time."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))      # (profiles/ and scratch/ are both one level below the repository root)
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bltvqg_amd.synthetic as synthetic
from bltvqg_amd.engine import StepEngine, make_config
from bltvqg_amd.trainer import DataParallelStep, init_reference_style
C = (256, 512, 2048, 512, 300, 6, 8, 8000)
dev = torch.device("cuda", 0)
eng = StepEngine(make_config(*C, dtype=1), dev)
for kv in os.environ.get("BLT_DEBUG", "").split(","):
    if kv: eng.lib.bltvqg_debug_set(int(kv.split("=")[0]), int(kv.split("=")[1]))
eng.allocate(); init_reference_style(eng, seed=0)
conv_s = torch.cuda.Stream(); eng.adopt_conv_stream(conv_s)
step = DataParallelStep(eng, None, overlap_optimizer=True)
b = synthetic.make_batch(C[0], C[7], C[3], seed=1234)
d = {k: b[k].to(dev) for k in ("images", "answers", "posteriors", "questions")}
gen = torch.Generator(device=dev).manual_seed(1)
pre = os.environ.get("PREFETCH", "1") == "1"
def one(i):
    eps = torch.randn(C[0], C[3], device=dev, generator=gen)
    step.run(d["images"], d["answers"], d["posteriors"], d["questions"], eps, True, seed=i, kl_weight=0.5, lr=1e-4, next_images=d["images"] if pre else None)
for i in range(5): one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(30): one(5 + i)
th = time.perf_counter() - t0
step.finish(); torch.cuda.synchronize()
print("prefetch=%s: %.3f ms/step; host enqueue %.3f ms/step" % (pre, (time.perf_counter() - t0) / 30 * 1e3, th / 30 * 1e3))
eng.lib.bltvqg_debug_set(12, 1)
acc = [0.0] * 12; n = 0
for rep in range(6):
    for i in range(6): one(100 + i)
    torch.cuda.synchronize()
    ms = (ctypes.c_float * 12)()
    eng.lib.bltvqg_engine_phase_stamps(eng.h, ms)
    n += 1
    for k in range(12): acc[k] += ms[k]
names = ["start", "cnn+enc joined", "decoder starts", "decoder done", "forward end", "losses", "dec bwd starts", "dec bwd done", "enc bwd starts", "enc bwd done", "backward end"]
print("%-18s at %7.3f ms" % ("image feature done", acc[11] / n))
prev = 0.0
for k in range(1, 11):
    v = acc[k] / n
    print("%-18s at %7.3f ms  (+%.3f)" % (names[k], v, v - prev)); prev = v
