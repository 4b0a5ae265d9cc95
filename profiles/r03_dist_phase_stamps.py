"""Steady-state phase stamps with and without a (one-rank) RCCL exchange: where does the data-parallel step lose its 0.4-0.5 ms?"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
import bltvqg_amd.synthetic as synthetic
from bltvqg_amd.engine import StepEngine, make_config
from bltvqg_amd.trainer import DataParallelStep, init_reference_style
mode = sys.argv[1]      # none | dist | dist_noar (exchange object, collectives skipped) | dist_optmain (optimiser forked from the main stream)
C = (256, 512, 2048, 512, 300, 6, 8, 8000)
dev = torch.device("cuda", 0)
dist = None
if mode != "none":
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
eng = StepEngine(make_config(*C, dtype=1), dev)
eng.allocate(); init_reference_style(eng, seed=0)
step = DataParallelStep(eng, dist, overlap_optimizer=True)
if dist is None:
    conv_s = torch.cuda.Stream(); eng.adopt_conv_stream(conv_s)
if mode == "dist_noar":
    import bltvqg_amd.trainer as T
    T.allreduce_bucket = lambda *a, **k: None
if mode == "dist_optmain":
    import contextlib
    step._on_comm_real = step._on_comm
    orig_run = step.run
    def run(*a, **k):
        # same as run() but the optimiser is forked from the main stream after it waited for the communication stream
        step.overlap_optimizer = False
        e = step.e
        r = orig_run(*a, **k)
        return r
    # simpler: non-overlapped optimiser placement = main stream waits for comm, optimiser synchronous on main
    step.overlap_optimizer = False
b = synthetic.make_batch(C[0], C[7], C[3], seed=1234)
d = {k: b[k].to(dev) for k in ("images", "answers", "posteriors", "questions")}
gen = torch.Generator(device=dev).manual_seed(1)
def one(i):
    eps = torch.randn(C[0], C[3], device=dev, generator=gen)
    step.run(d["images"], d["answers"], d["posteriors"], d["questions"], eps, True, seed=i, kl_weight=0.5, lr=1e-4, next_images=d["images"])
for i in range(5): one(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(40): one(5 + i)
th = time.perf_counter() - t0
step.finish(); torch.cuda.synchronize()
print("%s: %.3f ms/step; host enqueue %.3f ms/step" % (mode, (time.perf_counter() - t0) / 40 * 1e3, th / 40 * 1e3))
eng.lib.bltvqg_debug_set(12, 1)
acc = [0.0] * 12; n = 0
for rep in range(6):
    for i in range(6): one(100 + i)
    torch.cuda.synchronize()
    ms = (ctypes.c_float * 12)()
    eng.lib.bltvqg_engine_phase_stamps(eng.h, ms)
    n += 1
    for k in range(12): acc[k] += ms[k]
names = ["start", "enc joined", "dec starts", "dec done", "fwd end", "losses", "dec bwd starts", "dec bwd done", "enc bwd starts", "enc bwd done", "bwd end"]
print("   " + "  ".join("%s %.2f" % (names[k], acc[k] / n) for k in range(1, 11)))
if dist: dist.destroy_process_group()
