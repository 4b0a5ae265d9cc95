"""How many streams really run concurrently?  k streams x one 1-workgroup kernel spinning 300 us each."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bltvqg_amd import _lib
from bltvqg_amd._lib import check, ptr
lib = _lib.load()
hip = ctypes.CDLL("libamdhip64.so")
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
out = torch.zeros(64, dtype=torch.int32, device="cuda")
def mk(kind):
    if kind == "torch":
        return torch.cuda.Stream()
    st = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0
    return torch.cuda.ExternalStream(st.value)
for kind in ("hip", "torch"):
    streams = [mk(kind) for _ in range(12)]
    for k in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12):
        torch.cuda.synchronize()
        for rep in range(2):
            t0 = time.perf_counter()
            for s in streams[:k]:
                check(_lib.load_experiments().bltvqg_hw_id_probe(ptr(out), 1, 30000, ctypes.c_void_p(s.cuda_stream)), "probe")
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) * 1e6
        print("%s streams: %2d x 300 us kernels -> %.0f us" % (kind, k, dt), flush=True)
