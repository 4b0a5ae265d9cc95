"""Two ranks driving the REAL engine: over RCCL on two GPUs where there are two, otherwise both ranks on the ONE GPU of the test box with the
gloo backend moving the (device-resident) gradients — the same DataParallelStep code path: in-stack weight-gradient flushes, one event per
gradient bucket, the communication stream, the optimiser forked from it.  (tests/test_dp_gloo.py covers the host logic on CPU with a
stand-in engine.)  Different shards per rank -> the all-reduced flat gradient equals the mean of the
two single-rank gradients, the phase-1 step leaves the phase-2-only bucket untouched, and the overlapped optimiser schedule (clip +
Adam forked from the communication stream) gives the synchronous result.  Reference: pl.Trainer(gpus=N) = Lightning DDP,
train_iq.py:372-373."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    sys.path.insert(0, os.path.join(root, "tests", "golden"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch.distributed as dist
    import bltvqg_amd.synthetic as synthetic
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.trainer import DataParallelStep, init_reference_style, shard_seed
    shared = torch.cuda.device_count() < world                   # one-GPU box: both ranks share cuda:0, gloo carries the exchange
    dev = torch.device("cuda", 0 if shared else rank)
    torch.cuda.set_device(dev)
    if shared:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        B, V, Z = 8, 97, 64
        cfg = make_config(B, 64, 128, Z, 20, 2, 4, V, image_hw=(64, 64), dtype=0, attention_dropout=0.0, relu_dropout=0.0)
        finals = {}
        for overlap in (False, True):
            e = StepEngine(cfg, dev)
            e.allocate()
            init_reference_style(e, seed=5 + rank)              # ranks start apart: the broadcast must equalise them
            dp = DataParallelStep(e, dist, overlap_optimizer=overlap)
            assert len(dp.buckets) >= 5 and dp.comm is not None
            for stepi, phase2 in enumerate((False, True, True)):
                b = synthetic.make_batch(B, V, Z, seed=shard_seed(100 + stepi, rank), image_hw=64)
                d = {k: v.to(dev) for k, v in b.items()}
                # single-rank gradient of this shard (same engine, no exchange)
                e.optimizer_wait()
                e.forward(d["images"], d["answers"], d["posteriors"], d["questions"], d["eps"] if phase2 else None, phase2, 7)
                e.loss_backward(0.25)
                torch.cuda.synchronize()
                mine = e.flat_grad.clone()
                both = [torch.zeros_like(mine) for _ in range(world)]
                dist.all_gather(both, mine)
                want = sum(both) / world
                # BatchNorm running statistics were advanced by the probe forward: the step below repeats the forward, which is fine for
                # the gradient (train-mode statistics come from the batch)
                e.forward(d["images"], d["answers"], d["posteriors"], d["questions"], d["eps"] if phase2 else None, phase2, 7)
                e.loss_backward(0.25)
                dp.reduce_gradients(phase2)
                torch.cuda.current_stream().wait_stream(dp.comm)
                torch.cuda.synchronize()
                got = e.flat_grad
                lo = e.late_offset
                hi = e.train_size if phase2 else lo
                err = float((got[:hi] - want[:hi]).abs().max()) / max(float(want[:hi].abs().max()), 1e-12)
                assert err < 1e-5, (stepi, err)
                if not phase2:
                    assert float(got[lo:].abs().max()) == 0.0
                if overlap:
                    with torch.cuda.stream(dp.comm):
                        e.optimizer_step(1e-3, 5.0, overlap=True)
                else:
                    e.optimizer_step(1e-3, 5.0)
            dp.finish()
            torch.cuda.synchronize()
            chk = [torch.zeros_like(e.flat_train) for _ in range(world)]
            dist.all_gather(chk, e.flat_train)
            assert torch.equal(chk[0], chk[1])                   # replicas stay identical
            finals[overlap] = e.flat_train.clone()
        d = float((finals[False] - finals[True]).abs().max())
        assert d <= 2.5 * 3e-3, d                                 # Adam turns rounding-level gradient noise into +-lr moves
        out[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_two_rank_engine_gradients_are_the_mean_of_the_single_rank_gradients():
    import torch.multiprocessing as mp
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: "ok", 1: "ok"}


@pytest.mark.parametrize("dtype", [1, 0])
def test_in_stack_flush_waits_for_the_branch_stream(dtype):
    """with in-stack weight-gradient flushes (the data-parallel mode) the decoder's FIRST flush carries the image
    reconstructor's and z_classifier's weight gradients, whose operands are written on the branch stream (side 0).  The weight-gradient
    stream must order itself behind that branch: with the branch delayed by a long kernel, the flushed run must still give the
    gradients of the unflushed one.  BASELINE configs[2] widths (the in-stack bucket boundaries exist from ~32 MB of layers on)."""
    import ctypes
    import bltvqg_amd.synthetic as synthetic
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.trainer import init_reference_style
    B, hw = 4, 64
    c = make_config(B, 512, 2048, 512, 300, 6, 8, 8000, image_hw=(hw, hw), dtype=dtype, attention_dropout=0.0, relu_dropout=0.0)
    # a DIFFERENT batch every repetition: operands left over from the previous step must not pass for the current ones
    ds = [{k: v.cuda() for k, v in synthetic.make_batch(B, 8000, 512, seed=11 + r, image_hw=hw).items() if torch.is_tensor(v)} for r in range(3)]
    grads = {}
    for flush in (0, 1):
        e = StepEngine(c)
        e.allocate()
        init_reference_style(e, seed=3)
        assert any(late == 0 for _, _, late in e.buckets()[:3])
        e.set_bucket_flush(bool(flush))
        for rep in range(3):
            d = ds[rep]
            e.forward(d["images"], d["answers"], d["posteriors"], d["questions"], d["eps"], True, 5)
            if flush:      # delay the branch stream: a few ms of sleep in front of the loss kernels it runs
                p = ctypes.c_void_p()
                assert e.lib.bltvqg_engine_side_stream(e.h, 0, ctypes.byref(p)) == 0
                with torch.cuda.stream(torch.cuda.ExternalStream(p.value)):
                    torch.cuda._sleep(8_000_000)
            e.loss_backward(0.4)
            torch.cuda.synchronize()
        grads[flush] = e.flat_grad.clone()
        info = dict(e.train_info)
        del e
    for n in ("image_reconstructor.layers.fc1.weight", "image_reconstructor.layers.fc0.weight", "image_reconstructor.layers.fc0.bias",
              "decoder.z_classifier.weight", "decoder.z_classifier.bias", "decoder.output.weight"):
        i = info[n]
        a, b = grads[0][i.offset:i.offset + i.numel], grads[1][i.offset:i.offset + i.numel]
        assert float(a.abs().max()) > 0.0
        assert float((a - b).abs().max()) <= 2e-4 * float(a.abs().max()), n
