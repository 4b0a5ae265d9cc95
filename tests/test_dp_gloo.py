"""Data-parallel host logic on CPU: two gloo ranks (world_size 2) exercise the bucketed gradient mean, the phase-1 bucket
skip, identical initial weights on every rank and per-rank batch sharding.  (The GPU path uses the same functions with RCCL.)"""
import os
import socket
from types import SimpleNamespace

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


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
    import bltvqg_amd  # noqa: F401
    from bltvqg_amd import synthetic
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.iq import IQ
    from bltvqg_amd.train_iq import SyntheticVocabulary
    from bltvqg_amd.trainer import active_buckets, allreduce_bucket, comm_plan, shard_seed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = make_config(4, 64, 128, 64, 20, 1, 4, 97, image_hw=(64, 64), dtype=0)
        eng = StepEngine(cfg, "cpu")                       # host-side descriptor only: layout + buckets, no device memory
        buckets = eng.buckets()
        n = eng.train_size
        assert sum(b[1] for b in buckets) == n and buckets[0][0] == 0 and buckets[2][2] == 1
        assert buckets[2][0] == eng.late_offset
        res = {}
        for phase2 in (False, True):
            g = torch.full((n,), float(rank + 1)) + torch.arange(n, dtype=torch.float32) * 1e-3 * (rank + 1)
            for i, off, cnt in active_buckets(buckets, phase2):
                allreduce_bucket(dist, g, off, cnt)
            want = torch.full((n,), 1.5) + torch.arange(n, dtype=torch.float32) * 1e-3 * 1.5
            lo = eng.late_offset
            assert torch.allclose(g[:lo], want[:lo], rtol=1e-6)
            if phase2:
                assert torch.allclose(g[lo:], want[lo:], rtol=1e-6)
            else:   # untouched before the phase switch
                assert torch.equal(g[lo:], torch.full((n - lo,), float(rank + 1)) + torch.arange(lo, n, dtype=torch.float32) * 1e-3 * (rank + 1))
            # the step's actual collectives (DataParallelStep.reduce_gradients): decoder bucket alone, the rest merged into one
            plan = comm_plan(buckets, phase2)
            assert len(plan) == 2 and plan[0] == ([0], buckets[0][0], buckets[0][1])
            assert plan[1][0] == ([1, 2] if phase2 else [1]) and plan[1][1] == buckets[1][0]
            assert plan[1][2] == (buckets[1][1] + buckets[2][1] if phase2 else buckets[1][1])
            g2 = torch.full((n,), float(rank + 1)) + torch.arange(n, dtype=torch.float32) * 1e-3 * (rank + 1)
            for ids, off, cnt in plan:
                allreduce_bucket(dist, g2, off, cnt)
            assert torch.equal(g2, g)
            res[phase2] = True
        # identical initial weights on every rank (same seed), different data shards
        args = SimpleNamespace(emb_dim=20, hidden_dim=64, latent_dim=64, pwffn_dim=128, num_layers=1, num_heads=4, device="cpu", emb_file=None,
                               root_dir=".", seed=3)
        m = IQ(False, SyntheticVocabulary(97), args)
        chk = torch.tensor([float(m._flat_train.double().sum()), float(m._flat_frozen.double().sum())], dtype=torch.float64)
        gathered = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(gathered, chk)
        assert torch.equal(gathered[0], gathered[1])
        b = synthetic.make_batch(4, 97, 64, seed=shard_seed(1234, rank), image_hw=64)
        q = b["questions"].double().sum().reshape(1)
        qs = [torch.zeros_like(q) for _ in range(world)]
        dist.all_gather(qs, q)
        assert float(qs[0]) != float(qs[1])
        out[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_bucketed_gradient_mean_world_size_2():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: "ok", 1: "ok"}
