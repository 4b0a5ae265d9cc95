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
        # buckets come in backward-COMPLETION order (decoder groups, latent-phase heads, encoder stacks, tail), not in address order;
        # together they tile the flat buffer exactly once, the late ones tile the phase-2-only region
        assert sum(b[1] for b in buckets) == n and buckets[0][0] == 0
        pos = 0
        for off, cnt, late in sorted(buckets):
            assert off == pos and (late == 1) == (off >= eng.late_offset)
            pos += cnt
        assert pos == n and buckets[-1][2] == 0 and any(b[2] for b in buckets)
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
            # the step's actual collectives (DataParallelStep.reduce_gradients): one per active bucket, each waiting on its own event only
            plan = comm_plan(buckets, phase2)
            assert [p_[0] for p_ in plan] == [[i] for i, b in enumerate(buckets) if phase2 or not b[2]]
            assert all((off, cnt) == (buckets[ids[0]][0], buckets[ids[0]][1]) for ids, off, cnt in plan)
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


class _StubEngine(object):
    """CPU stand-in with the StepEngine surface DataParallelStep uses: the 'gradient' is a deterministic function of the rank's batch,
    the 'optimiser' is plain SGD with norm clipping, and every call is logged so that the ORDER of the step can be checked."""

    def __init__(self, buckets, n, late_offset):
        self.device = torch.device("cpu")
        self._buckets, self.late_offset = buckets, late_offset
        self.flat_train = torch.zeros(n)
        self.flat_frozen = torch.zeros(8)
        self.flat_grad = torch.zeros(n)
        self.log = []
        self.seeds = []

    def buckets(self):
        return self._buckets

    def bucket_wait(self, i, stream):
        assert stream is None                      # CPU: no communication stream
        self.log.append(("wait", i))

    def forward(self, images, context, posterior, target, eps, phase2, seed):
        self.log.append(("forward", bool(phase2)))
        self.seeds.append(seed)
        self._x, self._phase2 = float(images.sum()), bool(phase2)

    def loss_backward(self, kl_weight):
        self.log.append(("backward",))
        n = self.flat_grad.numel()
        self.flat_grad.zero_()
        hi = n if self._phase2 else self.late_offset          # phase 1: the latent-phase parameters receive no gradient
        self.flat_grad[:hi] = self._x * (1.0 + torch.arange(hi, dtype=torch.float32) / n) + self.flat_train[:hi]

    def optimizer_step(self, lr, max_norm, overlap=False):
        self.log.append(("opt", bool(overlap), self.flat_grad.clone()))
        g = self.flat_grad
        scale = min(1.0, max_norm / (float(g.norm()) + 1e-6))
        self.flat_train -= lr * scale * g

    def optimizer_wait(self):
        self.log.append(("opt_wait",))


def _dp_worker(rank, world, port, out):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bltvqg_amd  # noqa: F401
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.trainer import DataParallelStep, rank_dropout_seed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cfg = make_config(4, 64, 128, 64, 20, 1, 4, 97, image_hw=(64, 64), dtype=0)
        real = StepEngine(cfg, "cpu")                        # the real engine's bucket layout (host-side descriptor only)
        buckets, n, lo = real.buckets(), real.train_size, real.late_offset
        finals = {}
        for overlap in (False, True):
            e = _StubEngine(buckets, n, lo)
            e.flat_train += float(rank)                       # ranks start apart: the constructor's broadcast must make them equal
            dp = DataParallelStep(e, dist, overlap_optimizer=overlap)
            assert float(e.flat_train.abs().max()) == 0.0     # rank 0's parameters everywhere
            singles = []                                      # what each rank's gradient would be on its own
            for stepi, phase2 in enumerate((False, True, True)):
                x = torch.full((2, 2), float(10 * rank + stepi + 1))
                before = e.flat_train.clone()
                dp.run(x, None, None, None, None, phase2, seed=500 + stepi, kl_weight=0.5, lr=0.1, max_norm=1e9)
                # the gradient the optimiser saw = mean over ranks of the single-rank gradients (DDP semantics)
                hi = n if phase2 else lo
                mine = torch.zeros(n)
                mine[:hi] = float(x.sum()) * (1.0 + torch.arange(hi, dtype=torch.float32) / n) + before[:hi]
                both = [torch.zeros(n) for _ in range(world)]
                dist.all_gather(both, mine)
                want = sum(both) / world
                seen = [r for r in e.log if r[0] == "opt"][-1][2]
                assert torch.allclose(seen, want, rtol=1e-6, atol=1e-6), (stepi, float((seen - want).abs().max()))
                if not phase2:
                    assert float(seen[lo:].abs().max()) == 0.0          # the phase-2-only bucket is neither produced nor exchanged
                singles.append(want)
                # order inside one step: forward, backward, one wait per active bucket in completion order, optimiser
                act = [i for i, b in enumerate(buckets) if phase2 or not b[2]]
                names = [r[0] if r[0] != "wait" else "wait%d" % r[1] for r in e.log[-(3 + len(act)):]]
                assert names == ["forward", "backward"] + ["wait%d" % i for i in act] + ["opt"], names
                assert [r for r in e.log if r[0] == "opt"][-1][1] == overlap
            dp.finish()
            assert e.log[-1] == ("opt_wait",)
            # replicas draw different dropout masks (the rank is mixed into the seed), rank 0 keeps the caller's seed
            assert e.seeds == [rank_dropout_seed(500 + i, rank) for i in range(3)]
            if rank == 0:
                assert e.seeds == [500, 501, 502]
            # parameters stay identical across ranks after the steps
            chk = [torch.zeros(n) for _ in range(world)]
            dist.all_gather(chk, e.flat_train)
            assert torch.equal(chk[0], chk[1])
            finals[overlap] = e.flat_train.clone()
        assert torch.equal(finals[False], finals[True])       # the overlapped optimiser schedule gives the synchronous result
        # bf16 wire format: same mean up to bf16 rounding of the exchanged values
        e = _StubEngine(buckets, n, lo)
        dp = DataParallelStep(e, dist, bf16_wire=True)
        dp.run(torch.full((2, 2), float(rank + 1)), None, None, None, None, True, seed=1, kl_weight=0.5, lr=0.0, max_norm=1e9)
        seen = [r for r in e.log if r[0] == "opt"][-1][2]
        want = (4.0 * 1.5) * (1.0 + torch.arange(n, dtype=torch.float32) / n)
        assert torch.allclose(seen, want, rtol=2e-2), float((seen - want).abs().max())
        # a second engine of the same model (ragged last batch) must not broadcast again
        e2 = _StubEngine(buckets, n, lo)
        e2.flat_train += float(rank + 1)
        DataParallelStep(e2, dist, broadcast=False)
        assert float(e2.flat_train[0]) == float(rank + 1)
        out[rank] = "ok"
    finally:
        dist.destroy_process_group()


def test_data_parallel_step_two_ranks_gloo():
    """DataParallelStep itself (the class bench.py and TrainIQ.fused_training_step drive) over two gloo ranks with a CPU stand-in
    engine: averaged gradient = mean of the single-rank gradients, bucket waits precede their collectives in plan order, phase 1 skips
    the late bucket, overlapped = synchronous optimiser, per-rank dropout seeds, optional bf16 wire format."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: "ok", 1: "ok"}


def test_bucket_plan_of_the_benchmark_configuration():
    """BASELINE configs[2..3] (6-layer d_model 512): the engine closes a gradient bucket at every weight-gradient flush point — groups of
    whole layers of >= 32 MB — so that the exchange is >= 6 collectives in backward-completion order, none above ~40 MB, all but the last
    final before backward ends; phase 1 skips the phase-2-only ones."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bltvqg_amd  # noqa: F401
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.trainer import comm_plan
    eng = StepEngine(make_config(256, 512, 2048, 512, 300, 6, 8, 8000, dtype=1), "cpu")
    buckets = eng.buckets()
    p1, p2 = comm_plan(buckets, False), comm_plan(buckets, True)
    assert len(p1) >= 6 and len(p2) >= 9
    assert all(cnt * 4 <= 41 * 2 ** 20 for _, _, cnt in p2), [cnt * 4 / 2 ** 20 for _, _, cnt in p2]
    assert all(not buckets[ids[0]][2] for ids, _, _ in p1)
    names = list(eng.train_info)
    first = [n for n in names if eng.train_info[n].offset < buckets[0][1]]
    assert first[0] == "decoder.output.weight" and any(n.startswith("decoder.decoder.dec.5.") for n in first)
    # the last collective is the small tail (shared embedding + CNN head) that only the end of backward makes final
    tail = buckets[-1]
    assert tail[1] * 4 < 12 * 2 ** 20
    assert eng.train_info["embedding.0.weight"].offset >= tail[0] and eng.train_info["encoder_cnn.bn.bias"].offset < tail[0] + tail[1]
