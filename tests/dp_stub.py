"""Test seam for `train_iq.py --num_gpus N` on CPU ranks (BLT_TRAINER_FACTORY=dp_stub:factory): a trainer with the surface main() uses
(`fit(loader, max_steps, dist=...)`) that drives the REAL DataParallelStep over gloo with a stand-in for the GPU engine (the product
has no CPU engine).  Every rank writes its final parameters to $BLT_STUB_OUT.rank<r>.pt for the test to compare."""
import os

import torch

from bltvqg_amd.engine import StepEngine, make_config
from bltvqg_amd.trainer import DataParallelStep


class StubEngine(object):
    """The StepEngine surface DataParallelStep uses; gradient = a deterministic function of the rank's batch and the parameters,
    optimiser = clipped SGD."""

    def __init__(self, buckets, n, late_offset, rank):
        self.device = torch.device("cpu")
        self._buckets, self.late_offset = buckets, late_offset
        self.flat_train = torch.full((n,), float(rank))      # ranks start APART: the constructor's broadcast must make them equal
        self.flat_frozen = torch.zeros(8)
        self.flat_grad = torch.zeros(n)
        self.batches = []

    def buckets(self):
        return self._buckets

    def bucket_wait(self, i, stream):
        assert stream is None

    def forward(self, images, context, posterior, target, eps, phase2, seed):
        self._x, self._phase2 = float(images.double().mean()) + float(target.double().mean()) * 1e-3, bool(phase2)
        self.batches.append(self._x)

    def loss_backward(self, kl_weight):
        n = self.flat_grad.numel()
        self.flat_grad.zero_()
        hi = n if self._phase2 else self.late_offset
        self.flat_grad[:hi] = self._x * (1.0 + torch.arange(hi, dtype=torch.float32) / n) + 0.1 * self.flat_train[:hi]

    def optimizer_step(self, lr, max_norm, overlap=False):
        g = self.flat_grad
        self.flat_train -= 0.05 * min(1.0, max_norm / (float(g.norm()) + 1e-6)) * g

    def optimizer_wait(self):
        pass


class StubTrainer(object):
    def __init__(self, vocab, args):
        self.args = args
        cfg = make_config(4, 64, 128, 64, 20, 1, 4, 97, image_hw=(64, 64), dtype=0)
        real = StepEngine(cfg, "cpu")                        # the real engine's bucket layout (host-side descriptor only)
        self._layout = (real.buckets(), real.train_size, real.late_offset)

    def fit(self, loader, max_steps, log_every=0, dist=None):
        rank = dist.get_rank() if dist is not None else 0
        eng = StubEngine(*self._layout, rank)
        dp = DataParallelStep(eng, dist, overlap_optimizer=True)
        it = iter(loader)
        for step in range(int(max_steps)):
            b = next(it)
            phase2 = step >= int(self.args.num_pretraining_steps)
            dp.run(b["images"], b["answers"], b["posteriors"], b["questions"], None, phase2, seed=step, kl_weight=0.5, lr=0.1, max_norm=5.0)
        dp.finish()
        out = os.environ.get("BLT_STUB_OUT")
        if out:
            torch.save({"params": eng.flat_train, "batches": eng.batches, "world": dist.get_world_size() if dist is not None else 1},
                       "%s.rank%d.pt" % (out, rank))


def factory(vocab, args):
    return StubTrainer(vocab, args)
