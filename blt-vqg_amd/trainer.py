"""Host-side step driver: reference-style initialisation, the fused train step and its data-parallel form.

Data parallelism mirrors what `pl.Trainer(gpus=N)` gives the reference (train_iq.py:372-373: Lightning DDP = one process per
GPU, gradient mean across ranks, BatchNorm statistics per replica), re-designed for xGMI: the engine writes gradients into ONE
flat fp32 buffer laid out in backward-completion order and closes a bucket (a group of whole layers, >= ~32 MB) at every
weight-gradient flush point, so the exchange is one all-reduce per bucket issued on a side stream as soon as the bucket's event
fires, overlapping the rest of backward; only the last (embedding + CNN head, ~10 MB) is final with the end of backward.  The
optimiser step waits for the side stream.
"""
import math

import torch


def noam_lr(step, hidden_dim, warmup_steps=4000):
    """TrainIQ.custom_optimizer (reference train_iq.py:252-257); 0 at step 0."""
    return math.sqrt(1.0 / hidden_dim) * min(math.sqrt(1.0 / (step + 1)), step * warmup_steps ** -1.5)


def kl_weight(kliter, full_kl_step):
    """reference train_iq.py:96-97."""
    return min(math.tanh(6.0 * kliter / full_kl_step - 3.0) + 1.0, 1.0)


def init_reference_style(eng, seed=0, resnet_state=None):
    """Fills the engine's flat buffers with the reference's initialisers (distributions, not its RNG stream):
    nn.Linear default U(+-1/sqrt(fan_in)) (iq.py:76, transformer_layers.py:453-456, ...), embedding randn*0.01 (iq.py:58),
    cnn.fc N(0, 0.02) / bias 0 (encoder_cnn.py:24-28), MLP He-normal / bias 0 (mlp.py:37-38), LayerNorm/BatchNorm 1/0.
    The backbone is `resnet18(pretrained=True)` in the reference (encoder_cnn.py:17); those weights cannot be fetched here, so
    it is He-initialised unless `resnet_state` (a torchvision-format state dict) is given."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    dev = eng.device
    with torch.no_grad():
        for name, info in eng.train_info.items():
            v = eng.view(name, 0)
            shape = info.shape
            if name == "embedding.0.weight":
                t = torch.randn(shape, generator=g) * 0.01
            elif name == "encoder_cnn.cnn.fc.weight":
                t = torch.randn(shape, generator=g) * 0.02
            elif name.startswith("image_reconstructor") and name.endswith("weight"):
                t = torch.randn(shape, generator=g) * math.sqrt(2.0 / shape[1])
            elif name in ("encoder_cnn.cnn.fc.bias",) or (name.startswith("image_reconstructor") and name.endswith("bias")):
                t = torch.zeros(shape)
            elif "layer_norm" in name or name.startswith("encoder_cnn.bn."):
                t = torch.ones(shape) if name.endswith("weight") else torch.zeros(shape)
            elif len(shape) == 2:
                bound = 1.0 / math.sqrt(shape[1])
                t = (torch.rand(shape, generator=g) * 2 - 1) * bound
            else:   # bias of an nn.Linear: U(+-1/sqrt(fan_in)) with the fan_in of its weight
                wname = name[:-4] + "weight"
                fan_in = eng.train_info[wname].shape[1]
                t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
            v.copy_(t.to(dev))
        for name, info in eng.frozen_info.items():
            v = eng.view(name, 1)
            shape = info.shape
            key = name[len("encoder_cnn.cnn."):] if name.startswith("encoder_cnn.cnn.") else None
            if resnet_state is not None and key in resnet_state:
                t = resnet_state[key].float()
            elif len(shape) == 4:
                fan_out = shape[0] * shape[2] * shape[3]
                t = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_out)
            elif name.endswith("running_var") or name.endswith(".weight"):
                t = torch.ones(shape)
            else:
                t = torch.zeros(shape)
            v.copy_(t.to(dev))
    eng.lib.bltvqg_engine_invalidate_frozen(eng.h)


def active_buckets(buckets, phase2):
    """Gradient buckets that carry gradients in this phase: the phase-2-only buckets are skipped before the switch."""
    return [(i, off, n) for i, (off, n, late) in enumerate(buckets) if phase2 or not late]


def comm_plan(buckets, phase2):
    """The collectives of one step as (bucket ids to wait for, offset, count), in the order backward completes the buckets (the order the
    engine lists them in).  One collective per bucket: the engine closes a bucket at every weight-gradient flush point — groups of whole
    layers of >= ~32 MB — and records its event right behind the flush, so every collective but the last (embedding + CNN head, final
    with the end of backward) is enqueued while backward is still running and none exceeds a few layers (25-32 MB buckets
    launched as they become final).  Every bucket is its own collective, also when several become final at one flush (no in-stack
    flushes: all of a stack's buckets at its end) — what a collective costs on the exposed tail has not been measured with more than
    one rank (no multi-GPU box was available to any round), so nothing is merged on a guess."""
    plan = []
    for i, off, n in active_buckets(buckets, phase2):
        plan.append(([i], off, n))
    return plan


def allreduce_bucket(dist, flat_grad, off, n, wire=None):
    """Mean of one contiguous gradient bucket across ranks, in place (RCCL has AVG; gloo — used by the CPU tests — only SUM).
    `wire`: optional bf16 staging buffer of at least n elements — the gradients then cross xGMI as bf16 (half the bytes: 164 instead
    of 329 MB per step on the big configuration), are summed in bf16 by the collective and widened back into the fp32 buffer; the
    default (None) keeps the exchange in fp32 like Lightning DDP."""
    view = flat_grad[off:off + n]
    if wire is not None:
        w = wire[:n]
        w.copy_(view)
        if dist.get_backend() == "nccl":
            dist.all_reduce(w, op=dist.ReduceOp.AVG)
            view.copy_(w)
        else:
            dist.all_reduce(w, op=dist.ReduceOp.SUM)
            view.copy_(w)
            view.div_(dist.get_world_size())
        return
    if dist.get_backend() == "nccl":
        dist.all_reduce(view, op=dist.ReduceOp.AVG)
    else:
        dist.all_reduce(view, op=dist.ReduceOp.SUM)
        view.div_(dist.get_world_size())


def shard_seed(base_seed, rank):
    """Each rank draws its own shard of the global batch (weak scaling: per-GPU batch fixed)."""
    return int(base_seed) + int(rank)


def rank_dropout_seed(seed, rank):
    """Dropout seed of one rank: DDP replicas draw independent dropout masks (each process has its own RNG stream), so the rank is
    mixed into the step seed — rank 0 keeps the caller's seed, which keeps single-GPU runs and their tests unchanged."""
    return (int(seed) + 0x9E3779B97F4A7C15 * int(rank)) & 0xFFFFFFFFFFFFFFFF


class DataParallelStep(object):
    """forward -> fused losses + backward -> (overlapped gradient all-reduce) -> clip + Adam.

    `engine` is a StepEngine (or anything with its surface: device, flat_train / flat_grad / flat_frozen, buckets(), bucket_wait(),
    forward(), loss_backward(), optimizer_step(), optimizer_wait() — tests/test_dp_gloo.py drives this class with a CPU stand-in over
    gloo).  On a CPU device there are no streams: the collectives run inline."""

    def __init__(self, engine, dist=None, overlap_optimizer=False, broadcast=True, bf16_wire=False, check_ids_every=0, comm_stream="shared"):
        self.e = engine
        self.dist = dist
        self.rank = dist.get_rank() if dist is not None else 0
        self.cuda = torch.device(engine.device).type == "cuda"
        # clip + Adam (and, behind it, the tail of the gradient all-reduce) overlapped with the next step's frozen CNN
        # (StepEngine.optimizer_step(overlap=True)); call finish() before reading parameters
        self.overlap_optimizer = overlap_optimizer
        self.comm = torch.cuda.Stream(device=engine.device) if (dist is not None and self.cuda) else None
        # comm_stream = "shared" (default): the communication stream is idle from the start of a step until the first gradient bucket is
        # final (~60 % of the step) — exactly where the NEXT batch's frozen conv stack runs when it is enqueued one batch ahead
        # (run(next_images=...)).  One stream serves both, in order: conv stack of batch i+1, then the all-reduces of step i, then the
        # optimiser fork; the step stays at four live streams (the command processor runs four queues side by side).
        # comm_stream = "own": the collectives get a stream to themselves and the conv look-ahead a second one (five live streams + RCCL's
        # own): an in-order shared stream makes collective 1 of step i queue behind the whole conv stack of batch i+1, which the one-rank
        # rehearsal cannot price — with real RCCL kernels on N > 1 GPUs one sweep of this switch (bench.py --comm-stream) decides it.
        if comm_stream not in ("shared", "own"):
            raise ValueError("comm_stream must be 'shared' or 'own'")
        self.comm_stream_mode = comm_stream if self.comm is not None else "none"
        self.conv_stream = None
        if self.comm is not None and hasattr(engine, "adopt_conv_stream") and engine.cfg.num_regions == 0:
            if comm_stream == "shared":
                engine.adopt_conv_stream(self.comm)
                self.conv_stream = self.comm
            else:
                self.conv_stream = torch.cuda.Stream(device=engine.device)
                engine.adopt_conv_stream(self.conv_stream)
        self.buckets = engine.buckets()
        # with an exchange to overlap, the engine flushes weight gradients at every bucket boundary so that each bucket's all-reduce can
        # start under the rest of backward; on one GPU those extra launches only compete with the chain (+0.15 ms per step measured)
        if dist is not None and hasattr(engine, "set_bucket_flush"):
            engine.set_bucket_flush(True)
        # optional bf16 wire format of the gradient exchange (allreduce_bucket): one staging buffer as large as the largest collective
        self.wire = None
        if dist is not None and bf16_wire:
            self.wire = torch.empty(max(n for _, _, n in comm_plan(self.buckets, True)), dtype=torch.bfloat16, device=engine.flat_grad.device)
        self.steps_run = 0
        self._prefetched = None       # the tensor whose conv stack the engine holds (identity, not address: the allocator reuses addresses)
        # exposed communication: with measure_exposed on, every step records an event pair (end of backward on the step's stream, end of the
        # last all-reduce on the communication stream); exposed_ms() = how long the optimiser had to wait for the exchange after backward
        self.measure_exposed = False
        self._exposed = []
        self._bucket_marks = []
        self.check_ids_every = int(check_ids_every)      # every n-th step: raise if the batch held token ids outside the vocabulary
        if dist is not None and broadcast:
            # one-time parameter broadcast from rank 0 (DDP does the same at construction); engines of other batch shapes share these
            # buffers and must not repeat it
            dist.broadcast(engine.flat_train, 0)
            dist.broadcast(engine.flat_frozen, 0)
            if hasattr(engine, "params_changed"):
                engine.params_changed()

    def _on_comm(self):
        import contextlib
        return torch.cuda.stream(self.comm) if self.comm is not None else contextlib.nullcontext()

    def reduce_gradients(self, phase2):
        e, dist = self.e, self.dist
        marks = [] if (self.measure_exposed and self.comm is not None) else None
        for ids, off, n in comm_plan(self.buckets, phase2):
            for i in ids:
                e.bucket_wait(i, self.comm)      # the side stream waits for the engine's "bucket i is final" event(s)
            if marks is not None:                # when the communication stream got past the wait = when this collective could start
                ev = torch.cuda.Event(enable_timing=True)
                ev.record(self.comm)
                marks.append((ids[0], ev))
            with self._on_comm():
                allreduce_bucket(dist, e.flat_grad, off, n, self.wire)
        if marks is not None:
            self._bucket_marks.append(marks)
        if not self.overlap_optimizer and self.comm is not None:
            torch.cuda.current_stream(e.device).wait_stream(self.comm)

    def run(self, images, context, posterior, target, eps, phase2, seed, kl_weight, lr, max_norm=5.0, next_images=None):
        """One training step.  `next_images` (image mode, optional): the NEXT step's image batch — its frozen conv stack is enqueued on the
        engine's conv stream now (StepEngine.prefetch_images; models/encoder_cnn.py:18-19 freezes the backbone, so it depends on nothing
        this step updates); the next run() then starts at the trainable head and must be given that same tensor (or images=None)."""
        e = self.e
        # this driver updates the parameters only through the engine's own optimiser, so the update may keep the bf16 weight shadows
        # current; the promise is (re)made per step, and the autograd path (models.IQ.forward) withdraws it with params_touched()
        if hasattr(e, "trust_shadows"):
            e.trust_shadows(True)
        pending = e.prefetch_pending() if hasattr(e, "prefetch_pending") else 0
        if pending and images is not None and images is not self._prefetched:
            raise RuntimeError("DataParallelStep.run: the engine holds the prefetched conv stack of another image batch than the one passed "
                               "(pass the tensor that was given as next_images to the previous run, or images=None)")
        if next_images is not None:
            if pending == 0 and images is not None:      # cold start: this step's own stack goes through the same path
                e.prefetch_images(images)
                pending = 1
            e.prefetch_images(next_images)
            self._prefetched = next_images
        e.forward(None if pending else images, context, posterior, target, eps, phase2, rank_dropout_seed(seed, self.rank))
        e.loss_backward(kl_weight)
        ev_a = None
        if self.measure_exposed and self.comm is not None:      # end of backward on the step's stream (every engine stream is joined)
            ev_a = torch.cuda.Event(enable_timing=True)
            ev_a.record()
        if self.dist is not None:
            self.reduce_gradients(phase2)
        if ev_a is not None:                                     # ... against the end of the last collective on the communication stream
            ev_b = torch.cuda.Event(enable_timing=True)
            ev_b.record(self.comm)
            self._exposed.append((ev_a, ev_b))
        if self.dist is not None and self.overlap_optimizer:
            # the update is forked from the COMMUNICATION stream (behind the all-reduces); the main stream never waits for them, the
            # next forward's parameter consumers wait for the update
            with self._on_comm():
                e.optimizer_step(lr, max_norm, overlap=True)
        else:
            e.optimizer_step(lr, max_norm, overlap=self.overlap_optimizer)
        # token ids outside the vocabulary are counted on the device (they were treated as <pad>); reading the counter is a host sync, so
        # the fused loop looks at it on a cadence instead of every step (the reference's embedding lookup raises at once)
        self.steps_run += 1
        if self.check_ids_every and self.steps_run % self.check_ids_every == 0 and hasattr(e, "stats"):
            e.stats(check_ids=True)

    def exposed_ms(self):
        """Per measured step: milliseconds between the end of backward and the end of the last gradient all-reduce (0 when the exchange
        finished first); synchronises.  Clears the record."""
        out = []
        for a, b in self._exposed:
            b.synchronize()
            out.append(max(0.0, a.elapsed_time(b)))
        self._exposed = []
        return out

    def bucket_start_ms(self):
        """Per measured step: for every collective, milliseconds from the moment it could start (the communication stream got past its
        bucket's event) to the END of backward — positive = the collective was free to run that long underneath backward.  Pairs with
        exposed_ms(); call it first (exposed_ms clears the record)."""
        out = []
        for (a, _), marks in zip(self._exposed, self._bucket_marks):
            a.synchronize()
            out.append([(i, round(ev.elapsed_time(a), 3)) for i, ev in marks])
        self._bucket_marks = []
        return out

    def finish(self):
        self.e.optimizer_wait()
