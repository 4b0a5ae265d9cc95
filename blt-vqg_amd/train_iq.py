"""`train_iq.TrainIQ` drop-in (reference train_iq.py:28-261): same constructor, `forward(batch)`, `calculate_losses` 7-tuple,
`training_step`, `custom_optimizer` (Noam), `configure_optimizers` (Adam), attributes `iter`, `kliter`, `latent_transformer`.

pytorch_lightning is not a dependency: when it is importable TrainIQ subclasses pl.LightningModule (so pl.Trainer can drive
`training_step`), otherwise torch.nn.Module, and `fit()` below is a minimal trainer loop.  Two ways to run a step:
  * `training_step(batch, idx)`  — the reference contract: returns the loss tensor; autograd backward runs the HIP backward
    through `IQ.forward`'s autograd node; clip / optimizer are the caller's (Lightning's) job;
  * `fused_training_step(batch)` — forward + losses + backward + clip 5 + Adam entirely inside the HIP engine (what bench.py
    measures); no autograd graph, no per-step host synchronisation.
"""
import argparse
import math
import os
import sys
from types import SimpleNamespace

import torch
from torch import nn

from .iq import IQ
from .trainer import DataParallelStep, kl_weight, noam_lr

try:  # pragma: no cover - not installed in this image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    pl = None
    _Base = nn.Module


class TrainIQ(_Base):
    def __init__(self, vocab, args):
        super().__init__()
        self.latent_transformer = False
        self.vocab = vocab
        self.args = args
        self.hp_string = "{}_{}_{}_{}_{}_{}_{}_{}_{}_{}. {}".format(
            getattr(args, "input_mode", "ans"), args.emb_dim, "True", args.hidden_dim, args.latent_dim, args.pwffn_dim, args.num_layers,
            args.num_heads, getattr(args, "lr", 3e-5), getattr(args, "batch_size", 128), getattr(args, "print_note", ""))
        self.iter = 0
        self.kliter = 0
        self.logged = {}
        self.model = IQ(self.latent_transformer, vocab, args)
        pad = vocab.word2idx[vocab.SYM_PAD] if hasattr(vocab, "SYM_PAD") else 0
        self.criterion = nn.CrossEntropyLoss(ignore_index=pad)
        self.image_recon_criterion = nn.MSELoss()
        self._optimizer = None
        self._dp = None
        self.val_metrics = {k: [] for k in ("loss", "img", "ppl", "kld", "aux", "elbo", "rec")}      # train_iq.py:45-51

    # ---- reference surface ------------------------------------------------------------------------------------------
    def _device(self):
        d = getattr(self.args, "device", "cuda")
        return torch.device(d) if not isinstance(d, torch.device) else d

    def _unpack(self, batch):
        # dict order = reference collate_fn (utils/data_loader.py:175); the reference hard-codes .cuda() (train_iq.py:69,114),
        # routed through args.device here
        dev = self._device()
        images = batch["images"].to(dev) if batch["images"] is not None else None      # None: already in the engine's stem input
        questions, posteriors = batch["questions"].to(dev), batch["posteriors"].to(dev)
        mode = getattr(self.args, "input_mode", "ans")
        context = batch["answers"].to(dev) if mode == "ans" else batch["answer_types_for_input"].to(dev)
        return images, context, posteriors, questions

    def forward(self, batch):
        images, context, posteriors, questions = self._unpack(batch)
        eps = batch["eps"].to(images.device) if (isinstance(batch, dict) and "eps" in batch and self.latent_transformer) else None
        output, z, kld_loss, image_recon = self.model(images, context, posteriors, questions, eps=eps)
        return output, z, kld_loss, image_recon

    def calculate_losses(self, output, image_recon, kld_loss, z_logit, target):
        """reference train_iq.py:81-103 (same 7-tuple; the five .item() host syncs are the reference's)."""
        loss_rec = self.criterion(output.reshape(-1, output.size(-1)), target.reshape(-1))
        loss_img = self.image_recon_criterion(image_recon[0], image_recon[1])
        if not self.latent_transformer:
            kld_loss = torch.tensor([0])
            loss = loss_rec + self.args.image_recon_lambda * loss_img
            elbo = loss_rec
            aux = 0
        else:
            z_rep = z_logit.unsqueeze(1).expand(-1, output.size(1), -1)
            loss_aux = self.criterion(z_rep.reshape(-1, z_rep.size(-1)), target.reshape(-1))
            w = kl_weight(self.kliter, self.args.full_kl_step)
            aux = loss_aux.item()
            elbo = loss_rec + kld_loss
            loss = loss_rec + self.args.kl_ceiling * w * kld_loss + self.args.aux_ceiling * loss_aux + self.args.image_recon_lambda * loss_img
        return loss, loss_rec.item(), loss_img.item(), math.exp(min(loss_rec.item(), 100)), kld_loss.item(), aux, elbo.item()

    def _phase_switch(self):
        if self.iter == self.args.num_pretraining_steps:          # train_iq.py:108-111
            self.latent_transformer = True
            self.model.switch_GVT_train_mode(True)

    def log(self, name, value, *a, **k):
        if pl is not None:  # pragma: no cover
            return super().log(name, value, *a, **k)
        self.logged[name] = value

    def training_step(self, batch, batch_idx=0):
        self._phase_switch()
        output, z_logit, kld_loss, image_recon = self(batch)
        target = batch["questions"].to(output.device)
        loss, loss_rec, loss_img, ppl, kld, aux, elbo = self.calculate_losses(output, image_recon, kld_loss, z_logit, target)
        if self.latent_transformer:
            self.kliter += 1
        for k, v in (("train loss", loss), ("train rec loss", loss_rec), ("image recon loss", loss_img), ("perplexity", ppl),
                     ("kld loss", kld), ("aux loss", aux), ("elbo", elbo)):
            self.log(k, v)
        self.custom_optimizer(self.iter)
        self.iter += 1
        return loss

    def validation_step(self, batch, batch_idx=0):
        """reference train_iq.py:133-157 (Lightning runs it under model.eval() and torch.no_grad(): no dropout, BatchNorm from the
        running statistics, nothing updated)."""
        with torch.no_grad():
            output, z_logit, kld_loss, image_recon = self(batch)
            target = batch["questions"].to(output.device)
            loss, loss_rec, loss_img, ppl, kld, aux, elbo = self.calculate_losses(output, image_recon, kld_loss, z_logit, target)
        for k, v in (("loss", loss.item()), ("img", self.args.image_recon_lambda * loss_img), ("ppl", ppl), ("kld", kld), ("aux", aux),
                     ("elbo", elbo), ("rec", loss_rec)):
            self.val_metrics[k].append(v)
        for k, v in (("val_loss", loss.item()), ("val_loss_rec", loss_rec), ("val_img_loss", loss_img), ("val_ppl", ppl),
                     ("val_kld_loss", kld), ("val_aux", aux), ("val_elbo", elbo)):
            self.log(k, v)
        return batch

    def custom_optimizer(self, step, warmup_steps=4000):
        """Noam schedule written into the optimizer (reference train_iq.py:252-257)."""
        lr = noam_lr(step, self.args.hidden_dim, warmup_steps)
        self.current_lr = lr
        opt = None
        if pl is not None and getattr(self, "trainer", None) is not None:  # pragma: no cover
            opt = self.trainer.lightning_optimizers[0]
        elif self._optimizer is not None:
            opt = self._optimizer
        if opt is not None:
            opt.param_groups[0]["lr"] = lr
        return lr

    def configure_optimizers(self):
        self._optimizer = torch.optim.Adam(self.parameters(), lr=getattr(self.args, "lr", 3e-5))
        return self._optimizer

    # ---- fused path ----------------------------------------------------------------------------------------------------
    def _images_on_device(self, batch):
        """The batch's image tensor as the engine takes it (device, fp32, contiguous) — cached per batch object, so that the tensor handed
        to the conv look-ahead as `next_batch` is the very one the next step passes."""
        src = batch["images"]
        # keyed on the SOURCE tensor and its version counter, not only on the dict: a loader that refills one dict / pinned buffer in
        # place must not get the previous batch's device copy back
        key = (id(batch), id(src), src.data_ptr(), int(getattr(src, "_version", 0)))
        cache = getattr(self, "_img_cache", None)
        if cache is None:
            cache = self._img_cache = {}
        hit = cache.get(key)
        if hit is not None and hit[0] is src:
            return hit[1]
        t = src.to(self._device()).contiguous().float()
        if len(cache) >= 2:      # the current batch and the look-ahead one
            cache.pop(next(iter(cache)))
        cache[key] = (src, t)
        return t

    def fused_training_step(self, batch, dist=None, next_batch=None):
        """One full reference training step inside the HIP engine.  Returns nothing; `last_stats()` syncs and reads the losses.
        next_batch (optional, image mode): the batch of the NEXT call — its frozen ResNet-18 forward is enqueued one batch ahead
        (models/encoder_cnn.py:18-19 freezes the backbone; DataParallelStep.run(next_images=...))."""
        self._phase_switch()
        images, context, posteriors, questions = self._unpack(batch)
        if images is not None:
            images = self._images_on_device(batch)
        if images is None:      # DeviceBatchProducer.batch(engine=...) wrote the images into this engine's packed stem input
            eng = batch["engine"]
        else:
            eng = self.model.engine(images, context, posteriors, questions)
        if self._dp is None or self._dp.e is not eng:
            # one driver per engine (= per batch shape: the ragged last batch of an epoch has its own); they share the parameter,
            # gradient and optimiser buffers, so only the first one broadcasts them
            if not hasattr(self, "_dps"):
                self._dps = {}
            if id(eng) not in self._dps:
                # padded widths: the module's parameters go into the engine's (padded) buffers BEFORE the rank-0 broadcast of the
                # constructor, and the engine owns them from then on — a scatter after the broadcast would put every rank's local
                # values back over rank 0's
                self.model.sync_to_engine(for_fused=True)
                self._dps[id(eng)] = DataParallelStep(eng, dist, broadcast=not self._dps, check_ids_every=100)
            self._dp = self._dps[id(eng)]
        if getattr(self, "_pending_adam", None) is not None:      # optimiser state of a loaded checkpoint (fused path)
            ad, self._pending_adam = self._pending_adam, None
            eng.adam_m.copy_(ad["m"].to(eng.adam_m.device))
            eng.adam_v.copy_(ad["v"].to(eng.adam_v.device))
            eng.set_adam_steps(*ad["steps"])
        phase2 = self.latent_transformer
        eps = None
        if phase2:
            eps = batch["eps"].to(questions.device) if "eps" in batch else torch.randn(questions.shape[0], self.args.latent_dim,
                                                                                       device=questions.device)
        # padded widths (models.IQ / padded.py): the engine's optimiser owns the (padded) parameter buffers from here on; the latent noise
        # has the real width, the pad columns get none
        self.model.sync_to_engine(for_fused=True)
        if eps is not None and eps.shape[1] != eng.cfg.latent_dim:
            eps = torch.nn.functional.pad(eps.float(), (0, eng.cfg.latent_dim - eps.shape[1]))
        w = kl_weight(self.kliter, self.args.full_kl_step) if phase2 else 0.0
        self.model._step_seed += 1
        nxt = None
        look_ahead = (images is not None and images.dim() == 4 and not getattr(self.args, "no_prefetch", False))
        if look_ahead and next_batch is not None and next_batch.get("images") is not None and \
                tuple(next_batch["images"].shape) == tuple(images.shape):
            nxt = self._images_on_device(next_batch)
        elif eng.prefetch_pending() and images is not None and self._dp._prefetched is not images:
            raise RuntimeError("fused_training_step: the engine holds the look-ahead conv stack of another batch than the one passed "
                               "(pass the batch that was given as next_batch to the previous call)")
        self._dp.run(images, context.contiguous(), posteriors.contiguous(),
                     questions.contiguous(), eps, phase2,
                     self.model._base_seed + self.model._step_seed, w, noam_lr(self.iter, self.args.hidden_dim), 5.0, next_images=nxt)
        self._last_engine, self._last_w = eng, w
        if phase2:
            self.kliter += 1
        self.iter += 1

    def last_stats(self):
        st = self._last_engine.stats()
        st["loss"] = st["rec"] + self.args.image_recon_lambda * st["img"] + (
            self.args.kl_ceiling * self._last_w * st["kld"] + self.args.aux_ceiling * st["aux"] if self.latent_transformer else 0.0)
        st["ppl"] = math.exp(min(st["rec"], 100))
        return st

    # ---- checkpoints -----------------------------------------------------------------------------------
    # The reference saves through Lightning (train_iq.py:277-309, trainer.save_checkpoint): a torch-pickled dict whose "state_dict"
    # holds the LightningModule's tensors under "model.<IQ key>" (the 260 keys of tests/golden/state_keys_small.txt, aliases
    # included) next to "epoch" / "global_step" / optimizer and scheduler state.  save_checkpoint writes that layout (tensors and
    # plain Python scalars only); load_checkpoint reads it — from this class or from the reference — with the SAFE loader only
    # (torch.load(weights_only=True)): a file that needs arbitrary unpickling (e.g. an argparse.Namespace under "hyper_parameters") is
    # refused by torch with a message naming the offending global, and is not loaded any other way.
    def save_checkpoint(self, path):
        eng0 = getattr(self, "_last_engine", None)
        if eng0 is not None:
            eng0.optimizer_wait()
            eng0.conv_stream_wait()
        sd = {"model." + k: v.detach().float().cpu().clone() for k, v in self.model.state_dict().items()}
        ckpt = {"epoch": 0, "global_step": int(self.iter), "pytorch-lightning_version": "1.1.8", "state_dict": sd,
                "blt_vqg": {"iter": int(self.iter), "kliter": int(self.kliter), "latent_transformer": bool(self.latent_transformer)}}
        eng = getattr(self, "_last_engine", None)
        if eng is not None:      # fused path: Adam moments of the flat buffer (the autograd path keeps them in its torch optimizer)
            eng.optimizer_wait()
            # a conv stack that runs one batch ahead advances the BatchNorm2d running statistics on the conv stream: the statistics
            # saved here INCLUDE that look-ahead batch (whole, not torn) — on resume that batch's stack runs again
            eng.conv_stream_wait()
            ckpt["blt_vqg"]["adam"] = {"m": eng.adam_m.detach().cpu().clone(), "v": eng.adam_v.detach().cpu().clone(),
                                       "steps": [int(x) for x in eng.adam_steps()]}
        torch.save(ckpt, path)

    def load_checkpoint(self, path, strict=True):
        ckpt = torch.load(path, map_location="cpu", weights_only=True)
        sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
        sd = {(k[len("model."):] if k.startswith("model.") else k): v for k, v in sd.items()
              if not k.startswith("criterion") and not k.startswith("image_recon_criterion")}
        res = self.model.load_state_dict(sd, strict=strict)
        meta = ckpt.get("blt_vqg", {}) if isinstance(ckpt, dict) else {}
        self.iter = int(meta.get("iter", ckpt.get("global_step", 0) if isinstance(ckpt, dict) else 0))
        self.kliter = int(meta.get("kliter", max(0, self.iter - getattr(self.args, "num_pretraining_steps", 0))))
        self.latent_transformer = bool(meta.get("latent_transformer", self.iter >= getattr(self.args, "num_pretraining_steps", 1 << 62)))
        self.model.switch_GVT_train_mode(self.latent_transformer)
        self._pending_adam = meta.get("adam")
        return res

    def fit(self, loader, max_steps, log_every=100, dist=None):
        """Minimal stand-in for pl.Trainer(max_steps=..., gradient_clip_val=5).fit (reference train_iq.py:372-374)."""
        step = 0

        def batches():
            while True:
                n = 0
                for b in loader:
                    n += 1
                    yield b
                if n == 0:
                    return
        it = batches()
        cur = next(it, None)
        while cur is not None and step < max_steps:
            nxt = next(it, None) if step + 1 < max_steps else None      # one batch of look-ahead: its conv stack runs underneath this step
            self.fused_training_step(cur, dist, next_batch=nxt)
            step += 1
            if log_every and step % log_every == 0:
                print("step %d %s" % (step, {k: round(v, 4) for k, v in self.last_stats().items()}), flush=True)
            cur = nxt


def _fit_from_producer(self, producer, batch_size, max_steps, shuffle=True, log_every=100, dist=None):
    """Training straight from a store that lives in HBM (blt-vqg_amd/batch.py): per step the producer's two gather kernels write the
    token rows and the transformed images (into the engine's packed stem input), then the fused step runs — no host pixels at all."""
    sa = (producer.a_len + 1) if getattr(self.args, "input_mode", "ans") == "ans" else 3
    step = 0
    while step < max_steps:
        for idx in producer.epoch(batch_size, shuffle=shuffle, drop_last=True):
            eng = self.model.engine_for_shape(batch_size, sa, producer.q_len + 1, producer.q_len, producer.out_size, producer.out_size,
                                              producer.device)
            b = producer.batch(idx, engine=eng)
            b["engine"] = eng
            self.fused_training_step(b, dist)
            step += 1
            if log_every and step % log_every == 0:
                print("step %d %s" % (step, {k: round(v, 4) for k, v in self.last_stats().items()}), flush=True)
            if step >= max_steps:
                break


TrainIQ.fit_from_producer = _fit_from_producer


def build_parser():
    """CLI flags and defaults of the reference (train_iq.py:313-351) + precision."""
    p = argparse.ArgumentParser()
    p.add_argument("--emb_dim", type=int, default=300)
    # hidden_dim / latent_dim / pwffn_dim: reference defaults; the MI355X engine needs multiples of 8 (IQ.__init__ says so): pass
    # e.g. 304 / 304 / 608 or the BASELINE widths 256 / 512
    p.add_argument("--hidden_dim", type=int, default=300)
    p.add_argument("--latent_dim", type=int, default=300)
    p.add_argument("--pwffn_dim", type=int, default=600)
    p.add_argument("--num_layers", type=int, default=4)
    p.add_argument("--num_heads", type=int, default=4)
    p.add_argument("--lr", type=float, default=3e-5)
    p.add_argument("--num_pretraining_steps", type=float, default=12000)
    p.add_argument("--total_training_steps", type=int, default=35000)
    p.add_argument("--full_kl_step", type=int, default=15000)
    p.add_argument("--kl_ceiling", type=float, default=0.5)
    p.add_argument("--aux_ceiling", type=float, default=1.0)
    p.add_argument("--image_recon_lambda", type=float, default=0.1)
    p.add_argument("--batch_size", type=int, default=128)
    p.add_argument("--emb_file", type=str, default="vectors/glove.6B.300d.txt")
    p.add_argument("--dataset", type=str, default="data/processed/iq_dataset.hdf5")
    p.add_argument("--val_dataset", type=str, default="data/processed/iq_val_dataset.hdf5")
    p.add_argument("--vocab", type=str, default="vocab.pkl")
    p.add_argument("--use_gpu", type=bool, default=True)
    p.add_argument("--num_gpus", type=int, default=1)
    p.add_argument("--print_note", type=str, default="")
    p.add_argument("--input_mode", type=str, default="ans")
    p.add_argument("--precision", type=str, default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--synthetic", action="store_true", help="train on seeded synthetic batches (no dataset in this environment)")
    return p


class SyntheticVocabulary(object):
    """Reserved ids of utils/train_utils.py:18-37 + `n` synthetic words."""
    SYM_PAD, SYM_SOQ, SYM_SOR, SYM_EOS, SYM_UNK, SYM_POS = "<pad>", "<start>", "<resp>", "<end>", "<unk>", "<pos>"

    def __init__(self, size=8000):
        self.word2idx, self.idx2word = {}, {}
        for w in (self.SYM_PAD, self.SYM_SOQ, self.SYM_SOR, self.SYM_EOS, self.SYM_UNK, self.SYM_POS):
            self.add_word(w)
        i = 0
        while len(self.word2idx) < size:
            self.add_word("w%d" % i)
            i += 1

    def add_word(self, w):
        if w not in self.word2idx:
            self.idx2word[len(self.word2idx)] = w
            self.word2idx[w] = len(self.word2idx)

    def __len__(self):
        return len(self.word2idx)


def _init_distributed(args):
    """Inside a rank (torch.distributed.run exported RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*): pick this rank's GPU and join the process
    group.  backend `nccl` IS RCCL on ROCm; BLT_DIST_BACKEND=gloo runs the same step with gloo carrying the gradients (rehearsals on a
    box with fewer GPUs than ranks: BLT_SHARE_GPU=1 maps every rank to LOCAL_RANK modulo the device count)."""
    from .launch import rank_env
    import torch.distributed as dist
    rank, world, local = rank_env()
    backend = os.environ.get("BLT_DIST_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if torch.cuda.is_available() and args.use_gpu:
        if os.environ.get("BLT_SHARE_GPU") == "1":
            local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
        args.device = torch.device("cuda", local)
    else:
        args.device = torch.device("cpu")
    if backend == "nccl" and args.device.type == "cuda":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=args.device)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist, rank, world


def _load_factory(spec):
    """`module:callable` -> the callable (the trainer factory seam of main(): tests drive the launch + data-parallel path on CPU ranks
    with a stand-in for the GPU engine; the product path has no CPU fallback)."""
    import importlib
    mod, _, fn = spec.partition(":")
    return getattr(importlib.import_module(mod), fn)


def main(argv=None):
    """`python train_iq.py --synthetic [--num_gpus N] ...` (reference train_iq.py:313-374).  --num_gpus N > 1 is the reference's
    `pl.Trainer(gpus=N)`: started plainly, this process spawns N ranks (one per GPU, `python -m torch.distributed.run`) BEFORE any GPU
    call and relays their exit code; started by a launcher it IS one rank: it joins the process group, trains on its shard of every
    global batch (seed = base + step * world + rank) and exchanges gradients through DataParallelStep (RCCL all-reduce per bucket)."""
    from .launch import spawn_ranks, under_launcher
    argv = list(sys.argv[1:] if argv is None else argv)
    args = build_parser().parse_args(argv)
    if args.num_gpus > 1 and not under_launcher():
        script = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "train_iq.py")
        return spawn_ranks(args.num_gpus, script, argv)
    from . import synthetic
    from .trainer import shard_seed
    dist, rank, world = None, 0, 1
    if under_launcher() and int(os.environ["WORLD_SIZE"]) > 1:
        dist, rank, world = _init_distributed(args)
        if args.num_gpus != world and rank == 0:
            print("train_iq: --num_gpus %d but WORLD_SIZE %d (the launcher's world size is what runs)" % (args.num_gpus, world), file=sys.stderr)
    else:
        args.device = torch.device("cuda" if torch.cuda.is_available() and args.use_gpu else "cpu")
    args.root_dir = os.getcwd()
    if not args.synthetic:
        raise SystemExit("only --synthetic batches are available here (the HDF5 container needs h5py, absent from this image; "
                         "bltvqg_amd.batch.IQStore takes the six dataset arrays: TrainIQ.fit_from_producer)")
    if not os.path.exists(os.path.join(args.root_dir, args.emb_file)):
        args.emb_file = None
    vocab = SyntheticVocabulary(8000)
    factory = os.environ.get("BLT_TRAINER_FACTORY")
    model = _load_factory(factory)(vocab, args) if factory else TrainIQ(vocab, args).to(args.device)

    def loader():
        i = 0
        while True:      # global batch i is sharded by rank: every rank draws its own batch_size samples (weak scaling, like DDP)
            yield synthetic.make_batch(args.batch_size, len(vocab), args.latent_dim, seed=shard_seed(1234 + i * world, rank))
            i += 1
    try:
        model.fit(loader(), args.total_training_steps, dist=dist)
    finally:
        if dist is not None:
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
