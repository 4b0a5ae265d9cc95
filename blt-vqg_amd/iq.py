"""`models.IQ` drop-in: same constructor, `forward` 4-tuple, `switch_GVT_train_mode` and `state_dict` key set as the reference
(models/iq.py:22-152), with every operator underneath running in libbltvqg_hip.so through the train-step engine.

The module tree only HOLDS parameters (as views into the engine's flat fp32 buffers, under the reference's names, aliases
included); there is no PyTorch compute path and no CPU fallback: `forward` on a machine without the HIP library / a GPU raises.
"""
import math
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import _lib
from .engine import StepEngine, make_config
from .padded import PaddedLayout, needs_padding
from .trainer import init_reference_style


class _Node(nn.Module):
    """Pure container (a node of the reference's module tree)."""

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("this module only holds parameters; compute runs in the HIP engine via IQ.forward")


def _attach(root, name, tensor, kind):
    parts = name.split(".")
    node = root
    for p in parts[:-1]:
        if p not in node._modules:
            node.add_module(p, _Node())
        node = node._modules[p]
    leaf = parts[-1]
    if kind == "param":
        node.register_parameter(leaf, nn.Parameter(tensor, requires_grad=True))
    elif kind == "frozen":
        node.register_parameter(leaf, nn.Parameter(tensor, requires_grad=False))
    else:
        node.register_buffer(leaf, tensor)


def _is_buffer(name):
    return name.endswith("running_mean") or name.endswith("running_var")


def _bn_prefixes(frozen_names):
    return sorted({n[: -len(".running_mean")] for n in frozen_names if n.endswith(".running_mean")})


class _IQFunction(torch.autograd.Function):
    """One autograd node for the whole model: forward = engine forward, backward = engine backward from output gradients."""

    @staticmethod
    def forward(ctx, model, images, answers, response, target, eps, *params):
        eng = model._engine_for(images, answers, response, target)
        phase2 = bool(model.latent_transformer)
        model._step_seed += 1
        # autograd path: the nn.Parameters (views of the flat buffer) belong to whoever optimises them — a torch optimiser does not
        # tell the engine when it writes them, so no bf16 weight shadow written by an earlier FUSED step may be trusted here
        if model._pad is not None:      # padded widths: the engine computes on ITS buffers; bring them (or the module) up to date first
            model.sync_from_engine()
            model._master = "module"
            model._scatter_to_engine(eng)
            if eps is not None and eps.shape[1] != eng.cfg.latent_dim:      # latent noise of the real width; the pad columns get none
                eps = torch.nn.functional.pad(eps.float(), (0, eng.cfg.latent_dim - eps.shape[1]))
        eng.params_touched()
        eng.forward(images.contiguous().float(), answers.contiguous(), response.contiguous(), target.contiguous(),
                    None if eps is None else eps.contiguous().float(), phase2, model._base_seed + model._step_seed)
        eng.params_touched()      # ... nor may the shadow this forward just built outlive it: the caller's optimiser writes the parameters next
        # The engine keeps ONE set of saved activations per shape: stamp this forward so that backward can tell whether they are
        # still its own (a second forward of the same shape, or a second backward, would otherwise silently use the wrong ones)
        eng.generation = getattr(eng, "generation", 0) + 1
        ctx.eng, ctx.phase2, ctx.names, ctx.was_training = eng, phase2, model._train_names, bool(model.training)
        ctx.model = model
        ctx.generation = eng.generation
        output = eng.read(0)
        feats, recon = eng.read(2), eng.read(3)
        if model._pad is not None:
            mh = model._pad.maps["H"][0].to(feats.device)
            feats, recon = feats[:, mh].contiguous(), recon[:, mh].contiguous()      # back to the reference's 300 columns
            if model.training:
                model._gather_from_engine(eng, frozen_only=True)                     # BatchNorm running statistics
        stats = eng.read(4)
        if float(stats[6]) > 0:      # one host sync; the reference's loss code syncs five times per step (train_iq.py:98,103)
            raise _lib.HipError("%d token id(s) outside [0, %d) in the batch (vocabulary / dataset mismatch?)"
                                % (int(stats[6]), eng.cfg.vocab_size))
        if phase2:
            z_logit = eng.read(1)
            kld = stats[2].clone()
        else:
            z_logit = torch.zeros(0, device=images.device)
            kld = torch.zeros((), device=images.device)
        return output, z_logit, kld, feats, recon

    @staticmethod
    def backward(ctx, d_out, d_zl, d_kld, d_feats, d_recon):
        eng = ctx.eng
        if not ctx.was_training:
            raise RuntimeError("backward through a model.eval() forward is not implemented (BatchNorm backward is the train-mode one); "
                               "call model.train() for training steps")
        if getattr(eng, "generation", 0) != ctx.generation:
            raise RuntimeError("IQ backward: the engine's saved activations belong to a later forward of the same batch shape (or were "
                               "already consumed by a backward); run backward before the next forward, once (retain_graph / double "
                               "backward are not supported)")
        eng.generation += 1          # consumed: backward_external overwrites the logits with their gradient
        f = lambda t: None if t is None else t.contiguous().float()   # noqa: E731
        model = ctx.model
        if model._pad is not None:
            mh = model._pad.maps["H"][0].to(eng.flat_grad.device)

            def widen(t):      # gradients of the 300-wide feature tensors -> the engine's padded width (zeros at the pads)
                if t is None:
                    return None
                o = torch.zeros(t.shape[0], eng.cfg.hidden_dim, device=t.device, dtype=torch.float32)
                o[:, mh] = t.float()
                return o
            d_feats, d_recon = widen(d_feats), widen(d_recon)
        eng.backward_external(f(d_out), f(d_zl) if ctx.phase2 else None, float(d_kld) if (ctx.phase2 and d_kld is not None) else 0.0,
                              f(d_feats), f(d_recon))
        grads = []
        if model._pad is not None:
            flat_g = eng.flat_grad[model._pad_index(eng.flat_grad.device)[0]]
            for n in ctx.names:
                info = model._train_info[n]
                grads.append(flat_g[info.offset:info.offset + info.numel].view(info.shape).clone() if (ctx.phase2 or not info.late) else None)
            return (None, None, None, None, None, None) + tuple(grads)
        for n in ctx.names:
            info = eng.train_info[n]
            grads.append(eng.grad_view(n).clone() if (ctx.phase2 or not info.late) else None)
        return (None, None, None, None, None, None) + tuple(grads)


class IQ(nn.Module):
    """Information-maximising VQG model (reference models/iq.py:22).  `args` is the reference's namespace: emb_dim, hidden_dim,
    latent_dim, pwffn_dim, num_layers, num_heads, device, emb_file, root_dir (+ optional: precision in {"bf16","fp32"},
    attention_dropout, relu_dropout, resnet_weights = path of a torchvision resnet18 state dict; num_regions + region_dim > 0 =
    bottom-up feature mode, BASELINE configs[4]: `images` is then a [B, num_regions, region_dim] tensor of precomputed region
    features and `encoder_cnn` holds `region_proj` + `bn` instead of the ResNet — the reference has no such mode, SURVEY A2';
    region_pool = "attention" adds region-attention pooling with the extra parameter `encoder_cnn.region_attn.weight`)."""

    def __init__(self, latent_transformer, vocab, args, num_att_layers=2):
        super().__init__()
        self.vocab = vocab
        self.vocab_size = len(vocab.word2idx)
        self.latent_transformer = latent_transformer
        self.args = args
        if num_att_layers != 2:
            raise ValueError("image_reconstructor is the reference's 2-layer MLP (iq.py:46-48)")
        # The HIP kernels move activations in 16-byte vectors and tile the head width: feature widths that are not multiples of 8 (the
        # reference CLI defaults, train_iq.py:315-325: hidden_dim = latent_dim = 300, pwffn_dim = 600, 4 heads of 75) run on an engine
        # with PADDED widths (hidden 320 = 4 heads of 80 columns with 75 real ones, latent 304); the parameters keep the reference's shapes here
        # and are scattered into the padded layout (padded.py).  state_dict / checkpoints are unchanged.
        if int(args.emb_dim) % 4 != 0 or int(args.hidden_dim) % int(args.num_heads) != 0 or int(args.hidden_dim) % 2 != 0:
            raise ValueError("unsupported model widths for the MI355X engine: emb_dim (%s) must be a multiple of 4 and hidden_dim (%s) even and "
                             "divisible by num_heads (%s)" % (args.emb_dim, args.hidden_dim, args.num_heads))
        self._pad = (PaddedLayout(args.hidden_dim, args.latent_dim, args.pwffn_dim, args.num_heads)
                     if needs_padding(int(args.hidden_dim), int(args.latent_dim), int(args.pwffn_dim), int(args.num_heads)) else None)
        self._master = "module"        # padded mode: who holds the current parameter values, the module's buffers or the engine's
        self._dtype = _lib.F32 if getattr(args, "precision", "bf16") in ("fp32", "f32", 32) else _lib.BF16
        self._engines = {}
        self._primary = None
        self._base_seed = int(getattr(args, "seed", 0)) * 1000003
        self._step_seed = 0
        self.eps_generator = None      # optional torch.Generator for the latent noise (transformer_layers.py:45)
        # A shape-less probe engine gives the canonical parameter list; parameters start on the CPU like the reference's.
        probe = self._make_engine(1, 5, 21, 20, 224, 224, allocate=False)
        self._train_names = list(probe.train_info.keys())
        self._train_info, self._frozen_info = probe.train_info, probe.frozen_info
        n_t, n_f = probe.train_size, probe.frozen_size
        if self._pad is not None:      # reference-shaped parameters in a flat buffer of the module's own + the scatter index into the engine's
            self._train_info, self._pad_train_index, n_t = self._pad.build(probe.train_info)
            self._frozen_info, self._pad_frozen_index, n_f = self._pad.build(probe.frozen_info)
        flat_t = torch.zeros(n_t)
        flat_f = torch.zeros(n_f)
        self._install(flat_t, flat_f)
        init_reference_style(SimpleNamespace(train_info=self._train_info, frozen_info=self._frozen_info, device=torch.device("cpu"),
                                             view=self._cpu_view, lib=SimpleNamespace(bltvqg_engine_invalidate_frozen=lambda h: None), h=None),
                             seed=int(getattr(args, "seed", 0)), resnet_state=self._resnet_state(args))
        if getattr(args, "emb_file", None):
            self._load_embeddings(args)

    # ---- construction helpers -----------------------------------------------------------------------------------
    def _make_engine(self, B, Sa, Sp, T, h, w, allocate=True, device="cuda", training=True):
        a = self.args
        # module.eval() (Lightning's validation loop): nn.Dropout is the identity
        p_attn = float(getattr(a, "attention_dropout", 0.1)) if training else 0.0
        p_relu = float(getattr(a, "relu_dropout", 0.1)) if training else 0.0
        H, F, Z, dht = self._widths()
        cfg = make_config(B, H, F, Z, a.emb_dim, a.num_layers, a.num_heads, self.vocab_size, Sa, Sp, T,
                          (h, w), self._dtype, p_attn, p_relu,
                          float(getattr(a, "kl_ceiling", 0.5)), float(getattr(a, "aux_ceiling", 1.0)),
                          float(getattr(a, "image_recon_lambda", 0.1)), int(getattr(a, "num_regions", 0) or 0),
                          int(getattr(a, "region_dim", 0) or 0), getattr(a, "region_pool", 0) or 0, head_dim_true=dht)
        e = StepEngine(cfg, device if allocate else "cpu")
        return e

    def _widths(self):
        """(hidden, pwffn, latent, head_dim_true) the ENGINE is created with: the padded widths when the model's own are not tileable."""
        a = self.args
        if self._pad is None:
            return int(a.hidden_dim), int(a.pwffn_dim), int(a.latent_dim), 0
        return self._pad.Hp, self._pad.Fp, self._pad.Zp, self._pad.dh

    # ---- padded mode: the module's flat buffers (reference shapes) <-> the engine's (padded) --------------------------------------
    def _pad_index(self, dev):
        if getattr(self, "_pad_index_dev", None) is None or self._pad_index_dev[0].device != dev:
            self._pad_index_dev = (self._pad_train_index.to(dev), self._pad_frozen_index.to(dev))
        return self._pad_index_dev

    def _scatter_to_engine(self, eng):
        """module -> engine: every real element to its padded position; the pads were zeroed at allocation and stay zero."""
        it, if_ = self._pad_index(eng.flat_train.device)
        with torch.no_grad():
            eng.flat_train.index_copy_(0, it, self._flat_train.to(eng.flat_train.device))
            eng.flat_frozen.index_copy_(0, if_, self._flat_frozen.to(eng.flat_frozen.device))
        eng.params_changed()

    def _gather_from_engine(self, eng, frozen_only=False):
        it, if_ = self._pad_index(eng.flat_train.device)
        with torch.no_grad():
            if not frozen_only:
                self._flat_train.copy_(eng.flat_train[it])
            self._flat_frozen.copy_(eng.flat_frozen[if_])

    def sync_from_engine(self):
        """Padded mode, after fused steps (the engine's optimiser updates ITS buffers): bring the nn.Parameters up to date."""
        if self._pad is not None and self._master == "engine" and self._primary is not None:
            self._primary.optimizer_wait()
            self._gather_from_engine(self._primary)

    def sync_to_engine(self, for_fused=False):
        """Padded mode: make the engine's buffers current before it runs on them; for_fused: the engine's optimiser owns them from now on."""
        if self._pad is None or self._primary is None:
            return
        if self._master == "module":
            self._scatter_to_engine(self._primary)
        if for_fused:
            self._master = "engine"

    def state_dict(self, *a, **k):
        # a conv stack that runs one batch ahead writes the BatchNorm2d running statistics on the engine's conv stream: order this
        # stream behind it (the statistics then include that batch, whole)
        for e in self._engines.values():
            if hasattr(e, "conv_stream_wait") and e.bound and e.device.type == "cuda":
                e.conv_stream_wait()
        self.sync_from_engine()
        return super().state_dict(*a, **k)

    def _cpu_view(self, name, which=None):
        info = self._train_info.get(name) if which in (None, 0) and name in self._train_info else self._frozen_info[name]
        flat = self._flat_train if (which in (None, 0) and name in self._train_info) else self._flat_frozen
        return flat[info.offset:info.offset + info.numel].view(info.shape)

    def _install(self, flat_t, flat_f):
        """Creates the module tree (once) as views into the given flat buffers; later calls only re-point `.data`, so that
        Parameter objects (and any optimizer holding them) stay valid."""
        self._flat_train, self._flat_frozen = flat_t, flat_f
        built = "embedding" in self._modules

        def views():
            for name, info in self._train_info.items():
                yield name, flat_t[info.offset:info.offset + info.numel].view(info.shape), "param"
            for name, info in self._frozen_info.items():
                yield name, flat_f[info.offset:info.offset + info.numel].view(info.shape), ("buffer" if _is_buffer(name) else "frozen")

        if built:
            for name, v, kind in views():
                (self.get_buffer(name) if kind == "buffer" else self.get_parameter(name)).data = v
            for pre in _bn_prefixes(self._frozen_info.keys()):
                b = self.get_buffer(pre + ".num_batches_tracked")
                b.data = b.data.to(flat_t.device)
            return
        for name, v, kind in views():
            _attach(self, name, v, kind)
        for pre in _bn_prefixes(self._frozen_info.keys()):
            _attach(self, pre + ".num_batches_tracked", torch.zeros((), dtype=torch.long, device=flat_t.device), "buffer")
        # aliases of the reference tree (iq.py:32,41,43; encoder_transformer.py:8-10; decoder_transformer.py:9)
        self.answer_encoder.add_module("embedding", self.embedding)
        self.answer_encoder.add_module("latent_layer", self.latent_layer)
        self.decoder.add_module("embedding", self.embedding)

    @staticmethod
    def _resnet_state(args):
        path = getattr(args, "resnet_weights", None)
        if not path:
            return None
        return torch.load(path, map_location="cpu", weights_only=True)

    def _load_embeddings(self, args):
        """GloVe-style text file, reference iq.py:60-71."""
        import os
        path = os.path.join(getattr(args, "root_dir", "."), args.emb_file)
        if not os.path.exists(path):
            return
        w = self._cpu_view("embedding.0.weight") if self._flat_train.device.type == "cpu" else None
        if w is None:
            return
        with open(path) as fh:
            for line in fh:
                sp = line.split()
                if len(sp) == args.emb_dim + 1 and sp[0] in self.vocab.word2idx:
                    w[self.vocab.word2idx[sp[0]]] = torch.tensor([float(x) for x in sp[1:]])

    # ---- reference API -------------------------------------------------------------------------------------------
    def switch_GVT_train_mode(self, new_mode):
        """reference iq.py:51-54."""
        self.latent_transformer = new_mode

    def load_state_dict(self, state_dict, strict=True, **kw):
        r = super().load_state_dict(state_dict, strict=strict, **kw)
        self._master = "module"
        for e in self._engines.values():
            e.lib.bltvqg_engine_invalidate_frozen(e.h)
        return r

    def _aliased(self):
        """True when the parameters still live inside the primary engine's flat buffers (False after .to()/.cuda()/.float())."""
        e = self._primary
        first, last = self._train_names[0], self._train_names[-1]
        base = self._flat_train if self._pad is not None else e.flat_train      # padded mode: the module's own device buffers
        if self._pad is not None and self._flat_train.device != e.flat_train.device:
            return False
        for n in (first, last):
            if self.get_parameter(n).data_ptr() != base.data_ptr() + 4 * self._train_info[n].offset:
                return False
        return True

    def _engine_for(self, images, answers, response, target):
        if not images.is_cuda:
            raise RuntimeError("IQ.forward runs on MI355X only (libbltvqg_hip.so); there is no CPU fallback. Move the batch to the GPU.")
        h, w = (images.shape[2], images.shape[3]) if images.dim() == 4 else (0, 0)       # region mode: [B, regions, dim]
        return self.engine_for_shape(images.shape[0], answers.shape[1], response.shape[1], target.shape[1], h, w, images.device)

    def engine_for_shape(self, B, Sa, Sp, T, h, w, device):
        """The (cached) engine of one static shape; used directly by loops that fill the engine's stem input themselves
        (DeviceBatchProducer.batch(engine=...))."""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("the HIP engine runs on MI355X only; there is no CPU fallback")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        # train / eval engines differ in dropout (config) and BatchNorm mode; they share the parameter buffers
        key = (int(B), int(Sa), int(Sp), int(T), int(h), int(w), device.index, bool(self.training))
        eng = self._engines.get(key)
        if eng is None:
            eng = self._make_engine(*key[:6], device=device, training=self.training)
            if self._primary is None:
                eng.allocate()
                self._primary = eng
                self._adopt(eng)
            else:
                eng.allocate(share_from=self._primary)
            self._engines[key] = eng
            eng.set_bn_train(self.training)      # eval: BatchNorm2d / BatchNorm1d use their running statistics and do not update them
        if not self._aliased():
            self._adopt(self._primary)
        return eng

    def _adopt(self, eng):
        """Copies the current parameter values into the engine's flat device buffers and re-points the module tree at them."""
        if self._pad is not None:
            # padded mode: the module keeps flat buffers of its own (reference shapes) on the engine's device; values reach the engine's
            # padded buffers through the scatter index
            dev = eng.flat_train.device
            sd = {k: v.detach() for k, v in super().state_dict().items()}
            flat_t = torch.zeros(self._flat_train.numel(), device=dev)
            flat_f = torch.zeros(self._flat_frozen.numel(), device=dev)
            with torch.no_grad():
                for name, info in self._train_info.items():
                    flat_t[info.offset:info.offset + info.numel].view(info.shape).copy_(sd[name].to(dev, torch.float32))
                for name, info in self._frozen_info.items():
                    flat_f[info.offset:info.offset + info.numel].view(info.shape).copy_(sd[name].to(dev, torch.float32))
            self._install(flat_t, flat_f)
            self._master = "module"
            self._scatter_to_engine(eng)
            return
        with torch.no_grad():
            eng.load_state({k: v.detach() for k, v in self.state_dict().items()})
        self._install(eng.flat_train, eng.flat_frozen)
        for e in self._engines.values():
            e.lib.bltvqg_engine_invalidate_frozen(e.h)

    def forward(self, images, answers, response, target, eps=None):
        """reference iq.py:82-114.  Returns (output (B,T,V), z_logit (B,V) | None, kld | None, (image_features, reconstructed))."""
        if self.latent_transformer and eps is None:
            eps = torch.randn(images.shape[0], self.args.latent_dim, device=images.device, generator=self.eps_generator)
        params = [self.get_parameter(n) for n in self._train_names]
        output, z_logit, kld, feats, recon = _IQFunction.apply(self, images, answers, response, target, eps, *params)
        if self.training:
            with torch.no_grad():
                for pre in _bn_prefixes(self._frozen_info.keys()):
                    self.get_buffer(pre + ".num_batches_tracked").add_(1)
        if not self.latent_transformer:
            return output, None, None, (feats, recon)
        return output, z_logit, kld, (feats, recon)

    def decode_greedy(self, images, answers, max_decode_length=50, eps=None):
        """reference models/iq.py:117-152.  Returns (sentences, top_args (B, L+1, 6), top_vals (B, L+1, 6)).  BatchNorm follows
        `self.training` (Lightning calls this under model.eval(): running statistics).  `eps` injects the latent noise."""
        if not images.is_cuda:
            raise RuntimeError("IQ.decode_greedy runs on MI355X only (libbltvqg_hip.so); there is no CPU fallback.")
        T = max_decode_length + 1
        B = images.shape[0]
        h, w = (images.shape[2], images.shape[3]) if images.dim() == 4 else (0, 0)
        key = ("decode", B, answers.shape[1], T, h, w, images.device.index)
        eng = self._engines.get(key)
        if eng is None:
            a = self.args
            Hh, Ff, Zz, dht = self._widths()
            cfg = make_config(B, Hh, Ff, Zz, a.emb_dim, a.num_layers, a.num_heads, self.vocab_size,
                              answers.shape[1], 21, T, (h, w), self._dtype, 0.0, 0.0, num_regions=int(getattr(a, "num_regions", 0) or 0),
                              region_dim=int(getattr(a, "region_dim", 0) or 0), region_pool=getattr(a, "region_pool", 0) or 0,
                              head_dim_true=dht)
            eng = StepEngine(cfg, images.device)
            if self._primary is None:
                eng.allocate()
                self._primary = eng
                self._adopt(eng)
            else:
                eng.allocate(share_from=self._primary)
            self._engines[key] = eng
        if not self._aliased():
            self._adopt(self._primary)
        phase2 = bool(self.latent_transformer)
        if phase2 and eps is None:
            eps = torch.randn(B, self.args.latent_dim, device=images.device, generator=self.eps_generator)
        if self._pad is not None:
            self.sync_to_engine()
            if eps is not None and eps.shape[1] != eng.cfg.latent_dim:
                eps = torch.nn.functional.pad(eps.float(), (0, eng.cfg.latent_dim - eps.shape[1]))
        tokens, top_idx, top_vals = eng.decode_greedy(images.contiguous().float(), answers.contiguous(),
                                                      None if eps is None else eps.contiguous().float(), phase2, train_bn=self.training)
        eos = self.vocab.word2idx[self.vocab.SYM_EOS] if hasattr(self.vocab, "SYM_EOS") else 3
        sentences = []
        for row in tokens.cpu().tolist():
            st = ""
            for tok in row:
                if tok == eos:
                    break
                st += self.vocab.idx2word[tok] + " "
            sentences.append(st)
        return sentences, top_idx.float(), top_vals

    # ---- fused train-step access (used by TrainIQ.fused_training_step and bench.py) ---------------------------------
    def engine(self, images, answers, response, target):
        return self._engine_for(images, answers, response, target)
