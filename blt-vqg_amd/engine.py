"""Python handle on the HIP train-step engine (csrc/engine.hip).

Owns the device memory the C library borrows: one flat fp32 buffer of trainable parameters (and same-layout gradient /
Adam-moment buffers), one flat fp32 buffer of frozen backbone parameters + BatchNorm running statistics, and the
activation workspace.  Parameters are exposed as views into the flat buffers under the reference's state_dict names.
"""
import ctypes
from collections import OrderedDict

import torch

from . import _lib
from ._lib import Config, check, ptr, stream_ptr

F32, BF16 = _lib.F32, _lib.BF16


def make_config(batch, hidden_dim, pwffn_dim, latent_dim, emb_dim, num_layers, num_heads, vocab_size, len_context=5,
                len_posterior=21, len_target=20, image_hw=(224, 224), dtype=BF16, attention_dropout=0.1, relu_dropout=0.1,
                kl_ceiling=0.5, aux_ceiling=1.0, image_recon_lambda=0.1, num_regions=0, region_dim=0, region_pool=0, head_dim_true=0):
    """num_regions > 0: bottom-up feature mode (BASELINE configs[4]) — `images` is [B, num_regions, region_dim] fp32; region_pool = 0 mean
    over the regions, 1 (or "attention") region-attention pooling."""
    region_pool = {"mean": 0, "attention": 1}.get(region_pool, region_pool)
    return Config(batch=batch, hidden_dim=hidden_dim, pwffn_dim=pwffn_dim, latent_dim=latent_dim, emb_dim=emb_dim,
                  num_layers=num_layers, num_heads=num_heads, vocab_size=vocab_size, len_context=len_context,
                  len_posterior=len_posterior, len_target=len_target, image_h=image_hw[0], image_w=image_hw[1], dtype=dtype,
                  attention_dropout=attention_dropout, relu_dropout=relu_dropout, kl_ceiling=kl_ceiling,
                  aux_ceiling=aux_ceiling, image_recon_lambda=image_recon_lambda, num_regions=num_regions, region_dim=region_dim,
                  region_pool=int(region_pool), head_dim_true=int(head_dim_true))


class ParamInfo(object):
    __slots__ = ("name", "offset", "numel", "shape", "late")

    def __init__(self, name, offset, numel, shape, late):
        self.name, self.offset, self.numel, self.shape, self.late = name, offset, numel, shape, late


class StepEngine(object):
    """One engine = one static shape (batch, sequence lengths, image size, dtype)."""

    def __init__(self, cfg, device="cuda"):
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        self.h = self.lib.bltvqg_engine_create(ctypes.byref(cfg))
        if not self.h:
            raise _lib.HipError("bltvqg_engine_create: " + self.lib.bltvqg_last_error_string().decode())
        self.h = ctypes.c_void_p(self.h)
        self.train_info = self._infos(0)
        self.frozen_info = self._infos(1)
        self.train_size = int(self.lib.bltvqg_engine_flat_size(self.h, 0))
        self.frozen_size = int(self.lib.bltvqg_engine_flat_size(self.h, 1))
        self.late_offset = int(self.lib.bltvqg_engine_late_offset(self.h))
        self.workspace_bytes = int(self.lib.bltvqg_engine_workspace_bytes(self.h))
        self.flat_train = self.flat_grad = self.adam_m = self.adam_v = self.flat_frozen = self.workspace = None
        self.bound = False

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.bltvqg_engine_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _infos(self, which):
        n = self.lib.bltvqg_engine_num_params(self.h, which)
        out = OrderedDict()
        buf = ctypes.create_string_buffer(256)
        off, numel = _lib.L(), _lib.L()
        dims = (ctypes.c_int32 * 4)()
        ndim, late = ctypes.c_int32(), ctypes.c_int32()
        for i in range(n):
            check(self.lib.bltvqg_engine_param_info(self.h, which, i, buf, 256, ctypes.byref(off), ctypes.byref(numel), dims,
                                                    ctypes.byref(ndim), ctypes.byref(late)), "engine_param_info")
            name = buf.value.decode()
            shape = tuple(int(dims[k]) for k in range(ndim.value))
            out[name] = ParamInfo(name, int(off.value), int(numel.value), shape, int(late.value))
        return out

    # ------------------------------------------------------------------------------------------------
    def allocate(self, share_from=None):
        """Allocates the flat buffers + workspace on the device and binds them.  `share_from`: another engine of the same
        model (different batch shape) whose parameter / gradient / optimiser buffers are reused."""
        dev = self.device
        if share_from is not None:
            assert share_from.train_size == self.train_size and share_from.frozen_size == self.frozen_size
            self.flat_train, self.flat_grad = share_from.flat_train, share_from.flat_grad
            self.adam_m, self.adam_v, self.flat_frozen = share_from.adam_m, share_from.adam_v, share_from.flat_frozen
            # ... and the Adam bias-correction counters that belong to those moments (a fresh engine starting at t = 1 against warm
            # moments would scale its first update by ~0.3)
            check(self.lib.bltvqg_engine_share_optimizer_state(self.h, share_from.h), "engine_share_optimizer_state")
        else:
            self.flat_train = torch.zeros(self.train_size, dtype=torch.float32, device=dev)
            self.flat_grad = torch.zeros(self.train_size, dtype=torch.float32, device=dev)
            self.adam_m = torch.zeros(self.train_size, dtype=torch.float32, device=dev)
            self.adam_v = torch.zeros(self.train_size, dtype=torch.float32, device=dev)
            self.flat_frozen = torch.zeros(self.frozen_size, dtype=torch.float32, device=dev)
        self.workspace = torch.empty(self.workspace_bytes + 256, dtype=torch.uint8, device=dev)
        self._bind()

    def _bind(self):
        base = self.workspace.data_ptr()
        aligned = (base + 255) // 256 * 256
        torch.cuda.synchronize(self.device)
        check(self.lib.bltvqg_engine_bind(self.h, ptr(self.flat_train), ptr(self.flat_grad), ptr(self.adam_m), ptr(self.adam_v),
                                          ptr(self.flat_frozen), ctypes.c_void_p(aligned), self.workspace_bytes), "engine_bind")
        self.bound = True

    def view(self, name, which=None):
        """Tensor view of a parameter inside its flat buffer (state_dict name)."""
        if name in self.train_info and which in (None, 0):
            i = self.train_info[name]
            return self.flat_train[i.offset:i.offset + i.numel].view(i.shape)
        i = self.frozen_info[name]
        return self.flat_frozen[i.offset:i.offset + i.numel].view(i.shape)

    def grad_view(self, name):
        i = self.train_info[name]
        return self.flat_grad[i.offset:i.offset + i.numel].view(i.shape)

    def load_state(self, state):
        """Copies a reference-style state dict (name -> tensor) into the flat buffers."""
        with torch.no_grad():
            for name in self.train_info:
                self.view(name, 0).copy_(state[name].to(self.device, torch.float32))
            for name in self.frozen_info:
                self.view(name, 1).copy_(state[name].to(self.device, torch.float32))
        self.lib.bltvqg_engine_invalidate_frozen(self.h)

    # ------------------------------------------------------------------------------------------------
    def forward(self, images, context, posterior, target, eps=None, phase2=False, seed=0):
        c = self.cfg
        if images is None:      # the caller filled the stem input itself (image_input(), DeviceBatchProducer.batch(engine=...))
            assert c.num_regions == 0
        else:
            assert images.is_cuda and images.dtype == torch.float32 and images.is_contiguous()
            expect = (c.batch, c.num_regions, c.region_dim) if c.num_regions > 0 else (c.batch, 3, c.image_h, c.image_w)
            assert tuple(images.shape) == expect, (tuple(images.shape), expect)
        for t, n in ((context, c.len_context), (posterior, c.len_posterior), (target, c.len_target)):
            assert t.is_cuda and t.dtype == torch.int64 and t.is_contiguous() and tuple(t.shape) == (c.batch, n), (t.shape, n)
        if eps is not None:
            assert eps.is_cuda and eps.dtype == torch.float32 and eps.is_contiguous() and tuple(eps.shape) == (c.batch, c.latent_dim)
        check(self.lib.bltvqg_engine_forward(self.h, ptr(images), ptr(context), ptr(posterior), ptr(target), ptr(eps),
                                             1 if phase2 else 0, int(seed), stream_ptr()), "engine_forward")

    def prefetch_images(self, images):
        """Enqueue the frozen conv stack of the NEXT batch on the engine's conv stream (bltvqg_engine_prefetch_images); the next
        forward() must then be called with images=None.  `images` None: the caller filled image_input() itself."""
        c = self.cfg
        if images is not None:
            assert images.is_cuda and images.dtype == torch.float32 and images.is_contiguous()
            assert tuple(images.shape) == (c.batch, 3, c.image_h, c.image_w), tuple(images.shape)
        check(self.lib.bltvqg_engine_prefetch_images(self.h, ptr(images), stream_ptr()), "engine_prefetch_images")

    def set_prefetch_split(self, stages):
        """Leading stages (1..10) of the conv stack a prefetch runs ahead; 10 = all (bltvqg_engine_set_prefetch_split)."""
        check(self.lib.bltvqg_engine_set_prefetch_split(self.h, int(stages)), "engine_set_prefetch_split")
        self.prefetch_split = int(stages)

    def prefetch_pending(self):
        return int(self.lib.bltvqg_engine_prefetch_pending(self.h))

    @staticmethod
    def cu_mask(cus_per_xcd_lo, cus_per_xcd_hi, n_xcd=8, cus_per_xcd=32):
        """CU mask words selecting CUs [lo, hi) of every XCD.  Bit i of a HIP CU mask addresses XCD i % n_xcd, CU i // n_xcd
        (tests/test_partition_gpu.py checks this on the device with the hardware-id probe)."""
        words = [0] * ((n_xcd * cus_per_xcd + 31) // 32)
        for cu in range(cus_per_xcd_lo, cus_per_xcd_hi):
            for x in range(n_xcd):
                b = cu * n_xcd + x
                words[b // 32] |= 1 << (b % 32)
        return words

    def set_cu_masks(self, chain=None, side=None, conv=None, chain_cus=0):
        """Complementary CU partition of the engine's streams (bltvqg_engine_set_cu_masks): lists of 32-bit mask words or None."""
        def arr(m):
            return None if m is None else (ctypes.c_uint32 * 8)(*(list(m) + [0] * (8 - len(m))))
        torch.cuda.synchronize(self.device)
        a, b, c_ = arr(chain), arr(side), arr(conv)
        check(self.lib.bltvqg_engine_set_cu_masks(self.h, a, b, c_, 8, int(chain_cus)), "engine_set_cu_masks")
        self._chain_stream = None

    def chain_stream(self):
        """torch stream (engine-owned, under the chain CU mask) to run forward / loss_backward / optimizer_step on."""
        if getattr(self, "_chain_stream", None) is None:
            p = ctypes.c_void_p()
            check(self.lib.bltvqg_engine_chain_stream(self.h, ctypes.byref(p)), "engine_chain_stream")
            self._chain_stream = torch.cuda.ExternalStream(p.value, device=self.device)
        return self._chain_stream

    def adopt_conv_stream(self, stream):
        """Prefetched conv stacks run on `stream` (a torch.cuda.Stream the caller keeps alive) instead of an engine-owned stream."""
        check(self.lib.bltvqg_engine_adopt_conv_stream(self.h, ctypes.c_void_p(stream.cuda_stream)), "engine_adopt_conv_stream")
        self._adopted_conv_stream = stream      # keep it alive as long as the engine

    def conv_stream(self):
        """The engine's prefetch stream as a torch stream (diagnostics only)."""
        p = ctypes.c_void_p()
        check(self.lib.bltvqg_engine_conv_stream(self.h, ctypes.byref(p)), "engine_conv_stream")
        return torch.cuda.ExternalStream(p.value, device=self.device)

    def conv_stream_wait(self):
        """Orders the current stream behind everything enqueued on the engine's conv look-ahead stream (bltvqg_engine_conv_stream_wait):
        before reading flat_frozen (BatchNorm2d running statistics) while a prefetched stack may still be running."""
        check(self.lib.bltvqg_engine_conv_stream_wait(self.h, stream_ptr()), "engine_conv_stream_wait")

    def image_input(self):
        """(device pointer, Hp, Wp, dtype) of the engine's zero-bordered NHWC4 stem input (bltvqg_engine_image_input)."""
        p, hp, wp, dt = ctypes.c_void_p(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        check(self.lib.bltvqg_engine_image_input(self.h, ctypes.byref(p), ctypes.byref(hp), ctypes.byref(wp), ctypes.byref(dt)),
              "engine_image_input")
        return p, hp.value, wp.value, dt.value

    def decode_greedy(self, images, context, eps=None, phase2=False, train_bn=False):
        """Greedy decode over len_target steps; returns (tokens [B,T] int32, top_idx [B,T,6] int32, top_val [B,T,6] fp32)."""
        c = self.cfg
        tokens = torch.zeros(c.batch, c.len_target, dtype=torch.int32, device=self.device)
        top_idx = torch.zeros(c.batch, c.len_target, 6, dtype=torch.int32, device=self.device)
        top_val = torch.zeros(c.batch, c.len_target, 6, dtype=torch.float32, device=self.device)
        check(self.lib.bltvqg_engine_decode_greedy(self.h, ptr(images), ptr(context), ptr(eps), 1 if phase2 else 0, 1 if train_bn else 0,
                                                   ptr(tokens), ptr(top_idx), ptr(top_val), stream_ptr()), "engine_decode_greedy")
        return tokens, top_idx, top_val

    def trust_shadows(self, on=True):
        """Promise that parameters change only through this engine family's optimizer_step (or are reported via load_state /
        params_changed): the bf16 weight shadow is then written by the update itself instead of being rebuilt every forward."""
        check(self.lib.bltvqg_engine_trust_shadows(self.h, 1 if on else 0), "engine_trust_shadows")

    def params_changed(self):
        """Call after writing flat_train / flat_frozen from outside the engine (a broadcast, an in-place edit)."""
        self.lib.bltvqg_engine_invalidate_frozen(self.h)

    def params_touched(self):
        """The trainable parameters may have changed outside the engine's own optimiser (a torch optimiser on the views): every bf16
        weight shadow is rebuilt from fp32 on the next forward, whatever trust_shadows() says."""
        self.lib.bltvqg_engine_invalidate_params(self.h)

    def set_bn_train(self, train):
        check(self.lib.bltvqg_engine_set_bn_train(self.h, 1 if train else 0), "engine_set_bn_train")

    def loss_backward(self, kl_weight=0.0):
        check(self.lib.bltvqg_engine_loss_backward(self.h, float(kl_weight), stream_ptr()), "engine_loss_backward")

    def backward_external(self, d_output=None, d_zlogit=None, d_kld=0.0, d_feats=None, d_recon=None):
        check(self.lib.bltvqg_engine_backward_external(self.h, ptr(d_output), ptr(d_zlogit), float(d_kld), ptr(d_feats), ptr(d_recon),
                                                       stream_ptr()), "engine_backward_external")

    def optimizer_step(self, lr, max_norm=5.0, beta1=0.9, beta2=0.999, eps=1e-8, overlap=False):
        """clip_grad_norm_(max_norm) + Adam.  overlap=True enqueues the update on the engine's optimiser stream: the next forward()
        starts its frozen CNN at once and only the parameter consumers wait; call optimizer_wait() before reading the flat
        parameter / moment tensors on the current stream."""
        fn = self.lib.bltvqg_engine_optimizer_step_async if overlap else self.lib.bltvqg_engine_optimizer_step
        check(fn(self.h, float(lr), float(max_norm), float(beta1), float(beta2), float(eps), stream_ptr()), "engine_optimizer_step")

    def adam_steps(self):
        """(steps taken by the always-trained region, by the latent-phase-only region): Adam's bias-correction counters."""
        a, b = ctypes.c_int32(), ctypes.c_int32()
        check(self.lib.bltvqg_engine_adam_steps(self.h, ctypes.byref(a), ctypes.byref(b)), "engine_adam_steps")
        return a.value, b.value

    def set_adam_steps(self, main, late):
        check(self.lib.bltvqg_engine_set_adam_steps(self.h, int(main), int(late)), "engine_set_adam_steps")

    def optimizer_wait(self):
        """Orders the current stream behind a pending overlapped optimiser update (no-op otherwise)."""
        check(self.lib.bltvqg_engine_optimizer_wait(self.h, stream_ptr()), "engine_optimizer_wait")

    _READ_SHAPES = {0: lambda c: (c.batch, c.len_target, c.vocab_size), 1: lambda c: (c.batch, c.vocab_size),
                    2: lambda c: (c.batch, c.hidden_dim), 3: lambda c: (c.batch, c.hidden_dim), 4: lambda c: (8,),
                    5: lambda c: (c.batch, c.len_context, c.hidden_dim), 6: lambda c: (c.batch, c.len_target, c.hidden_dim)}

    def read(self, what):
        out = torch.empty(self._READ_SHAPES[what](self.cfg), dtype=torch.float32, device=self.device)
        check(self.lib.bltvqg_engine_read(self.h, what, ptr(out), stream_ptr()), "engine_read")
        return out

    def stats(self, check_ids=True):
        """dict of python floats (one host sync): rec, img, kld, aux, grad_norm, n_targets, bad_ids.  Raises when the last forward saw
        token ids outside [0, vocab_size) (the reference's nn.Embedding raises a device-side index error there)."""
        s = self.read(4).tolist()
        if check_ids and s[6] > 0:
            raise _lib.HipError("%d token id(s) outside [0, %d) in the last batch (vocabulary / dataset mismatch?); they were treated as <pad>"
                                % (int(s[6]), self.cfg.vocab_size))
        return dict(rec=s[0], img=s[1], kld=s[2], aux=s[3], grad_norm=s[4] ** 0.5, n_targets=s[5], bad_ids=s[6])

    PROFILE_CONV, PROFILE_GEMM = 1, 2

    def profile_enable(self, mask=1):
        """mask: PROFILE_CONV | PROFILE_GEMM (True = convolutions only, False / 0 = pause)."""
        check(self.lib.bltvqg_engine_profile_enable(self.h, int(mask)), "profile_enable")

    def profile_read(self, cls=0):
        """(total kernel ms, launches, algorithmic flops) of class `cls` (0 = convolutions, 1 = Linear GEMMs) since the last read;
        synchronises on the recorded events."""
        ms, n, fl = ctypes.c_double(), ctypes.c_int32(), ctypes.c_double()
        by = (ctypes.c_double * 4)()
        check(self.lib.bltvqg_engine_profile_read_streams(self.h, int(cls), ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), by), "profile_read")
        self.last_profile_by_stream_ms = [float(x) for x in by]      # [caller's stream, side 0, side 1, conv look-ahead stream]
        return ms.value, n.value, fl.value

    def set_bucket_flush(self, on=True):
        """In-stack weight-gradient flushes at every gradient-bucket boundary (for the data-parallel exchange; off on one GPU)."""
        check(self.lib.bltvqg_engine_set_bucket_flush(self.h, 1 if on else 0), "engine_set_bucket_flush")

    def buckets(self):
        out = []
        off, n, late = _lib.L(), _lib.L(), ctypes.c_int32()
        for i in range(self.lib.bltvqg_engine_num_buckets(self.h)):
            check(self.lib.bltvqg_engine_bucket_info(self.h, i, ctypes.byref(off), ctypes.byref(n), ctypes.byref(late)), "bucket_info")
            out.append((int(off.value), int(n.value), int(late.value)))
        return out

    def bucket_wait(self, i, stream):
        check(self.lib.bltvqg_engine_bucket_wait(self.h, i, ctypes.c_void_p(stream.cuda_stream)), "bucket_wait")
