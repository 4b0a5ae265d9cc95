"""The frozen conv stack one batch ahead (bltvqg_engine_prefetch_images) and the CU partition of the engine's streams.

reference: models/encoder_cnn.py:18-19 freezes the backbone, :30-35 is the forward the prefetch splits at the trainable fc.  A step whose
conv stack ran ahead — whole or the leading stages only — must be BIT-identical to the inline step: image feature, BatchNorm2d running
statistics (they advance in batch order), losses, gradients and the updated parameters."""
import ctypes

import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu


def _engine(dtype, state, B, hw, cfg, dropout=0.1):
    from bltvqg_amd.engine import StepEngine, make_config
    c = make_config(B, cfg.hidden_dim, cfg.pwffn_dim, cfg.latent_dim, cfg.emb_dim, cfg.num_layers, cfg.num_heads, cfg.vocab_size,
                    image_hw=(hw, hw), dtype=dtype, attention_dropout=dropout, relu_dropout=dropout)
    e = StepEngine(c)
    e.allocate()
    e.load_state(state)
    return e


def _images(B, hw, n):
    g = torch.Generator().manual_seed(5)
    return [torch.randn(B, 3, hw, hw, generator=g).cuda() for _ in range(n)]


def _run(e, d, imgs, mode, split=10):
    """mode: 'inline' | 'before' (prefetch of batch i+1 enqueued before forward(i)) | 'after' (after forward(i))"""
    outs = []
    n = len(imgs)
    if mode != "inline":
        e.set_prefetch_split(split)
        e.prefetch_images(imgs[0])
    for i in range(n):
        if mode == "before" and i + 1 < n:
            e.prefetch_images(imgs[i + 1])
        e.forward(imgs[i] if mode == "inline" else None, d["answers"], d["posteriors"], d["questions"], d["eps"], True, 11 + i)
        if mode == "after" and i + 1 < n:
            e.prefetch_images(imgs[i + 1])
        feats = e.read(2).clone()
        e.loss_backward(0.4)
        st = e.read(4).clone()
        e.optimizer_step(1e-3, 5.0)
        torch.cuda.synchronize()
        # frozen buffer: only the backbone's part (conv weights + BatchNorm2d running statistics) — the BatchNorm1d statistics behind the
        # trainable fc follow the parameters, which are reproducible to fp32-atomic order only (tests/test_determinism_gpu.py)
        hi = max(i_.offset + i_.numel for n_, i_ in e.frozen_info.items() if n_.startswith("encoder_cnn.cnn."))
        outs.append((feats.cpu(), st.cpu()[:4], e.flat_frozen[:hi].clone().cpu(), e.flat_train.clone().cpu()))
    return outs


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("mode,split", [("before", 10), ("after", 10), ("after", 1), ("after", 5), ("after", 9)])
def test_prefetched_step_is_bit_identical_to_inline(dtype, mode, split):
    z, cfg, state, batch = load_golden("tiny")
    B, hw = int(z["meta_cfg"][7]), 64
    d = {k: v.cuda() for k, v in batch.items()}
    imgs = _images(B, hw, 4)
    ref = _run(_engine(dtype, state, B, hw, cfg, dropout=0.0), d, imgs, "inline")
    got = _run(_engine(dtype, state, B, hw, cfg, dropout=0.0), d, imgs, mode, split)
    for i, (r, g) in enumerate(zip(ref, got)):
        if i == 0:
            assert torch.equal(r[0], g[0]), ("image feature", i)      # same kernels on the same data: bit-identical
        else:                                                         # later steps: through parameters that went through fp32 atomics
            assert float((r[0] - g[0]).abs().max()) < 1e-4, ("image feature", i)
        # (a look-ahead run has already folded batch i+1 into the running statistics when step i ends: compare once no batch is ahead)
        if i == len(ref) - 1:
            assert torch.equal(r[2], g[2]), ("BatchNorm running statistics", i)
        # the losses and the update go through fp32 atomics (split-K, loss sums): reproducible to accumulation order, like
        # test_determinism_gpu.  From the third step on the bound is relative: at this test's lr = 1e-3 Adam turns the sign of a
        # near-zero gradient element — which the atomics' order decides — into a 2e-3 parameter difference, and two fresh INLINE runs of the
        # bf16 engine were seen to land on either of two trajectories 3e-4 apart in the KL term (round 4; scratch/bistable.py).  What a
        # mis-ordered look-ahead would break — image feature, BatchNorm statistics — is held tight above and below.
        if i < 2:
            assert float((r[1] - g[1]).abs().max()) < 1e-4, ("losses", i, r[1].tolist(), g[1].tolist())
        else:
            assert float(((r[1] - g[1]).abs() / r[1].abs().clamp(min=1.0)).max()) < 1e-3, ("losses", i, r[1].tolist(), g[1].tolist())
        # Adam turns a gradient of either sign into a step of +-lr: elements whose tiny gradient differs in the last bits may move apart by
        # 2 lr from the second step on, so the bound there is on the mean
        if i == 0:
            assert float((r[3] - g[3]).abs().max()) < 1e-5, ("parameters", i)
        else:
            assert float((r[3] - g[3]).abs().mean()) < 1e-6, ("parameters", i)


def test_prefetch_misuse_is_refused():
    from bltvqg_amd._lib import HipError
    z, cfg, state, batch = load_golden("tiny")
    B, hw = int(z["meta_cfg"][7]), 64
    d = {k: v.cuda() for k, v in batch.items()}
    imgs = _images(B, hw, 3)
    e = _engine(0, state, B, hw, cfg, dropout=0.0)
    e.prefetch_images(imgs[0])
    with pytest.raises(HipError):          # a prefetched batch is pending: its images were given to prefetch_images
        e.forward(imgs[0], d["answers"], d["posteriors"], d["questions"], d["eps"], True, 1)
    e.forward(None, d["answers"], d["posteriors"], d["questions"], d["eps"], True, 1)
    e.prefetch_images(imgs[1])
    with pytest.raises(HipError):          # the second pending batch would overwrite the slot the un-backpropagated step still reads
        e.prefetch_images(imgs[2])
    e.loss_backward(0.1)
    e.prefetch_images(imgs[2])
    with pytest.raises(HipError):
        e.prefetch_images(imgs[2])          # three pending
    with pytest.raises(HipError):
        e.set_prefetch_split(3)             # not while batches are pending
    assert e.prefetch_pending() == 2
    torch.cuda.synchronize()


def _cu_set(lib, stream, n=4096):
    from bltvqg_amd._lib import check, ptr
    out = torch.zeros(n * 2, dtype=torch.int32, device="cuda")
    import gpu_ops
    check(gpu_ops.exp_lib().bltvqg_hw_id_probe(ptr(out), n, 2000, ctypes.c_void_p(stream.cuda_stream)), "probe")
    torch.cuda.synchronize()
    o = out.view(n, 2).cpu()
    hw, xcc = o[:, 0], o[:, 1]
    return set(zip(xcc.tolist(), ((hw >> 13) & 7).tolist(), ((hw >> 12) & 1).tolist(), ((hw >> 8) & 15).tolist()))


def test_cu_masks_partition_the_chip_and_keep_the_step_correct():
    """Complementary masks: the chain stream and the conv stream land on disjoint CU sets that together cover the chip, the same number
    of CUs on every XCD; a masked step equals the unmasked one (same kernels; the GEMM planner may pick other tiles, so not bit-exact)."""
    from bltvqg_amd.engine import StepEngine
    z, cfg, state, batch = load_golden("tiny")
    B, hw = int(z["meta_cfg"][7]), 64
    d = {k: v.cuda() for k, v in batch.items()}
    imgs = _images(B, hw, 3)
    ref = _run(_engine(1, state, B, hw, cfg, dropout=0.0), d, imgs, "inline")
    e = _engine(1, state, B, hw, cfg, dropout=0.0)
    full = _cu_set(e.lib, torch.cuda.current_stream())
    assert len(full) == 256 and len(set(x for x, *_ in full)) == 8
    k = 8
    e.set_cu_masks(chain=StepEngine.cu_mask(0, 32 - k), side=StepEngine.cu_mask(0, 32 - k), conv=StepEngine.cu_mask(32 - k, 32), chain_cus=8 * (32 - k))
    chain = _cu_set(e.lib, e.chain_stream())
    assert len(chain) == 8 * (32 - k)
    per_xcd = {}
    for x, *_ in chain:
        per_xcd[x] = per_xcd.get(x, 0) + 1
    assert sorted(per_xcd.values()) == [32 - k] * 8, per_xcd
    conv = _cu_set(e.lib, e.conv_stream())
    assert len(conv) == 8 * k and not (conv & chain) and (conv | chain) == full
    with torch.cuda.stream(e.chain_stream()):
        got = _run(e, d, imgs, "before")
    for i, (r, g) in enumerate(zip(ref, got)):
        assert float((r[0] - g[0]).abs().max()) < 1e-4 and (i < len(ref) - 1 or torch.equal(r[2], g[2])), i      # the conv stack does not depend on the planner
        assert float((r[1] - g[1]).abs().max()) < 2e-2 * float(r[1].abs().max()), (i, r[1].tolist(), g[1].tolist())
    e.set_cu_masks(None, None, None, 0)
    assert _cu_set(e.lib, e.chain_stream()) == full
