"""Batch producer on the GPU: the HIP gather/transform kernels, through the C-ABI, against the CPU oracle
(oracle/batch_oracle.py, pinned to the reference's own IQDataset + collate_fn outputs and to Pillow) — everything bit-exact."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import batch_oracle as BO

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch_rows.npz")


def _store_from(z):
    from bltvqg_amd.batch import IQStore
    return IQStore(z["questions"], z["answers"], z["answer_types"], z["image_indices"], z["images"], z["image_ids"], z["cat_word_ids"])


def _synthetic_store(n_rows, n_images, S, V=200, n_cat=16, seed=0):
    from bltvqg_amd.batch import IQStore
    r = np.random.RandomState(seed)
    q = np.zeros((n_rows, 20), np.int32)
    a = np.zeros((n_rows, 4), np.int32)
    for i in range(n_rows):
        row = ([1] + list(r.randint(6 + n_cat, V, size=r.randint(1, 22))) + [3])[:20]
        q[i, :len(row)] = row
        arow = ([1] + list(r.randint(6 + n_cat, V, size=r.randint(1, 5))) + [3])[:4]
        a[i, :len(arow)] = arow
    images = r.randint(0, 256, size=(n_images, S, S, 3)).astype(np.float32)
    return IQStore(q, a, r.randint(0, n_cat, size=n_rows).astype(np.int32), r.randint(0, n_images, size=n_rows).astype(np.int32),
                   images, (5000 + np.arange(n_rows)).astype(np.int32), np.arange(6, 6 + n_cat, dtype=np.int32))


def test_image_store_u8_matches_oracle():
    from bltvqg_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(1)
    for n in (4, 7, 1024 * 3 + 1, 224 * 224 * 3):
        x = torch.rand(n, generator=g) * 255.0
        x[: min(n, 256)] = torch.arange(min(n, 256), dtype=torch.float32)     # the integer pixel values a real store holds
        d = x.cuda()
        out = torch.zeros(n + 3, dtype=torch.uint8, device="cuda")
        _lib.check(lib.bltvqg_image_store_u8(_lib.ptr(d), _lib.ptr(out), n, _lib.stream_ptr()))
        assert np.array_equal(out[:n].cpu().numpy(), BO.to_pil_bytes(x.numpy()))
        assert not out[n:].any()


def test_producer_reproduces_reference_batches_bit_exact():
    """The fixture batches the reference's IQDataset + collate_fn produced: token rows, ordering, ids — and the images through the
    transform (8x8 stored images, identity box: ToTensor -> ToPILImage wrap -> /255 -> Normalize)."""
    from bltvqg_amd.batch import DeviceBatchProducer, BATCH_KEYS
    z = np.load(GOLDEN, allow_pickle=False)
    p = DeviceBatchProducer(_store_from(z), out_size=8)
    for bi in range(int(z["n_batches"])):
        idx = z["b%d_index" % bi]
        b = p.batch(idx, return_u8=True)
        assert tuple(k for k in b if k != "images_u8") == BATCH_KEYS
        for k in ("questions", "posteriors", "answers", "answer_types", "answer_types_for_input", "qindicies"):
            assert b[k].dtype == torch.int64
            assert np.array_equal(b[k].cpu().numpy(), z["b%d_%s" % (bi, k)]), (bi, k)
        assert np.array_equal(np.array(b["image_ids"]), z["b%d_image_ids" % bi])
        raw = z["b%d_images" % bi]                       # the stored float images in collated order
        imgs = b["images"].cpu().numpy()
        for s in range(len(idx)):
            exp, exp_u8 = BO.transform_image(raw[s], (0, 0, 8, 8), out_size=8)
            assert np.array_equal(b["images_u8"][s].cpu().numpy(), exp_u8)
            assert np.array_equal(imgs[s], exp), (bi, s)


@pytest.mark.parametrize("S,out,scale", [(64, 32, (0.08, 1.0)), (96, 224, (0.3, 1.0)), (224, 224, (0.5, 1.0))])
def test_random_resized_crops_match_pillow_bit_exact(S, out, scale):
    from bltvqg_amd.batch import DeviceBatchProducer, crop_boxes
    store = _synthetic_store(64, 9, S, seed=S)
    p = DeviceBatchProducer(store, out_size=out, scale=scale)
    idx = np.random.RandomState(2).permutation(64)[:24]
    boxes = crop_boxes(len(idx), S, S, torch.Generator().manual_seed(S), scale=scale)
    boxes[0] = (0, 0, S, S)
    boxes[1] = (S - 1, S - 1, 1, 1)                    # one-pixel crop
    boxes[2] = (0, 3, S, min(out, S - 3)) if S - 3 >= 1 else boxes[2]
    b = p.batch(idx, boxes=boxes, return_u8=True)
    # position s of the batch holds sample idx[order[s]] with boxes[order[s]]
    samples = [BO.sample_rows(store.questions[i], store.answers[i], store.answer_types[i], store.cat_word_ids) for i in idx]
    order = BO.collate(samples, list(idx), None)["order"]
    imgs, u8 = b["images"].cpu().numpy(), b["images_u8"].cpu().numpy()
    for s, o in enumerate(order):
        exp, exp_u8 = BO.transform_image(store.images[store.image_indices[idx[o]]], tuple(int(v) for v in boxes[o]), out_size=out)
        assert np.array_equal(u8[s], exp_u8), (s, boxes[o])
        assert np.array_equal(imgs[s], exp), (s, boxes[o])


def test_out_of_range_requests_are_refused_on_the_host():
    from bltvqg_amd.batch import DeviceBatchProducer
    p = DeviceBatchProducer(_synthetic_store(8, 2, 16), out_size=16)
    with pytest.raises(IndexError):
        p.batch([0, 8])
    with pytest.raises(IndexError):
        p.batch([])
    with pytest.raises(ValueError):
        p.batch([0], boxes=[(0, 0, 17, 16)])


def test_full_size_batch_feeds_the_train_step():
    """Reference setting (224x224 store, scale (1.0,1.2), batch 128): the produced dict drives TrainIQ.fused_training_step as the
    reference's loader output would; spot-check samples against the oracle and the epoch iterator's coverage."""
    from bltvqg_amd.batch import DeviceBatchProducer
    from train_iq import SyntheticVocabulary, TrainIQ
    V = 400
    store = _synthetic_store(300, 40, 224, V=V, seed=9)
    p = DeviceBatchProducer(store, out_size=224, seed=4)
    seen = np.concatenate(list(p.epoch(128)))
    assert sorted(seen.tolist()) == list(range(300))
    idx = next(p.epoch(128, drop_last=True))
    b = p.batch(idx)
    assert b["images"].shape == (128, 3, 224, 224) and b["images"].is_cuda
    samples = [BO.sample_rows(store.questions[i], store.answers[i], store.answer_types[i], store.cat_word_ids) for i in idx]
    c = BO.collate(samples, list(idx), None)
    for k in ("questions", "posteriors", "answers", "answer_types", "answer_types_for_input", "qindicies"):
        assert np.array_equal(b[k].cpu().numpy(), c[k]), k
    for s in (0, 63, 127):
        i = idx[c["order"][s]]
        exp, _ = BO.transform_image(store.images[store.image_indices[i]], (0, 0, 224, 224))
        assert np.array_equal(b["images"][s].cpu().numpy(), exp)
    args = SimpleNamespace(emb_dim=32, hidden_dim=64, latent_dim=64, pwffn_dim=128, num_layers=1, num_heads=4, device="cuda", emb_file=None,
                           root_dir=".", lr=3e-5, num_pretraining_steps=1, full_kl_step=10, kl_ceiling=0.5, aux_ceiling=1.0,
                           image_recon_lambda=0.1, batch_size=128, input_mode="ans", print_note="", precision="bf16",
                           attention_dropout=0.0, relu_dropout=0.0)
    t = TrainIQ(SyntheticVocabulary(V), args).to("cuda")
    for step in range(3):                                  # crosses the phase switch at iter == 1
        t.fused_training_step(p.batch(next(p.epoch(128, drop_last=True))))
    st = t.last_stats()
    assert np.isfinite(st["loss"]) and st["loss"] > 0


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("resize", [False, True])
def test_packed_engine_input_is_bit_identical(dtype, resize):
    """DeviceBatchProducer.batch(engine=...) fills the engine's packed stem input directly: same image feature, bit for bit, as the
    fp32 NCHW batch + the engine's own img_pack; borders stay zero; identity crops and Pillow-resampled crops."""
    from bltvqg_amd.batch import DeviceBatchProducer, crop_boxes
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.trainer import init_reference_style
    S, out, B = (80, 64, 6) if resize else (64, 64, 6)
    store = _synthetic_store(40, 7, S, V=97, n_cat=8, seed=11)
    p = DeviceBatchProducer(store, out_size=out, scale=(0.3, 1.0) if resize else (1.0, 1.2))
    e = StepEngine(make_config(B, 64, 128, 64, 20, 1, 4, 97, image_hw=(out, out), dtype=dtype, attention_dropout=0.0, relu_dropout=0.0))
    e.allocate()
    init_reference_style(e, seed=3)
    idx = np.arange(B) * 3
    boxes = crop_boxes(B, S, S, torch.Generator().manual_seed(1), scale=(0.3, 1.0)) if resize else np.tile(np.array([0, 0, 64, 64], np.int32), (B, 1))
    b = p.batch(idx, boxes=boxes)
    e.forward(b["images"], b["answers"], b["posteriors"], b["questions"], None, False, 0)
    feats_ref, out_ref = e.read(2).clone(), e.read(0).clone()
    b2 = p.batch(idx, boxes=boxes, engine=e)
    assert b2["images"] is None and torch.equal(b2["questions"], b["questions"])
    e.forward(None, b2["answers"], b2["posteriors"], b2["questions"], None, False, 0)
    assert torch.equal(e.read(2), feats_ref) and torch.equal(e.read(0), out_ref)
    with pytest.raises(ValueError):
        p.batch(idx[:3], engine=e)


def test_fit_from_producer_matches_dict_batches():
    """TrainIQ.fit_from_producer (images written into the engine's stem input) follows the same trajectory as fused_training_step on
    the producer's fp32 batches."""
    from bltvqg_amd.batch import DeviceBatchProducer
    from train_iq import SyntheticVocabulary, TrainIQ
    V, B = 200, 16
    store = _synthetic_store(64, 10, 64, V=V, seed=5)

    def make():
        args = SimpleNamespace(emb_dim=32, hidden_dim=64, latent_dim=64, pwffn_dim=128, num_layers=1, num_heads=4, device="cuda", emb_file=None,
                               root_dir=".", lr=3e-5, num_pretraining_steps=2, full_kl_step=10, kl_ceiling=0.5, aux_ceiling=1.0,
                               image_recon_lambda=0.1, batch_size=B, input_mode="ans", print_note="", precision="fp32",
                               attention_dropout=0.0, relu_dropout=0.0, seed=3)
        return TrainIQ(SyntheticVocabulary(V), args).to("cuda")

    eps = torch.randn(B, 64, generator=torch.Generator().manual_seed(0)).cuda()
    p = DeviceBatchProducer(store, out_size=64, seed=0)
    t1 = make()
    for idx in list(p.epoch(B, shuffle=False, drop_last=True))[:4]:
        b = p.batch(idx)
        b["eps"] = eps
        t1.fused_training_step(b)
    t2 = make()
    real_batch = p.batch
    p.batch = lambda idx, **kw: dict(real_batch(idx, **kw), eps=eps)         # same latent noise in both runs
    t2.fit_from_producer(p, B, max_steps=4, shuffle=False, log_every=0)
    assert t1.iter == t2.iter == 4 and t2.latent_transformer
    s1, s2 = t1.last_stats(), t2.last_stats()
    assert abs(s1["loss"] - s2["loss"]) < 1e-4 * max(1.0, abs(s1["loss"]))
    for (k, v1), (_, v2) in zip(t1.model.state_dict().items(), t2.model.state_dict().items()):
        if v1.dtype.is_floating_point and v1.numel() > 1:
            assert float((v1 - v2).abs().max()) <= 1e-4 + 2.5 * 4 * 1e-3, k      # Adam turns rounding noise of ~0 gradients into +-lr
