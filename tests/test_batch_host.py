"""Batch producer, CPU side: pins oracle/batch_oracle.py to the fixtures the reference's own IQDataset + collate_fn
produced (tests/golden/make_batch_golden.py), and the host arithmetic of blt-vqg_amd/batch.py to the oracle and to Pillow."""
import os

import numpy as np
import pytest
import torch

from oracle import batch_oracle as BO
import bltvqg_amd.batch as PB

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "batch_rows.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN, allow_pickle=False)


def _oracle_batch(z, idx):
    samples = [BO.sample_rows(z["questions"][i], z["answers"][i], z["answer_types"][i], z["cat_word_ids"]) for i in idx]
    images = [z["images"][z["image_indices"][i]] for i in idx]
    return samples, BO.collate(samples, [int(z["image_ids"][i]) for i in idx], images)


def test_oracle_rows_and_collate_match_reference(gold):
    z = gold
    for bi in range(int(z["n_batches"])):
        idx = z["b%d_index" % bi]
        samples, c = _oracle_batch(z, idx)
        assert np.array_equal(np.array([s["posterior"] for s in samples]), z["b%d_sample_posterior" % bi])
        assert np.array_equal(np.array([s["answer"] for s in samples]), z["b%d_sample_answer" % bi])
        assert np.array_equal(np.array([s["qlength"] for s in samples]), z["b%d_sample_qlength" % bi])
        assert np.array_equal(np.array([s["alength"] for s in samples]), z["b%d_sample_alength" % bi])
        for k in ("questions", "posteriors", "answers", "answer_types", "answer_types_for_input", "qindicies", "image_ids", "images"):
            ref = z["b%d_%s" % (bi, k)]
            assert c[k].shape == ref.shape and np.array_equal(c[k], ref), (bi, k)
        assert c["posteriors"].shape[1] == 21 and c["answers"].shape[1] == 5


def test_fixture_covers_the_edge_rows(gold):
    z = gold
    q = z["questions"]
    assert (q[3] != 3).all()                      # truncated question: no <end>, nothing removed
    assert q[4, 1] == 3 and (q[4, 2:] == 0).all()  # empty question
    assert ((z["answers"] == 3).sum(1) == 0).any()  # truncated answers exist
    s = BO.sample_rows(q[3], z["answers"][3], z["answer_types"][3], z["cat_word_ids"])
    assert s["posterior"][0] == 5 and s["posterior"][2:] == [int(t) for t in q[3][1:]]


def test_collate_order_matches_oracle(gold):
    z = gold
    cw = z["cat_word_ids"]
    for bi in range(int(z["n_batches"])):
        idx = z["b%d_index" % bi]
        _, c = _oracle_batch(z, idx)
        qlen = (z["questions"][idx] != 0).sum(1)
        order, qidx = PB.collate_order(cw[z["answer_types"][idx]], qlen)
        assert np.array_equal(order, c["order"])
        assert np.array_equal(qidx, z["b%d_qindicies" % bi])
        assert np.array_equal(z["questions"][idx][order], z["b%d_questions" % bi])


def test_to_pil_bytes_matches_torch_float_to_byte():
    """ToPILImage's `pic.mul(255).byte()` (torch itself is the third-party arithmetic here): wraps modulo 256."""
    g = torch.Generator().manual_seed(0)
    x = torch.cat([torch.arange(0, 256, dtype=torch.float32), torch.rand(4096, generator=g) * 255.0,
                   torch.rand(1024, generator=g)])
    assert np.array_equal(BO.to_pil_bytes(x.numpy()), x.mul(255).byte().numpy())
    assert int(BO.to_pil_bytes(np.array([200.0], np.float32))[0]) == 56


def _resize_with_tables(u8_hwc, out):
    """Pillow's two passes driven by PB.resample_coeffs (numpy restatement of what batch_images_kernel executes)."""
    h, w = u8_hwc.shape[:2]
    hx, hn, hk = PB.resample_coeffs([w], out)
    vx, vn, vk = PB.resample_coeffs([h], out)
    src = u8_hwc.astype(np.int64)
    half = 1 << (PB.PRECISION_BITS - 1)
    tmp = np.zeros((h, out, 3), np.int64)
    for ox in range(out):
        acc = np.full((h, 3), half, np.int64)
        for x in range(int(hn[0, ox])):
            acc += src[:, hx[0, ox] + x, :] * int(hk[0, ox, x])
        tmp[:, ox, :] = np.clip(acc >> PB.PRECISION_BITS, 0, 255)
    res = np.zeros((out, out, 3), np.int64)
    for oy in range(out):
        acc = np.full((out, 3), half, np.int64)
        for y in range(int(vn[0, oy])):
            acc += tmp[vx[0, oy] + y, :, :] * int(vk[0, oy, y])
        res[oy] = np.clip(acc >> PB.PRECISION_BITS, 0, 255)
    return res.astype(np.uint8)


@pytest.mark.parametrize("h,w,out", [(224, 224, 224), (224, 168, 224), (200, 224, 224), (256, 256, 224), (448, 300, 224), (37, 91, 32),
                                      (13, 7, 32), (500, 333, 64), (1, 1, 16)])
def test_resample_coeffs_reproduce_pillow_bit_exact(h, w, out):
    from PIL import Image
    r = np.random.RandomState(h * 1000 + w)
    u8 = r.randint(0, 256, size=(h, w, 3)).astype(np.uint8)
    ref = np.asarray(Image.fromarray(u8, mode="RGB").resize((out, out), Image.BILINEAR))
    assert np.array_equal(_resize_with_tables(u8, out), ref)


def test_resample_coeffs_batched_equals_single():
    sizes = [224, 168, 300, 17]
    bx, bn, bk = PB.resample_coeffs(sizes, 224)
    for i, s in enumerate(sizes):
        x, n, k = PB.resample_coeffs([s], 224)
        assert np.array_equal(bx[i], x[0]) and np.array_equal(bn[i], n[0])
        assert np.array_equal(bk[i, :, :k.shape[2]], k[0]) and not bk[i, :, k.shape[2]:].any()
    assert (bk.sum(-1) > 0).all()


def test_crop_boxes_reference_setting_is_the_whole_image():
    """train_iq.py:267-269: scale=(1.0,1.2) on the stored square image can only ever yield the full 224x224 box."""
    g = torch.Generator().manual_seed(3)
    b = PB.crop_boxes(4096, 224, 224, g)
    assert (b == np.array([0, 0, 224, 224], np.int32)).all()


def test_crop_boxes_general_scale_against_oracle_attempts():
    g = torch.Generator().manual_seed(5)
    g2 = torch.Generator().manual_seed(5)
    n, H, W = 512, 200, 260
    b = PB.crop_boxes(n, H, W, g, scale=(0.08, 1.0))
    u = torch.rand(n, 10, 4, generator=g2, dtype=torch.float64).numpy()
    for i in range(n):
        exp = None
        for t in range(10):
            hw = BO.crop_attempt(H, W, u[i, t, 0], u[i, t, 1], scale=(0.08, 1.0))
            if hw is not None:
                exp = (hw, t)
                break
        if exp is None:
            assert tuple(b[i]) == BO.crop_params_fallback(H, W)
        else:
            assert (int(b[i, 2]), int(b[i, 3])) == exp[0]
        top, left, h, w = (int(v) for v in b[i])
        assert 0 <= top and top + h <= H and 0 <= left and left + w <= W
    assert BO.crop_params_fallback(100, 400) == (0, 133, 100, 133)
    assert BO.crop_params_fallback(400, 100) == (133, 0, 133, 100)
    bf = PB.crop_boxes(4, 100, 400, torch.Generator().manual_seed(0), scale=(5.0, 6.0))
    assert (bf == np.array([0, 133, 100, 133], np.int32)).all()
