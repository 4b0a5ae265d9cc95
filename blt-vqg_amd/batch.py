"""Batch producer with the store resident in HBM.

The reference feeds the step from 8 host workers: `IQDataset.__getitem__` builds token rows per sample and pushes a stored float
HWC image through ToTensor -> ToPILImage -> RandomResizedCrop(224, scale (1.0,1.2)) -> ToTensor -> Normalize; `collate_fn` sorts the
samples by category and stacks them (`utils/data_loader.py:45-129,132-175`, `train_iq.py:264-272`).  At this engine's step rate
that pipeline would have to process 29 GB/s of float pixels (602 KB per sample) on the host and ship them over PCIe (a finished pinned batch
alone costs the step 9 %, `bench.py --h2d`), while 288 GB of HBM hold the whole image table once the
deterministic ToTensor -> ToPILImage round trip has been applied (bytes, 150 KB per image).  So:

  * `IQStore`              the reference's HDF5 datasets as arrays (`utils/store_dataset.py:75-87`: questions, answers, answer_types,
                           image_indices, images, image_ids) — numpy arrays, memory maps or open h5py datasets, anything sliceable;
  * `DeviceBatchProducer`  uploads the store once (images -> uint8 HWC in HBM through `bltvqg_image_store_u8`), then per step: the
                           host decides the ORDER (shuffle, collate_fn's stable sort by category, `qindicies`) and the crop boxes
                           from small host tables, and two kernels gather token rows and transformed images straight into the
                           tensors `TrainIQ.forward` takes.  Per step the host ships a few hundred bytes of indices.

Host arithmetic that must be bit-exact (Pillow's resampling weights) is done in float64 numpy here and checked against Pillow in
`tests/test_batch_host.py`; pixels never touch the host.  No CPU fallback: the kernels are the only implementation.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream_ptr

PAD, SOQ, SOR, EOS, UNK, POS = 0, 1, 2, 3, 4, 5
MEAN = (0.485, 0.456, 0.406)      # train_iq.py:271
STD = (0.229, 0.224, 0.225)       # train_iq.py:272
PRECISION_BITS = 22               # Pillow's fixed point for 8-bit channels (32 - 8 - 2)
BATCH_KEYS = ("images", "image_ids", "questions", "posteriors", "answers", "answer_types", "answer_types_for_input",
              "qindicies")        # collate_fn's dict order, data_loader.py:163


class IQStore(object):
    """The tables `utils/store_dataset.py:75-87` writes.  `cat_word_ids[k]` is the vocabulary id of the k-th SORTED category name
    (`data_loader.py:42,78-79`)."""

    def __init__(self, questions, answers, answer_types, image_indices, images, image_ids, cat_word_ids):
        self.questions = questions
        self.answers = answers
        self.answer_types = answer_types
        self.image_indices = image_indices
        self.images = images
        self.image_ids = image_ids
        self.cat_word_ids = np.asarray(cat_word_ids, np.int32)
        n = len(questions)
        if not (len(answers) == n and len(answer_types) == n and len(image_indices) == n and len(image_ids) == n):
            raise ValueError("IQStore: per-question tables differ in length")
        if images.ndim != 4 or images.shape[3] != 3 or images.shape[1] != images.shape[2]:
            raise ValueError("IQStore: images must be [n_images, S, S, 3]")

    def __len__(self):
        return len(self.questions)


def resample_coeffs(in_size, out_size):
    """Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` (bilinear, support 1) for resizing `in_size[b]` pixels to `out_size`:
    returns (first [B,out] int32, count [B,out] int32, weights [B,out,KS] int32).  float64 throughout, like Pillow's C doubles."""
    in_size = np.asarray(in_size, np.float64).reshape(-1, 1)
    B = in_size.shape[0]
    scale = in_size / float(out_size)
    fscale = np.maximum(scale, 1.0)
    support = 1.0 * fscale
    KS = int(np.max(np.ceil(support))) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64).reshape(1, -1)
    center = (xx + 0.5) * scale
    ss = 1.0 / fscale
    xmin = np.maximum(np.trunc(center - support + 0.5), 0.0)
    xmax = np.minimum(np.trunc(center + support + 0.5), in_size) - xmin
    k = np.zeros((B, out_size, KS), np.float64)
    ww = np.zeros((B, out_size), np.float64)
    for x in range(KS):                                        # sequential accumulation, the order of the C loop
        arg = np.abs((x + xmin - center + 0.5) * ss)
        w = np.where(arg < 1.0, 1.0 - arg, 0.0)
        w = np.where(x < xmax, w, 0.0)
        k[:, :, x] = w
        ww = ww + w
    nz = ww != 0.0
    k = np.where(nz[:, :, None], k / np.where(nz, ww, 1.0)[:, :, None], k)
    kk = np.trunc(0.5 + k * float(1 << PRECISION_BITS)).astype(np.int32)     # weights are >= 0 for the triangle filter
    return xmin.astype(np.int32), xmax.astype(np.int32), kk


def crop_boxes(n, height, width, generator, scale=(1.0, 1.2), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """`RandomResizedCrop.get_params` for n samples: up to 10 attempts of (area ~ U(scale)·H·W, log-uniform aspect ratio), first
    one that fits wins with a uniform position; otherwise the central crop at the nearest allowed ratio.  With the reference's
    scale=(1.0,1.2) on a square image essentially every attempt fails and the box is the whole image (train_iq.py:267-269).
    The reference draws from torch's global RNG one sample at a time inside its workers; this draws the same distributions from
    `generator`, vectorised.  Returns int32 [n,4] (top, left, h, w)."""
    area = float(height * width)
    u = torch.rand(n, 10, 4, generator=generator, dtype=torch.float64).numpy()
    target = area * (scale[0] + (scale[1] - scale[0]) * u[:, :, 0])
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    aspect = np.exp(lo + (hi - lo) * u[:, :, 1])
    w = np.round(np.sqrt(target * aspect)).astype(np.int64)
    h = np.round(np.sqrt(target / aspect)).astype(np.int64)
    fits = (w > 0) & (w <= width) & (h > 0) & (h <= height)
    first = np.argmax(fits, axis=1)
    any_fit = fits.any(axis=1)
    rows = np.arange(n)
    hw = h[rows, first], w[rows, first]
    top = np.floor(u[rows, first, 2] * (height - hw[0] + 1)).astype(np.int64)
    left = np.floor(u[rows, first, 3] * (width - hw[1] + 1)).astype(np.int64)
    # fallback: central crop
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        fw, fh = width, int(round(width / min(ratio)))
    elif in_ratio > max(ratio):
        fh, fw = height, int(round(height * max(ratio)))
    else:
        fw, fh = width, height
    out = np.empty((n, 4), np.int32)
    out[:, 0] = np.where(any_fit, top, (height - fh) // 2)
    out[:, 1] = np.where(any_fit, left, (width - fw) // 2)
    out[:, 2] = np.where(any_fit, hw[0], fh)
    out[:, 3] = np.where(any_fit, hw[1], fw)
    return out


def collate_order(cat_words, qlengths):
    """collate_fn's ordering (data_loader.py:151,160-161): stable sort by category word id, descending (`list.sort(reverse=True)`
    keeps equal keys in their original order); `qindicies` = flipped argsort of the question lengths in that order."""
    cat_words = np.asarray(cat_words)
    order = np.argsort(-cat_words.astype(np.int64), kind="stable")
    qidx = np.flip(np.argsort(np.asarray(qlengths)[order]), axis=0).copy()
    return order, qidx


class DeviceBatchProducer(object):
    def __init__(self, store, device="cuda:0", out_size=224, seed=0, scale=(1.0, 1.2), ratio=(3.0 / 4.0, 4.0 / 3.0), chunk_images=256):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.out_size = int(out_size)
        self.scale, self.ratio = scale, ratio
        self.gen = torch.Generator().manual_seed(int(seed))
        self.n_rows = len(store)
        dev = self.device
        # small per-question tables: on the host for ordering decisions, on the device for the gather kernels
        self.h_answer_types = np.asarray(store.answer_types[:], np.int32)
        self.h_image_ids = np.asarray(store.image_ids[:])
        q = np.ascontiguousarray(store.questions[:], np.int32)
        a = np.ascontiguousarray(store.answers[:], np.int32)
        self.q_len, self.a_len = q.shape[1], a.shape[1]
        self.h_qlengths = (q != 0).sum(1)                            # data_loader.py:121
        self.h_cat_words = store.cat_word_ids[np.clip(self.h_answer_types, 0, len(store.cat_word_ids) - 1)]
        self.questions = torch.from_numpy(q).to(dev)
        self.answers = torch.from_numpy(a).to(dev)
        self.answer_types = torch.from_numpy(self.h_answer_types).to(dev)
        self.image_indices = torch.from_numpy(np.ascontiguousarray(store.image_indices[:], np.int32)).to(dev)
        self.cat_word_ids = torch.from_numpy(store.cat_word_ids).to(dev)
        # image table: float HWC on disk -> bytes in HBM, streamed in chunks so that host memory never holds the table twice
        n_img, S = store.images.shape[0], store.images.shape[1]
        self.n_images, self.S = int(n_img), int(S)
        self.table = torch.empty((n_img, S, S, 3), dtype=torch.uint8, device=dev)
        for i0 in range(0, n_img, chunk_images):
            i1 = min(i0 + chunk_images, n_img)
            chunk = torch.from_numpy(np.ascontiguousarray(store.images[i0:i1], np.float32)).to(dev)
            check(self.lib.bltvqg_image_store_u8(ptr(chunk), ptr(self.table[i0:i1]), chunk.numel(), stream_ptr()), "image_store_u8")
        torch.cuda.synchronize(dev)
        self.mean_std = (ctypes.c_float * 6)(*(MEAN + STD))

    def _h2d(self, arr):
        """Small host array -> device through torch's caching pinned allocator (it keeps the staging block alive until the copy ran)."""
        t = torch.empty(arr.shape, dtype=torch.from_numpy(arr[:0]).dtype, pin_memory=True)
        t.numpy()[...] = arr
        return t.to(self.device, non_blocking=True)

    def epoch(self, batch_size, shuffle=True, drop_last=False):
        """Index batches of one epoch (DataLoader(shuffle=True) = a fresh permutation per epoch)."""
        perm = torch.randperm(self.n_rows, generator=self.gen).numpy() if shuffle else np.arange(self.n_rows)
        for i0 in range(0, self.n_rows, batch_size):
            idx = perm[i0:i0 + batch_size]
            if drop_last and len(idx) < batch_size:
                return
            yield idx

    def batch(self, indices, boxes=None, return_u8=False, engine=None):
        """One collated batch (reference dict keys and order) for the given sample indices; tensors live on the device.
        engine: a bound StepEngine of matching batch / image size — the images are then written straight into its packed stem input
        (no fp32 NCHW tensor, no img_pack launch; bit-identical) and `images` is None: call `engine.forward(None, ...)` /
        `DataParallelStep.run(None, ...)` next."""
        idx = np.asarray(indices, np.int64)
        if idx.ndim != 1 or len(idx) == 0 or idx.min() < 0 or idx.max() >= self.n_rows:
            raise IndexError("DeviceBatchProducer.batch: indices out of range")
        B, osz, dev = len(idx), self.out_size, self.device
        order, qidx = collate_order(self.h_cat_words[idx], self.h_qlengths[idx])
        if boxes is None:
            boxes = crop_boxes(B, self.S, self.S, self.gen, self.scale, self.ratio)
        boxes = np.ascontiguousarray(np.asarray(boxes, np.int32).reshape(B, 4))
        if (boxes[:, 0] < 0).any() or (boxes[:, 1] < 0).any() or (boxes[:, 2] <= 0).any() or (boxes[:, 3] <= 0).any() or \
                (boxes[:, 0] + boxes[:, 2] > self.S).any() or (boxes[:, 1] + boxes[:, 3] > self.S).any():
            raise ValueError("DeviceBatchProducer.batch: crop box outside the stored image")
        idx, boxes = idx[order], boxes[order]               # the sample at position b of the batch, after collate_fn's sort
        coeffs, KS = None, 0
        if ((boxes[:, 2] != osz) | (boxes[:, 3] != osz)).any():
            hx, hn, hk = resample_coeffs(boxes[:, 3], osz)
            vx, vn, vk = resample_coeffs(boxes[:, 2], osz)
            KS = max(hk.shape[2], vk.shape[2])
            tab = np.zeros((B, 2, osz, 2 + KS), np.int32)
            tab[:, 0, :, 0], tab[:, 0, :, 1], tab[:, 0, :, 2:2 + hk.shape[2]] = hx, hn, hk
            tab[:, 1, :, 0], tab[:, 1, :, 1], tab[:, 1, :, 2:2 + vk.shape[2]] = vx, vn, vk
            coeffs = self._h2d(tab)
        # indices and boxes travel as ONE pinned, asynchronous copy: a pageable copy drains the stream first, which stalls a step
        # that is queued behind it (measured: +0.15 ms per step for two pageable copies, nothing for this one)
        packed = np.empty(B * 6, np.int32)
        packed[:2 * B] = np.ascontiguousarray(idx).view(np.int32)
        packed[2 * B:] = boxes.reshape(-1)
        d_packed = self._h2d(packed)
        d_idx = d_packed[:2 * B].view(torch.int64)
        d_box = d_packed[2 * B:].view(B, 4)
        if engine is not None:
            c = engine.cfg
            if (c.batch, c.image_h, c.image_w, c.num_regions) != (B, osz, osz, 0) or return_u8:
                raise ValueError("DeviceBatchProducer.batch: engine shape %s does not match the batch (%d, %d, %d)" %
                                 ((c.batch, c.image_h, c.image_w), B, osz, osz))
        out = {
            "images": None if engine is not None else torch.empty((B, 3, osz, osz), dtype=torch.float32, device=dev),
            "image_ids": tuple(self.h_image_ids[idx].tolist()),
            "questions": torch.empty((B, self.q_len), dtype=torch.int64, device=dev),
            "posteriors": torch.empty((B, self.q_len + 1), dtype=torch.int64, device=dev),
            "answers": torch.empty((B, self.a_len + 1), dtype=torch.int64, device=dev),
            "answer_types": torch.empty((B,), dtype=torch.int64, device=dev),
            "answer_types_for_input": torch.empty((B, 3), dtype=torch.int64, device=dev),
            "qindicies": torch.from_numpy(qidx.astype(np.int64)),
        }
        check(self.lib.bltvqg_batch_rows(ptr(self.questions), ptr(self.answers), ptr(self.answer_types), ptr(self.cat_word_ids),
                                         int(self.cat_word_ids.numel()), self.n_rows, ptr(d_idx), B, self.q_len, self.a_len,
                                         ptr(out["questions"]), ptr(out["posteriors"]), ptr(out["answers"]), ptr(out["answer_types"]),
                                         ptr(out["answer_types_for_input"]), stream_ptr()), "batch_rows")
        if engine is not None:
            p_img, hp, wp, edt = engine.image_input()
            check(self.lib.bltvqg_batch_images_packed(ptr(self.table), self.n_images, self.S, ptr(self.image_indices), self.n_rows, ptr(d_idx),
                                                      ptr(d_box), ptr(coeffs), KS, B, osz, self.mean_std, edt, p_img, hp, wp, 3, 3,
                                                      stream_ptr()), "batch_images_packed")
            return out
        u8 = torch.empty((B, osz, osz, 3), dtype=torch.uint8, device=dev) if return_u8 else None
        check(self.lib.bltvqg_batch_images(ptr(self.table), self.n_images, self.S, ptr(self.image_indices), self.n_rows, ptr(d_idx),
                                           ptr(d_box), ptr(coeffs), KS, B, osz, self.mean_std, ptr(out["images"]), ptr(u8),
                                           stream_ptr()), "batch_images")
        if return_u8:
            out["images_u8"] = u8
        return out
