"""CPU oracle of the batch producer — TEST INFRASTRUCTURE, never imported by the product path.

Per-sample Python restatement of the reference's sample builder and collation, and of its image transform:

  * `sample_rows` / `collate`      follow `utils/data_loader.py:45-129` and `132-175`.  PINNED: `tests/golden/batch_rows.npz` holds
    the outputs of the reference's own `IQDataset.__getitem__` + `collate_fn` run in the build container
    (`tests/golden/make_batch_golden.py`), `tests/test_batch_host.py` checks this file against them bit for bit.
  * `transform_image`              follows the `transforms.Compose` at `train_iq.py:264-272`.  The transforms themselves live in
    torchvision 0.8.2 (`environment.yml:99`), which is ABSENT here and on the GPU box, so their composition is restated from the
    published algorithm (ToTensor: float ndarrays are transposed, not scaled; ToPILImage: float tensors go through
    `mul(255).byte()`; RandomResizedCrop: `resized_crop` = PIL crop + PIL `resize(BILINEAR)`; ToTensor of a PIL image: `/255`;
    Normalize: `(x - mean) / std`) on top of the REAL Pillow, which is present: **parity unpinned** for the composition, pinned to
    Pillow for the resampling arithmetic.  The reference has no test or fixture for it.
"""
import math

import numpy as np

PAD, SOQ, SOR, EOS, UNK, POS = 0, 1, 2, 3, 4, 5      # utils/train_utils.py:18-37
MEAN = (0.485, 0.456, 0.406)                        # train_iq.py:271
STD = (0.229, 0.224, 0.225)                         # train_iq.py:272


def sample_rows(question, answer, answer_type, cat_word_ids):
    """One sample's token rows (data_loader.py:59-86,115-116).  question: 20 ints, answer: 4 ints, answer_type: index into the
    SORTED category names; cat_word_ids[k] = vocabulary id of the k-th sorted category name (data_loader.py:42,78-79)."""
    posterior = [int(t) for t in question]
    posterior[0] = POS
    if EOS in posterior:                 # list.remove drops the FIRST <end>; a truncated row has none and keeps its length
        posterior.remove(EOS)
        posterior.append(PAD)
    ans = [int(t) for t in answer]
    if EOS in ans:
        ans.remove(EOS)
        ans.append(PAD)
    cat = int(cat_word_ids[int(answer_type)])
    type_for_input = [SOQ, cat, EOS]
    posterior.insert(1, cat)
    ans.insert(1, cat)
    qlength = len(question) - sum(1 for t in question if int(t) == 0)
    alength = len(ans) - sum(1 for t in ans if t == 0)
    return dict(question=[int(t) for t in question], posterior=posterior, answer=ans, answer_type=cat,
                answer_type_for_input=type_for_input, qlength=qlength, alength=alength)


def collate(samples, image_ids, images):
    """collate_fn (data_loader.py:150-163): stable sort by the category WORD id, descending; `qindicies` is the flipped argsort of
    the question lengths in the sorted order."""
    order = sorted(range(len(samples)), key=lambda i: samples[i]["answer_type"], reverse=True)
    s = [samples[i] for i in order]
    qlengths = [x["qlength"] for x in s]
    return {
        "order": np.array(order, np.int64),
        "images": None if images is None else np.stack([images[i] for i in order]),
        "image_ids": np.array([image_ids[i] for i in order]),
        "questions": np.array([x["question"] for x in s], np.int64),
        "posteriors": np.array([x["posterior"] for x in s], np.int64),
        "answers": np.array([x["answer"] for x in s], np.int64),
        "answer_types": np.array([x["answer_type"] for x in s], np.int64),
        "answer_types_for_input": np.array([x["answer_type_for_input"] for x in s], np.int64),
        "qindicies": np.flip(np.argsort(qlengths), axis=0).copy().astype(np.int64),
    }


def to_pil_bytes(image_hwc_f32):
    """ToTensor (float ndarray: transpose only) -> ToPILImage (float tensor: `pic.mul(255).byte()`): the float -> uint8 conversion
    of 255*x truncates and wraps modulo 256 (a stored pixel value 200.0 becomes 56).  Reproduced, not corrected."""
    v = np.asarray(image_hwc_f32, np.float32) * np.float32(255.0)
    return (np.trunc(v).astype(np.int64) & 255).astype(np.uint8)


def crop_params_fallback(height, width, ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """RandomResizedCrop.get_params after its 10 attempts failed: the central crop at the nearest allowed aspect ratio."""
    in_ratio = float(width) / float(height)
    if in_ratio < min(ratio):
        w = width
        h = int(round(w / min(ratio)))
    elif in_ratio > max(ratio):
        h = height
        w = int(round(h * max(ratio)))
    else:
        w, h = width, height
    return (height - h) // 2, (width - w) // 2, h, w


def crop_attempt(height, width, u_scale, u_logratio, scale=(1.0, 1.2), ratio=(3.0 / 4.0, 4.0 / 3.0)):
    """One attempt of RandomResizedCrop.get_params given its two uniform draws in [0,1): returns (h, w) or None."""
    area = height * width
    target_area = area * (scale[0] + (scale[1] - scale[0]) * u_scale)
    lo, hi = math.log(ratio[0]), math.log(ratio[1])
    aspect = math.exp(lo + (hi - lo) * u_logratio)
    w = int(round(math.sqrt(target_area * aspect)))
    h = int(round(math.sqrt(target_area / aspect)))
    if 0 < w <= width and 0 < h <= height:
        return h, w
    return None


def transform_image(image_hwc_f32, box, out_size=224):
    """The whole Compose for one stored image and one crop box (top, left, h, w): -> float32 (3, out, out)."""
    from PIL import Image
    u8 = to_pil_bytes(image_hwc_f32)
    im = Image.fromarray(u8, mode="RGB")
    top, left, h, w = box
    im = im.crop((left, top, left + w, top + h))
    im = im.resize((out_size, out_size), Image.BILINEAR)
    arr = np.asarray(im, np.uint8)
    x = arr.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    mean = np.array(MEAN, np.float32)[:, None, None]
    std = np.array(STD, np.float32)[:, None, None]
    return ((x - mean) / std).astype(np.float32), arr
