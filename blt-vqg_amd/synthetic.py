"""Seeded synthetic batches with the shape/statistics of the reference's VQA batches.

Mirrors what ``utils/data_loader.py:62-84,142-175`` (reference) produces per sample:
questions ``[<start>, w.., <end>, <pad>..]`` (20), posteriors ``[<pos>, category, w.., <pad>..]`` (21),
answers ``[<start>, category, a.., <pad>..]`` (5), answer_types_for_input ``[<start>, category, <end>]`` (3).
Used by bench.py, smoke() and the tests (there is no dataset and no network on the GPU box).
"""
import torch

PAD, SOQ, SOR, EOS, UNK, POS = 0, 1, 2, 3, 4, 5
T_Q, S_POST, S_ANS, S_CAT = 20, 21, 5, 3


def make_batch(batch_size, vocab_size, latent_dim, seed=1234, image_hw=224, first_word=6):
    g = torch.Generator().manual_seed(int(seed))
    B, V = batch_size, vocab_size
    # image-like statistics: per-sample colour offset + a smooth low-frequency pattern + pixel noise, so that samples differ from
    # each other the way normalised photographs do (pure i.i.d. noise makes every sample's pooled feature nearly identical, which
    # BatchNorm1d over the batch then amplifies into a rounding-noise test)
    coarse = torch.randn(B, 3, max(image_hw // 16, 2), max(image_hw // 16, 2), generator=g)
    images = torch.nn.functional.interpolate(coarse, size=(image_hw, image_hw), mode="bilinear", align_corners=False)
    images = images + 0.5 * torch.randn(B, 3, 1, 1, generator=g) + 0.5 * torch.randn(B, 3, image_hw, image_hw, generator=g)
    images = images.contiguous()
    questions = torch.zeros(B, T_Q, dtype=torch.long)
    posteriors = torch.zeros(B, S_POST, dtype=torch.long)
    answers = torch.zeros(B, S_ANS, dtype=torch.long)
    types_in = torch.zeros(B, S_CAT, dtype=torch.long)
    n_words = torch.randint(3, 18, (B,), generator=g)
    if B >= 2:
        n_words[0] = 18          # no padding at all: [<start>, 18 words, <end>]
        n_words[1] = 1           # nearly all padding
    ncat = min(16, max(1, V - first_word))
    cats = torch.randint(first_word, first_word + ncat, (B,), generator=g)
    n_ans = torch.randint(1, 3, (B,), generator=g)
    answer_types = cats.clone()
    for b in range(B):
        n = int(n_words[b])
        w = torch.randint(first_word, V, (n,), generator=g)
        questions[b, 0] = SOQ
        questions[b, 1:1 + n] = w
        questions[b, 1 + n] = EOS
        posteriors[b, 0] = POS
        posteriors[b, 1] = cats[b]
        posteriors[b, 2:2 + n] = w
        m = int(n_ans[b])
        answers[b, 0] = SOQ
        answers[b, 1] = cats[b]
        answers[b, 2:2 + m] = torch.randint(first_word, V, (m,), generator=g)
        types_in[b] = torch.tensor([SOQ, int(cats[b]), EOS])
    eps = torch.randn(B, latent_dim, generator=g)
    # key order = reference collate_fn dict order (utils/data_loader.py:175)
    return {
        "images": images,
        "image_ids": torch.arange(B),
        "questions": questions,
        "posteriors": posteriors,
        "answers": answers,
        "answer_types": answer_types,
        "answer_types_for_input": types_in,
        "qindicies": torch.arange(B),
        "eps": eps,
    }
