"""Fixture generator for the batch producer — runs ONLY in the build container (needs /root/reference).

Runs the reference's own `IQDataset.__getitem__` and `collate_fn` (`utils/data_loader.py:45-129,132-175`) on a small synthetic
store and records inputs + outputs in `batch_rows.npz`.  Offline shims, none of which restate reference logic:
  * `h5py` is absent: a stand-in module whose `File(path)` hands back the in-memory numpy tables of the synthetic store;
  * `data_loader.py:13-14` unpickles `vocab.pkl` from the working directory at import: this script writes that file itself (an
    instance of the reference's `Vocabulary`, `utils/train_utils.py`), together with `data/processed/cat2name.json`
    (`data_loader.py:42`), in a scratch directory and imports the module by file path from there;
  * `torchtext` stub for `utils/train_utils.py:6`.
No reference source is copied.

    python tests/golden/make_batch_golden.py
"""
import importlib.util
import json
import os
import pickle
import shutil
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"

V, NQ, NI, IM, NCAT = 60, 96, 12, 8, 5
PAD, SOQ, SOR, EOS, UNK, POS = 0, 1, 2, 3, 4, 5


def synthetic_store(seed=7):
    """Tables with the shapes/dtypes `utils/store_dataset.py:75-87` writes, including the rows its truncation produces
    (`utils/vocab.py:33-34`: a 20-token question or a 4-token answer has no <end>)."""
    r = np.random.RandomState(seed)
    questions = np.zeros((NQ, 20), np.int32)
    answers = np.zeros((NQ, 4), np.int32)
    first_word = 6 + NCAT
    for q in range(NQ):
        n = int(r.randint(1, 22))                      # words; > 18 means truncated
        row = [SOQ] + list(r.randint(first_word, V, size=n)) + [EOS]
        row = row[:20]
        questions[q, :len(row)] = row
        m = int(r.randint(1, 5))
        arow = ([SOQ] + list(r.randint(first_word, V, size=m)) + [EOS])[:4]
        answers[q, :len(arow)] = arow
    questions[3, :] = [SOQ] + [first_word] * 19        # full row, no <end>
    questions[4, :] = 0
    questions[4, :2] = [SOQ, EOS]                      # empty question
    answer_types = r.randint(0, NCAT, size=NQ).astype(np.int32)
    image_indices = r.randint(0, NI, size=NQ).astype(np.int32)
    image_ids = (100000 + r.permutation(NQ)).astype(np.int32)
    images = r.randint(0, 256, size=(NI, IM, IM, 3)).astype(np.float32)
    return dict(questions=questions, answers=answers, answer_types=answer_types, image_indices=image_indices,
                image_ids=image_ids, images=images)


def main():
    os.makedirs(os.path.join(ROOT, "scratch"), exist_ok=True)
    work = tempfile.mkdtemp(dir=os.path.join(ROOT, "scratch"))
    cwd = os.getcwd()
    try:
        store = synthetic_store()
        sys.modules["torchtext"] = types.ModuleType("torchtext")
        spec = importlib.util.spec_from_file_location("ref_train_utils", os.path.join(REF, "utils/train_utils.py"))
        tu = importlib.util.module_from_spec(spec)
        sys.modules["ref_train_utils"] = tu
        spec.loader.exec_module(tu)
        vocab = tu.Vocabulary()
        cat_names = ["cat_%c" % c for c in "edcba"][:NCAT]          # deliberately not sorted: data_loader.py:42 sorts
        for name in sorted(cat_names):
            vocab.add_word(name)
        i = 0
        while len(vocab) < V:
            vocab.add_word("w%d" % i)
            i += 1
        os.makedirs(os.path.join(work, "data/processed"))
        with open(os.path.join(work, "vocab.pkl"), "wb") as f:
            pickle.dump(vocab, f)
        with open(os.path.join(work, "data/processed/cat2name.json"), "w") as f:
            json.dump(cat_names, f)
        h5 = types.ModuleType("h5py")
        h5.File = lambda path, mode="r": store
        sys.modules["h5py"] = h5
        os.chdir(work)
        spec = importlib.util.spec_from_file_location("ref_data_loader", os.path.join(REF, "utils/data_loader.py"))
        dl = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(dl)

        ds = dl.IQDataset("store", transform=lambda im: torch.from_numpy(np.ascontiguousarray(im)))
        assert len(ds) == NQ
        out = {k: v for k, v in store.items()}
        out["cat_names"] = np.array(cat_names)
        out["cat_word_ids"] = np.array([vocab.word2idx[n] for n in sorted(cat_names)], np.int32)
        r = np.random.RandomState(11)
        batches = [np.array([3, 4, 0, 1, 2, 5, 6, 7]), r.permutation(NQ)[:32], r.permutation(NQ)[:17], np.arange(NQ)]
        for bi, idx in enumerate(batches):
            samples = [ds[int(i)] for i in idx]
            # per-sample rows before collation (data_loader.py:127-129)
            out["b%d_index" % bi] = idx.astype(np.int64)
            out["b%d_sample_posterior" % bi] = np.stack([s[3].numpy() for s in samples])
            out["b%d_sample_answer" % bi] = np.stack([s[4].numpy() for s in samples])
            out["b%d_sample_qlength" % bi] = np.array([s[7] for s in samples], np.int64)
            out["b%d_sample_alength" % bi] = np.array([s[8] for s in samples], np.int64)
            batch = dl.collate_fn(samples)
            assert list(batch.keys()) == ["images", "image_ids", "questions", "posteriors", "answers", "answer_types",
                                          "answer_types_for_input", "qindicies"]
            for k, v in batch.items():
                out["b%d_%s" % (bi, k)] = np.asarray(v) if k == "image_ids" else v.numpy()
        out["n_batches"] = np.array(len(batches))
        np.savez_compressed(os.path.join(HERE, "batch_rows.npz"), **out)
        print("wrote batch_rows.npz:", {k: v.shape for k, v in out.items() if k.startswith("b1_")})
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
