"""Shared helpers for the tests: fixture loading and oracle runs (test infrastructure)."""
import os
from types import SimpleNamespace

import numpy as np
import torch

from synth import synth_state
import bltvqg_amd.synthetic as synthetic
from oracle import iq_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    m = z["meta_cfg"]
    cfg = SimpleNamespace(emb_dim=int(m[0]), hidden_dim=int(m[1]), latent_dim=int(m[2]), pwffn_dim=int(m[3]),
                          num_layers=int(m[4]), num_heads=int(m[5]), vocab_size=int(m[6]))
    B, hw, seed = int(m[7]), int(m[8]), int(m[9])
    state = synth_state(O.iq_spec(cfg), seed=seed)
    batch = synthetic.make_batch(B, cfg.vocab_size, cfg.latent_dim, seed=seed, image_hw=hw)
    # the context the fixture was produced with: `answers`, or `answer_types_for_input` for --input_mode cat fixtures (train_iq.py:72-75)
    cat = "meta_cat" in z.files and int(z["meta_cat"]) == 1
    batch["context"] = batch["answer_types_for_input"] if cat else batch["answers"]
    cfg.input_mode = "cat" if cat else "ans"
    return z, cfg, state, batch


def oracle_run(cfg, state, batch, phase2, kliter=0, hp=None, masks=None, p_drop=0.0):
    """Oracle forward + losses + backward.  Returns dict(out, z_logit, kld, feats, recon, loss, stats, grads, buffers)."""
    hp = hp or O.default_hp()
    P = O.clone_params(state)
    bufs = {}
    out, z_logit, kld, (feats, recon), extras = O.iq_forward(
        P, cfg, batch["images"], batch.get("context", batch["answers"]), batch["posteriors"], batch["questions"], phase2,
        batch["eps"], masks, p_drop, True, bufs)
    loss, stats = O.calculate_losses(out, (feats, recon), kld, z_logit, batch["questions"], phase2, kliter, hp)
    loss.backward()
    grads = {k: v.grad.detach() for k, v in P.items() if v.requires_grad and v.grad is not None}
    return dict(out=out.detach(), z_logit=None if z_logit is None else z_logit.detach(),
                kld=None if kld is None else kld.detach(), feats=feats.detach(), recon=recon.detach(),
                loss=loss.detach(), stats=stats, grads=grads, buffers=bufs, extras=extras)


def rel_err(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return float((a - b).norm() / (b.norm() + 1e-30))
