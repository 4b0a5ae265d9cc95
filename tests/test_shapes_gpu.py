"""Odd shapes through the whole engine against the CPU oracle (fp32 anchor): batch sizes that are not multiples of anything, image
sizes other than 224, odd vocabulary sizes, the 3-token "cat" context (`data_loader.py:81`), d_head 32/64, F > 2Z, Z > H."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import oracle_run, rel_err
from synth import synth_state

pytestmark = pytest.mark.gpu

CASES = {
    # name: (H, F, Z, L, heads, E, V, B, image, S_a, phase2)
    "b5_img96_v1003": (128, 256, 64, 2, 2, 52, 1003, 5, 96, 5, True),
    "cat_context": (64, 128, 64, 1, 4, 20, 97, 4, 64, 3, True),
    "b3_img160_l3_h8": (256, 512, 128, 3, 8, 100, 501, 3, 160, 5, True),
    "phase1_b7_img128": (128, 384, 96, 2, 4, 36, 333, 7, 128, 5, False),
    "wide_ffn_z_gt_h": (64, 512, 128, 1, 2, 24, 211, 6, 64, 5, True),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_engine_fp32_matches_oracle_on_odd_shapes(name):
    from oracle import iq_oracle as O
    import bltvqg_amd.synthetic as synthetic
    from bltvqg_amd.engine import StepEngine, make_config
    H, F, Z, L, h, E, V, B, hw, Sa, phase2 = CASES[name]
    cfg = SimpleNamespace(emb_dim=E, hidden_dim=H, latent_dim=Z, pwffn_dim=F, num_layers=L, num_heads=h, vocab_size=V)
    state = synth_state(O.iq_spec(cfg), seed=31)
    batch = synthetic.make_batch(B, V, Z, seed=31, image_hw=hw)
    if Sa == 3:
        batch["answers"] = batch["answer_types_for_input"]
    ref = oracle_run(cfg, state, batch, phase2, kliter=3000)
    e = StepEngine(make_config(B, H, F, Z, E, L, h, V, len_context=Sa, image_hw=(hw, hw), dtype=0, attention_dropout=0.0, relu_dropout=0.0))
    e.allocate()
    e.load_state(state)
    dev = "cuda"
    e.forward(batch["images"].to(dev), batch["answers"].to(dev), batch["posteriors"].to(dev), batch["questions"].to(dev),
              batch["eps"].to(dev) if phase2 else None, phase2, 0)
    out = e.read(0).cpu()
    kl_w = O.kl_weight(3000, 15000)
    e.loss_backward(kl_w)
    st = e.stats()
    total = st["rec"] + 0.1 * st["img"] + (0.5 * kl_w * st["kld"] + st["aux"] if phase2 else 0.0)
    assert rel_err(out, ref["out"]) < 2e-4
    assert np.array_equal(out.argmax(-1).numpy(), ref["out"].argmax(-1).numpy())
    assert abs(total - float(ref["loss"])) < 1e-3
    for n, g in ref["grads"].items():
        if float(g.abs().max()) < 1e-7 or n.endswith("fc.bias"):
            continue
        assert rel_err(e.grad_view(n).cpu(), g) < 3e-3, n
