"""Work the reference does and throws away, not done here — proven observably identical on the GPU.

1. Phase 1 (`latent_transformer` off, train_iq.py:108-111): the reference runs r_encoder and drops its output
   (models/encoder_transformer.py:23-25,33-35).  The engine skips that stack (and the embedding of its token rows); debug key 23 bit 0
   runs it anyway.  Outputs, losses, every gradient and the BatchNorm statistics must be bit-identical either way.
2. Phase 2: only row 0 of every sample of the posterior encoder's output is read (`response_encoder_outputs[:, 0]`,
   encoder_transformer.py:35).  Everything behind the top layer's attention core is row-wise, so the engine runs it (forward and
   backward) on those B rows; debug key 23 bit 1 runs all rows.  Forward values bit-identical, gradients to summation order.
"""
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu


def _engine(cfg, B, hw, dtype):
    from bltvqg_amd.engine import StepEngine, make_config
    c = make_config(B, cfg.hidden_dim, cfg.pwffn_dim, cfg.latent_dim, cfg.emb_dim, cfg.num_layers, cfg.num_heads, cfg.vocab_size,
                    image_hw=(hw, hw), dtype=dtype, attention_dropout=0.0, relu_dropout=0.0)
    e = StepEngine(c)
    e.allocate()
    return e


def _step(e, state, d, phase2, key23):
    e.lib.bltvqg_debug_set(23, key23)
    try:
        e.load_state(state)
        e.forward(d["images"], d["answers"], d["posteriors"], d["questions"], d["eps"] if phase2 else None, phase2, 3)
        out = {"output": e.read(0).clone(), "feats": e.read(2).clone(), "recon": e.read(3).clone()}
        if phase2:
            out["z_logit"] = e.read(1).clone()
        e.loss_backward(0.4)
        out["stats"] = e.read(4).clone()
        out["grad"] = e.flat_grad.clone()
        out["frozen"] = e.flat_frozen.clone()
        torch.cuda.synchronize()
    finally:
        e.lib.bltvqg_debug_set(23, 0)
    return out


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("name", ["tiny", "tiny2"])
def test_phase1_step_without_the_posterior_encoder_is_bit_identical(name, dtype):
    z, cfg, state, batch = load_golden(name)
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    d = {k: v.cuda() for k, v in batch.items()}
    e = _engine(cfg, B, hw, dtype)
    ref = _step(e, state, d, False, 1)       # the reference's form: r_encoder runs, its output is dropped
    got = _step(e, state, d, False, 0)       # shipped form
    for k in ("output", "feats", "recon", "frozen"):
        assert torch.equal(ref[k], got[k]), k
    # (the loss statistics are float-atomic sums: equal up to the order the atomics land in, as between any two runs of one form)
    assert float((ref["stats"][:4] - got["stats"][:4]).abs().max()) <= 1e-5
    # gradients: identical launches on identical operands (only float-atomic order inside a launch may differ, as between any two runs)
    scale = float(ref["grad"].abs().max())
    assert float((ref["grad"] - got["grad"]).abs().max()) <= 1e-6 * scale
    # the phase-2-only parameters got no gradient in either form
    assert float(got["grad"][e.late_offset:].abs().max()) == 0.0


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("name", ["tiny", "tiny2"])
def test_posterior_top_layer_on_row0_matches_all_rows(name, dtype):
    z, cfg, state, batch = load_golden(name)
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    d = {k: v.cuda() for k, v in batch.items()}
    e = _engine(cfg, B, hw, dtype)
    ref = _step(e, state, d, True, 2)        # all rows of the top layer
    got = _step(e, state, d, True, 0)        # shipped form: B rows
    for k in ("output", "z_logit", "feats", "recon"):
        assert torch.equal(ref[k], got[k]), k
    assert float((ref["stats"][:4] - got["stats"][:4]).abs().max()) <= 1e-6 * float(ref["stats"][:4].abs().max())
    # gradients: the rows left out carried exact zeros; what differs is the order in which the non-zero terms are summed
    tol = 2e-5 if dtype == 0 else 2e-3
    worst = 0.0
    for n, info in e.train_info.items():
        a = ref["grad"][info.offset:info.offset + info.numel]
        b = got["grad"][info.offset:info.offset + info.numel]
        den = float(a.norm())
        if den < 1e-12:
            assert float(b.norm()) < 1e-10, n
            continue
        err = float((a - b).norm()) / den
        worst = max(worst, err)
        assert err < tol, (n, err)
    print("worst relative gradient difference", worst)
