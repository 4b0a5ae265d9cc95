"""Pins oracle/iq_oracle.py against the fixtures produced by the real reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from helpers import load_golden, oracle_run, rel_err
from oracle import iq_oracle as O


@pytest.mark.parametrize("name", ["tiny", "tiny2"])
@pytest.mark.parametrize("phase2", [False, True])
def test_oracle_matches_reference_full(name, phase2):
    z, cfg, state, batch = load_golden(name)
    tag = "p2" if phase2 else "p1"
    r = oracle_run(cfg, state, batch, phase2, kliter=int(z[tag + ".kliter"]))
    assert rel_err(r["out"], z[tag + ".output"]) < 2e-6
    assert np.array_equal(r["out"].argmax(-1).numpy().astype(np.int32), z[tag + ".argmax"])   # bit-exact token ids
    assert rel_err(r["feats"], z[tag + ".feats"]) < 2e-6
    assert rel_err(r["recon"], z[tag + ".recon"]) < 2e-6
    assert abs(float(r["loss"]) - float(z[tag + ".loss"])) < 1e-5
    assert abs(r["stats"]["rec"] - float(z[tag + ".loss_rec"])) < 1e-5
    assert abs(r["stats"]["img"] - float(z[tag + ".loss_img"])) < 1e-5
    if phase2:
        assert rel_err(r["z_logit"], z[tag + ".z_logit"]) < 2e-6
        assert abs(float(r["kld"]) - float(z[tag + ".kld"])) < 1e-4
        assert abs(r["stats"]["aux"] - float(z[tag + ".loss_aux"])) < 1e-5
    gkeys = [k[len(tag) + 6:] for k in z.files if k.startswith(tag + ".grad.")]
    assert set(gkeys) == set(r["grads"].keys()), set(gkeys) ^ set(r["grads"].keys())
    for k in gkeys:
        ref = z["%s.grad.%s" % (tag, k)]
        e = rel_err(r["grads"][k], ref)
        assert e < 2e-5 or float(np.abs(ref).max()) < 1e-9, (k, e)


def test_phase1_unused_parameters_get_no_grad():
    """r_encoder, latent_*, z_classifier receive no gradient before the phase switch."""
    z, cfg, state, batch = load_golden("tiny")
    r = oracle_run(cfg, state, batch, False)
    for k in r["grads"]:
        assert not k.startswith("answer_encoder.r_encoder") and not k.startswith("latent_") and "z_classifier" not in k


def test_bn_running_stats_match_reference():
    z, cfg, state, batch = load_golden("tiny")
    r = oracle_run(cfg, state, batch, False)
    n = 0
    for k in z.files:
        if k.startswith("p1.buf."):
            name = k[len("p1.buf."):]
            got = r["buffers"][name]
            assert rel_err(got.double(), z[k].astype(np.float64)) < 1e-5, name
            n += 1
    assert n > 50


@pytest.mark.parametrize("name", ["small", "big", "ref300", "ref300l4", "runsh"])
@pytest.mark.parametrize("phase2", [False, True])
def test_oracle_matches_reference_small_cfg(name, phase2):
    """BASELINE.json configs[0..1] model (2-layer d_model 256, 224x224 images) at B=8, configs[2..3] model (6-layer d_model 512,
    8 heads, F 2048) at B=4, and the reference's CLI default widths (hidden 300 = 4 heads of 75, latent 300, FFN 600; train_iq.py:315-325):
    summary fixtures produced by the reference."""
    z, cfg, state, batch = load_golden(name)
    tag = "p2" if phase2 else "p1"
    r = oracle_run(cfg, state, batch, phase2, kliter=int(z[tag + ".kliter"]))
    assert abs(float(r["loss"]) - float(z[tag + ".loss"])) < 2e-5
    assert np.array_equal(r["out"].argmax(-1).numpy().astype(np.int32), z[tag + ".argmax"])
    idx = torch.from_numpy(z[tag + ".output_idx"])
    assert rel_err(r["out"].reshape(-1)[idx], z[tag + ".output_sample"]) < 5e-6
    assert abs(float(r["out"].double().sum()) - float(z[tag + ".output_sum"])) < 1e-2
    names = [str(s) for s in z[tag + ".grad_names"]]
    norms = z[tag + ".grad_norms"]
    assert set(names) == set(r["grads"].keys())
    for n_, g in zip(names, norms):
        got = float(r["grads"][n_].double().norm())
        assert abs(got - g) <= 5e-5 * max(g, 1e-6) + 1e-9, (n_, got, g)


def test_schedules():
    assert O.noam_lr(0, 256) == 0.0
    assert abs(O.noam_lr(4000, 256) - (1 / 16) * (1 / 4001) ** 0.5) < 1e-12
    assert abs(O.noam_lr(1, 256) - (1 / 16) * 4000 ** -1.5) < 1e-15
    assert abs(O.kl_weight(0, 15000) - (np.tanh(-3) + 1)) < 1e-12
    assert O.kl_weight(15000, 15000) == 1


def test_timing_signal_is_concat_sin_cos():
    s = O.timing_signal(21, 64)
    assert s.shape == (21, 64)
    assert torch.allclose(s[0, :32], torch.zeros(32)) and torch.allclose(s[0, 32:], torch.ones(32))
    assert abs(float(s[3, 0]) - np.sin(3.0)) < 1e-6 and abs(float(s[3, 32]) - np.cos(3.0)) < 1e-6


@pytest.mark.parametrize("name", ["tiny", "tiny2"])
@pytest.mark.parametrize("phase2", [False, True])
def test_oracle_greedy_decode_matches_reference(name, phase2):
    """IQ.decode_greedy under model.eval(): token ids and top-6 indices bit-exact, probabilities to 1e-5."""
    z, cfg, state, batch = load_golden(name)
    tag = "dec2" if phase2 else "dec1"
    P = O.clone_params(state, requires_grad=False)
    toks, tidx, tval = O.decode_greedy(P, cfg, batch["images"], batch["answers"], phase2, batch["eps"], max_decode_length=12)
    assert np.array_equal(tidx.numpy().astype(np.int32), z[tag + ".top_idx"])
    assert np.array_equal(toks.numpy().astype(np.int32), z[tag + ".top_idx"][:, :, 0])
    assert np.allclose(tval.numpy(), z[tag + ".top_val"], rtol=1e-4, atol=1e-7)


def test_region_attention_oracle_definition():
    """SURVEY N4 (build-defined, parity unpinned): the oracle's region-attention pooling equals its mean pooling when the scoring vector is
    zero, is a convex combination of the projected regions otherwise, and follows the argmax region when the scores are sharp."""
    from types import SimpleNamespace
    from oracle import iq_oracle as O
    torch.manual_seed(0)
    B, R, D, H = 3, 5, 16, 8
    x = torch.randn(B, R, D)
    P = {"encoder_cnn.region_proj.weight": torch.randn(H, D) * 0.3, "encoder_cnn.region_proj.bias": torch.randn(H) * 0.1,
         "encoder_cnn.bn.weight": torch.ones(H), "encoder_cnn.bn.bias": torch.zeros(H), "encoder_cnn.bn.running_mean": torch.zeros(H),
         "encoder_cnn.bn.running_var": torch.ones(H), "encoder_cnn.bn.num_batches_tracked": torch.zeros((), dtype=torch.long)}
    mean = O.encoder_cnn(dict(P), x, train=False)
    Pa = dict(P)
    Pa["encoder_cnn.region_attn.weight"] = torch.zeros(1, H)
    assert torch.allclose(O.encoder_cnn(Pa, x, train=False), mean, atol=1e-6)
    Pa["encoder_cnn.region_attn.weight"] = torch.randn(1, H) * 50.0
    p = x @ P["encoder_cnn.region_proj.weight"].t() + P["encoder_cnn.region_proj.bias"]
    best = (torch.tanh(p) @ Pa["encoder_cnn.region_attn.weight"].t()).squeeze(-1).argmax(dim=1)
    want = p[torch.arange(B), best] / (1.0 + 1e-5) ** 0.5          # eval-mode BatchNorm with unit statistics
    assert torch.allclose(O.encoder_cnn(Pa, x, train=False), want, atol=2e-2)      # softmax of scores 50x apart is one-hot up to ~1e-2
    spec = O.iq_spec(SimpleNamespace(emb_dim=20, hidden_dim=64, latent_dim=64, pwffn_dim=128, num_layers=1, num_heads=4, vocab_size=97,
                                     num_regions=36, region_dim=256, region_pool="attention"))
    assert spec["encoder_cnn.region_attn.weight"] == (1, 64)
