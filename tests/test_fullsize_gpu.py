"""Full-size parity at the BASELINE shapes: the fixtures pin the engine at B = 4..8; these tests run ONE oracle step
(forward + losses + backward on the GPU box's host cores, ~1 s / ~30 s / ~5 s) at the per-GPU batch sizes BASELINE.json names and hold
the HIP engine to it — configs[1] small at B=128, configs[2] big at B=256 (5 120-row GEMMs, 2 048 (batch, head) attention workgroups,
the full workspace carve), configs[4] regions at the per-GPU shard B=64.

Bars.  fp32 engine (exact-fp32 MFMA): total loss within 1e-3 absolute (north_star), argmax token ids bit-exact on every row whose
oracle top-2 margin exceeds 1e-4 (closer rows are ties at fp32 rounding level in either implementation; they are counted and must stay
under 0.5 % of the rows), sampled logits 3e-4 relative, 10+ parameter-gradient norms 3e-3 relative.  bf16 engine (the benched
dtype): loss 2 % relative, sampled logits 6 % relative (the frozen random-initialised 20-conv stack passes bf16 rounding on undamped;
DESIGN.md section 2), gradient norms 15 % relative.
Reference: models/iq.py:82-114, train_iq.py:81-103."""
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu

SHAPES = {
    "small": dict(B=128, H=256, F=512, Z=256, E=300, L=2, NH=4, V=8000, hw=224, regions=0),
    "big": dict(B=256, H=512, F=2048, Z=512, E=300, L=6, NH=8, V=8000, hw=224, regions=0),
    "regions": dict(B=64, H=512, F=2048, Z=512, E=300, L=6, NH=8, V=8000, hw=32, regions=36),
    # the one launch the reference documents (run.sh:1-10): hidden / latent 1024, FFN 2048, 6 layers, 8 heads of 128, batch 64, --input_mode cat
    "runsh": dict(B=64, H=1024, F=2048, Z=1024, E=300, L=6, NH=8, V=8000, hw=224, regions=0, cat=True),
}
GRAD_NAMES = ["decoder.output.weight", "decoder.decoder.dec.0.multi_head_attention_dec.query_linear.weight",
              "decoder.decoder.dec.0.multi_head_attention_enc_dec.key_linear.weight", "decoder.decoder.dec.0.positionwise_feed_forward.layers.0.weight",
              "decoder.decoder.dec.0.positionwise_feed_forward.layers.1.bias", "decoder.decoder.layer_norm.weight",
              "answer_encoder.encoder.enc.0.multi_head_attention.output_linear.weight", "answer_encoder.r_encoder.enc.0.positionwise_feed_forward.layers.1.weight",
              "embedding.0.weight", "embedding.1.weight", "latent_layer.mean_logvar_posterior.0.weight", "latent_projection.weight",
              "decoder.z_classifier.weight", "image_reconstructor.layers.fc0.weight", "encoder_cnn.bn.weight"]


def _case(name):
    from types import SimpleNamespace
    import bltvqg_amd.synthetic as synthetic
    from synth import synth_state
    from oracle import iq_oracle as O
    s = SHAPES[name]
    cfg = SimpleNamespace(emb_dim=s["E"], hidden_dim=s["H"], latent_dim=s["Z"], pwffn_dim=s["F"], num_layers=s["L"], num_heads=s["NH"],
                          vocab_size=s["V"], num_regions=s["regions"], region_dim=2048 if s["regions"] else 0)
    state = synth_state(O.iq_spec(cfg), seed=11)
    batch = synthetic.make_batch(s["B"], s["V"], s["Z"], seed=4321, image_hw=s["hw"])
    if s["regions"]:
        g = torch.Generator().manual_seed(99)
        batch["images"] = torch.relu(torch.randn(s["B"], s["regions"], 2048, generator=g) + 0.3 * torch.randn(s["B"], 1, 2048, generator=g)).contiguous()
    # --input_mode cat (train_iq.py:72-75): the context is the 3-token [<start>, category, <end>] row
    batch["context"] = batch["answer_types_for_input"] if s.get("cat") else batch["answers"]
    return s, cfg, state, batch


def _oracle_step(cfg, state, batch, kliter):
    import time
    from oracle import iq_oracle as O
    torch.set_num_threads(min(16, torch.get_num_threads() if torch.get_num_threads() > 0 else 16))
    t0 = time.time()
    P = O.clone_params(state)
    out, z_logit, kld, recon, _ = O.iq_forward(P, cfg, batch["images"], batch["context"], batch["posteriors"], batch["questions"], True,
                                               batch["eps"], None, 0.0, True, {})
    loss, st = O.calculate_losses(out, recon, kld, z_logit, batch["questions"], True, kliter, O.default_hp())
    loss.backward()
    gn = {n: float(P[n].grad.double().norm()) for n in GRAD_NAMES if n in P and P[n].grad is not None}
    print("oracle step: %.1f s, loss %.5f" % (time.time() - t0, float(loss)))
    return out.detach(), float(loss), st, gn


def _engine_step(s, cfg, state, batch, dtype, kl_w):
    from bltvqg_amd.engine import StepEngine, make_config
    c = make_config(s["B"], s["H"], s["F"], s["Z"], s["E"], s["L"], s["NH"], s["V"], len_context=3 if s.get("cat") else 5, image_hw=(s["hw"], s["hw"]),
                    dtype=dtype, attention_dropout=0.0, relu_dropout=0.0, num_regions=s["regions"], region_dim=2048 if s["regions"] else 0)
    e = StepEngine(c)
    e.allocate()
    e.load_state(state)
    d = {k: batch[k].cuda() for k in ("images", "context", "posteriors", "questions", "eps")}
    e.forward(d["images"], d["context"], d["posteriors"], d["questions"], d["eps"], True, 0)
    out = e.read(0).cpu()
    e.loss_backward(kl_w)
    st = e.stats()
    torch.cuda.synchronize()
    gn = {n: float(e.grad_view(n).double().norm()) for n in GRAD_NAMES if n in e.train_info}
    total = st["rec"] + 0.1 * st["img"] + 0.5 * kl_w * st["kld"] + st["aux"]
    return out, total, st, gn


@pytest.mark.parametrize("name", ["small", "big", "regions", "runsh"])
def test_full_size_step_matches_one_oracle_step(name):
    from oracle import iq_oracle as O
    s, cfg, state, batch = _case(name)
    kliter = 6000
    kl_w = O.kl_weight(kliter, 15000)
    ref_out, ref_loss, ref_st, ref_gn = _oracle_step(cfg, state, batch, kliter)
    assert len(ref_gn) >= 10
    V = s["V"]
    flat_ref = ref_out.reshape(-1, V)
    top2 = flat_ref.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    assert float((~clear).float().mean()) < 5e-3
    idx = torch.randint(0, flat_ref.numel(), (4096,), generator=torch.Generator().manual_seed(1))
    # ---- fp32 engine: the parity anchor ----
    out, total, st, gn = _engine_step(s, cfg, state, batch, 0, kl_w)
    print("%s fp32: loss %.6f vs oracle %.6f" % (name, total, ref_loss))
    assert abs(total - ref_loss) < 1e-3
    assert rel_err(out.reshape(-1)[idx], ref_out.reshape(-1)[idx]) < 3e-4
    am, ram = out.reshape(-1, V).argmax(-1), flat_ref.argmax(-1)
    assert torch.equal(am[clear], ram[clear])                                   # bit-exact token ids
    for n, g in ref_gn.items():
        if n in gn:
            assert abs(gn[n] - g) <= 3e-3 * max(g, 1e-6) + 1e-6, (n, gn[n], g)
    del out
    # ---- bf16 engine: the benched dtype ----
    out, total, st, gn = _engine_step(s, cfg, state, batch, 1, kl_w)
    err = rel_err(out.reshape(-1)[idx], ref_out.reshape(-1)[idx])
    agree = float((out.reshape(-1, V).argmax(-1)[clear] == ram[clear]).float().mean())
    worst = max(abs(gn[n] - g) / max(g, 1e-9) for n, g in ref_gn.items() if n in gn and g > 1e-6)
    print("%s bf16: loss %.5f vs %.5f, sampled logits rel %.4f, argmax agreement %.4f, worst gradient-norm rel %.4f" % (name, total, ref_loss, err, agree, worst))
    # bars = what is measured on MI355X (round 3: loss 0.003-0.04 %, sampled logits 1.0-1.9 %, gradient norms 0.2-0.5 %) with a
    # 4-5x margin, so that a regression shows: loss 0.2 %, sampled logits 4 %, gradient norms 2 %
    assert abs(total - ref_loss) < 2e-3 * abs(ref_loss)
    assert err < 4e-2
    assert worst < 0.02



@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_reference_cli_defaults_at_their_own_depth_and_batch(precision):
    """The reference's CLI defaults as a user launches them (train_iq.py:315-339: hidden 300 = 4 heads of 75, latent 300, FFN 600, 4 layers,
    batch 128, 224x224) through `TrainIQ` on the padded engine layout (blt-vqg_amd/padded.py) against ONE oracle step on the same inputs:
    fp32 engine: loss 1e-3, argmax bit-exact on every clear row, gradient norms 3e-3; bf16 engine: loss 0.2 %, sampled logits 4 %,
    gradient norms 2 %."""
    from types import SimpleNamespace
    import bltvqg_amd.synthetic as synthetic
    from synth import synth_state
    from oracle import iq_oracle as O
    from train_iq import SyntheticVocabulary, TrainIQ
    B, V = 128, 8000
    cfg = SimpleNamespace(emb_dim=300, hidden_dim=300, latent_dim=300, pwffn_dim=600, num_layers=4, num_heads=4, vocab_size=V)
    state = synth_state(O.iq_spec(cfg), seed=21)
    batch = synthetic.make_batch(B, V, 300, seed=77, image_hw=224)
    batch["context"] = batch["answers"]
    kliter = 6000
    ref_out, ref_loss, ref_st, ref_gn = _oracle_step(cfg, state, batch, kliter)
    flat_ref = ref_out.reshape(-1, V)
    top2 = flat_ref.topk(2, dim=-1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-4
    idx = torch.randint(0, flat_ref.numel(), (4096,), generator=torch.Generator().manual_seed(1))
    args = SimpleNamespace(emb_dim=300, hidden_dim=300, latent_dim=300, pwffn_dim=600, num_layers=4, num_heads=4, lr=3e-5,
                           num_pretraining_steps=0, full_kl_step=15000, kl_ceiling=0.5, aux_ceiling=1.0, image_recon_lambda=0.1, batch_size=B,
                           emb_file=None, root_dir=".", device=torch.device("cuda"), input_mode="ans", print_note="", precision=precision,
                           attention_dropout=0.0, relu_dropout=0.0)
    t = TrainIQ(SyntheticVocabulary(V), args)
    full = dict(state)
    for k in t.model.state_dict().keys():      # aliases of the shared embedding / latent layer
        base = k
        for alias in ("answer_encoder.embedding.", "decoder.embedding."):
            if k.startswith(alias):
                base = "embedding." + k[len(alias):]
        if k.startswith("answer_encoder.latent_layer."):
            base = k[len("answer_encoder."):]
        full[k] = state[base]
    t.model.load_state_dict(full)
    t = t.to("cuda")
    t.latent_transformer = True
    t.model.switch_GVT_train_mode(True)
    t.kliter = kliter
    b = {k: v.cuda() for k, v in batch.items()}
    output, z_logit, kld, recon = t(b)
    loss = t.calculate_losses(output, recon, kld, z_logit, b["questions"])[0]
    loss.backward()
    out = output.detach().cpu()
    total = float(loss)
    gn = {n: float(t.model.get_parameter(n).grad.double().norm()) for n in ref_gn}
    err = rel_err(out.reshape(-1)[idx], ref_out.reshape(-1)[idx])
    worst = max(abs(gn[n] - g) / max(g, 1e-9) for n, g in ref_gn.items() if g > 1e-6)
    agree = float((out.reshape(-1, V).argmax(-1)[clear] == flat_ref.argmax(-1)[clear]).float().mean())
    print("default300 %s: loss %.6f vs oracle %.6f, sampled logits rel %.5f, argmax agreement %.5f, worst gradient-norm rel %.5f" % (
        precision, total, ref_loss, err, agree, worst))
    if precision == "fp32":
        assert abs(total - ref_loss) < 1e-3
        assert err < 3e-4
        assert agree == 1.0
        assert worst < 3e-3
    else:
        assert abs(total - ref_loss) < 2e-3 * abs(ref_loss)
        assert err < 4e-2
        assert worst < 0.02
