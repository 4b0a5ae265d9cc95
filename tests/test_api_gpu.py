"""GPU tests of the drop-in Python surface: models.IQ.forward under torch autograd (the reference contract: the caller computes
the losses with torch criteria and calls backward) and TrainIQ.training_step / fused_training_step, against the CPU oracle."""
from types import SimpleNamespace

import os
import numpy as np
import pytest
import torch

from helpers import load_golden, oracle_run, rel_err

pytestmark = pytest.mark.gpu


def _args(cfg, **kw):
    a = SimpleNamespace(emb_dim=cfg.emb_dim, hidden_dim=cfg.hidden_dim, latent_dim=cfg.latent_dim, pwffn_dim=cfg.pwffn_dim,
                        num_layers=cfg.num_layers, num_heads=cfg.num_heads, device="cuda", emb_file=None, root_dir=".", lr=3e-5,
                        num_pretraining_steps=12000, full_kl_step=15000, kl_ceiling=0.5, aux_ceiling=1.0, image_recon_lambda=0.1,
                        batch_size=4, input_mode="ans", print_note="", precision="fp32", attention_dropout=0.0, relu_dropout=0.0)
    a.__dict__.update(kw)
    return a


def _full_state(model, state):
    full = {}
    for k in model.state_dict():
        base = k
        for alias in ("answer_encoder.embedding.", "decoder.embedding."):
            if k.startswith(alias):
                base = "embedding." + k[len(alias):]
        if k.startswith("answer_encoder.latent_layer."):
            base = k[len("answer_encoder."):]
        full[k] = state[base]
    return full


@pytest.mark.parametrize("phase2", [False, True])
def test_iq_forward_autograd_matches_reference(phase2):
    """The reference flow: model(...) -> torch criteria -> loss.backward() -> .grad on the nn.Parameters."""
    from models import IQ
    from train_iq import SyntheticVocabulary, TrainIQ
    from oracle import iq_oracle as O
    z, cfg, state, batch = load_golden("tiny")
    tag = "p2" if phase2 else "p1"
    t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg))
    t.model.load_state_dict(_full_state(t.model, state))
    t = t.to("cuda")                     # reference flow: TrainIQ(vocab, args).to(device)  (train_iq.py:371)
    opt = t.configure_optimizers()       # created BEFORE the first forward, like Lightning does
    if phase2:
        t.latent_transformer = True
        t.model.switch_GVT_train_mode(True)
        t.kliter = int(z[tag + ".kliter"])
    b = {k: v.cuda() for k, v in batch.items()}
    output, z_logit, kld, recon = t(b)
    assert output.shape == (4, 20, cfg.vocab_size) and (z_logit is None) == (not phase2)
    assert rel_err(output.detach().cpu(), z[tag + ".output"]) < 2e-4
    assert np.array_equal(output.argmax(-1).cpu().numpy().astype(np.int32), z[tag + ".argmax"])
    loss, rec, img, ppl, kl, aux, elbo = t.calculate_losses(output, recon, kld, z_logit, b["questions"])
    assert abs(float(loss) - float(z[tag + ".loss"])) < 1e-3
    assert abs(rec - float(z[tag + ".loss_rec"])) < 1e-4 and abs(img - float(z[tag + ".loss_img"])) < 1e-4
    loss.backward()
    n = 0
    for k in z.files:
        if not k.startswith(tag + ".grad."):
            continue
        name = k[len(tag) + 6:]
        ref = torch.from_numpy(z[k])
        if name == "encoder_cnn.cnn.fc.bias" or float(ref.abs().max()) < 1e-7:
            continue
        g = t.model.get_parameter(name).grad
        assert g is not None, name
        assert rel_err(g.cpu(), ref) < 3e-3, (name, rel_err(g.cpu(), ref))
        n += 1
    assert n > 40
    if not phase2:      # unused parameters keep grad None, exactly like the reference
        assert t.model.get_parameter("decoder.z_classifier.weight").grad is None
        assert t.model.get_parameter("answer_encoder.r_encoder.layer_norm.weight").grad is None
    # the optimizer created before the first forward still owns the live parameters
    before = t.model.get_parameter("decoder.output.weight").detach().clone()
    for gparam in opt.param_groups:
        gparam["lr"] = 1e-3
    opt.step()
    assert not torch.equal(before, t.model.get_parameter("decoder.output.weight").detach())
    assert int(t.model.state_dict()["encoder_cnn.bn.num_batches_tracked"]) == int(state["encoder_cnn.bn.num_batches_tracked"]) + 1


def test_fused_training_steps_match_oracle():
    """TrainIQ.fused_training_step (everything in the HIP engine) over the phase switch vs the oracle's Adam run."""
    from train_iq import SyntheticVocabulary, TrainIQ
    from oracle import iq_oracle as O
    import bltvqg_amd.synthetic as synthetic
    z, cfg, state, _ = load_golden("tiny")
    B, hw = 4, 64
    batches = [synthetic.make_batch(B, cfg.vocab_size, cfg.latent_dim, seed=300 + i, image_hw=hw) for i in range(3)]
    hp = O.default_hp(num_pretraining_steps=101)
    final, logs = O.train_steps(state, cfg, batches, hp, start_iter=100)
    t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg, num_pretraining_steps=101))
    t.model.load_state_dict(_full_state(t.model, state))
    t = t.to("cuda")
    t.iter = 100
    for i, b in enumerate(batches):
        t.fused_training_step(b)
        st = t.last_stats()
        assert abs(st["rec"] - logs[i]["rec"]) < 1e-3 and abs(st["loss"] - logs[i]["loss"]) < 2e-3, (i, st, logs[i])
        assert abs(st["grad_norm"] - logs[i]["grad_norm"]) < 2e-3 * logs[i]["grad_norm"]
    assert t.latent_transformer is True and t.kliter == 2 and t.iter == 103
    w = t.model.state_dict()["decoder.output.weight"].cpu()
    assert rel_err(w - state["decoder.output.weight"], final["decoder.output.weight"] - state["decoder.output.weight"]) < 0.1


def test_batch_shape_change_shares_parameters():
    """A second batch size gets its own engine workspace but the same flat parameter / optimiser buffers."""
    from train_iq import SyntheticVocabulary, TrainIQ
    import bltvqg_amd.synthetic as synthetic
    z, cfg, state, _ = load_golden("tiny")
    t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg)).to("cuda")
    t.fused_training_step(synthetic.make_batch(4, cfg.vocab_size, cfg.latent_dim, seed=1, image_hw=64))
    t.fused_training_step(synthetic.make_batch(2, cfg.vocab_size, cfg.latent_dim, seed=2, image_hw=64))
    e = list(t.model._engines.values())
    assert len(e) == 2 and e[0].flat_train.data_ptr() == e[1].flat_train.data_ptr()
    assert np.isfinite(t.last_stats()["loss"])


@pytest.mark.parametrize("name", ["tiny", "tiny2"])
@pytest.mark.parametrize("phase2", [False, True])
def test_decode_greedy_token_ids_bit_exact(name, phase2):
    """models.IQ.decode_greedy (HIP engine, fp32) vs the reference fixture: bit-exact token indices (north_star), top-6 too."""
    from train_iq import SyntheticVocabulary, TrainIQ
    z, cfg, state, batch = load_golden(name)
    tag = "dec2" if phase2 else "dec1"
    t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg))
    t.model.load_state_dict(_full_state(t.model, state))
    t = t.to("cuda")
    t.eval()                                   # Lightning runs validation_epoch_end (train_iq.py:159-206) under model.eval()
    t.model.switch_GVT_train_mode(phase2)
    sent, top_args, top_vals = t.model.decode_greedy(batch["images"].cuda(), batch["answers"].cuda(), max_decode_length=12,
                                                     eps=batch["eps"].cuda())
    assert top_args.shape == (batch["images"].shape[0], 13, 6) and len(sent) == batch["images"].shape[0]
    assert np.array_equal(top_args.cpu().numpy().astype(np.int32), z[tag + ".top_idx"])
    assert np.allclose(top_vals.cpu().numpy(), z[tag + ".top_val"], rtol=2e-3, atol=1e-6)
    # sentences are the argmax words up to <end>
    first = [t.model.vocab.idx2word[int(i)] for i in z[tag + ".top_idx"][0, :, 0]]
    want = ""
    for wd in first:
        if wd == "<end>":
            break
        want += wd + " "
    assert sent[0] == want
    # train-mode forward afterwards still uses batch statistics (the decode switch is local)
    t.train()
    out, _, _, _ = t.model(batch["images"].cuda(), batch["answers"].cuda(), batch["posteriors"].cuda(), batch["questions"].cuda(),
                           eps=batch["eps"].cuda())
    ptag = "p2" if phase2 else "p1"
    assert rel_err(out.detach().cpu(), z[ptag + ".output"]) < 2e-4


def test_checkpoint_roundtrip_and_lightning_layout(tmp_path):
    """TrainIQ.save_checkpoint writes the reference's (Lightning) checkpoint layout — "state_dict" with the module's
    tensors under "model.<IQ key>" — and load_checkpoint reads it back with the safe loader (weights_only=True); resuming from the
    checkpoint reproduces the uninterrupted run (parameters, Adam moments and step counters, phase flags)."""
    from train_iq import SyntheticVocabulary, TrainIQ
    import bltvqg_amd.synthetic as synthetic
    z, cfg, state, _ = load_golden("tiny")
    B, hw = 4, 64
    batches = [synthetic.make_batch(B, cfg.vocab_size, cfg.latent_dim, seed=500 + i, image_hw=hw) for i in range(4)]
    for b in batches:
        b["eps"] = torch.randn(B, cfg.latent_dim, generator=torch.Generator().manual_seed(7))

    def fresh():
        t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg, num_pretraining_steps=1))
        t.model.load_state_dict(_full_state(t.model, state))
        return t.to("cuda")

    a = fresh()
    for b in batches[:2]:
        a.fused_training_step(b)
    path = str(tmp_path / "blt.ckpt")
    a.save_checkpoint(path)
    for b in batches[2:]:
        a.fused_training_step(b)
    ref = {k: v.detach().float().cpu() for k, v in a.model.state_dict().items()}

    raw = torch.load(path, map_location="cpu", weights_only=True)          # nothing in the file needs unpickling of code
    want_keys = set(open(os.path.join(os.path.dirname(__file__), "golden", "state_keys_small.txt")).read().split())
    got_keys = set(k[len("model."):] for k in raw["state_dict"] if k.startswith("model."))
    assert len(got_keys) == len(raw["state_dict"])
    if len(want_keys) == len(got_keys):        # same layer count as the fixture: the key sets must be identical
        assert got_keys == want_keys
    assert raw["global_step"] == 2

    r = fresh()
    r.load_checkpoint(path)
    assert r.iter == 2 and r.kliter == a.kliter - 2 and r.latent_transformer is True
    for b in batches[2:]:
        r.fused_training_step(b)
    got = {k: v.detach().float().cpu() for k, v in r.model.state_dict().items()}
    lr_sum = 4e-3        # generous bound on two Noam steps at this width (Adam moves noise-level elements by +-lr)
    for k in ref:
        assert (got[k] - ref[k]).abs().max() <= 2.5 * lr_sum, k
        if ref[k].ndim == 2 and ref[k].numel() > 1000 and float((ref[k] - state.get(k, ref[k])).abs().max()) > 0 and k in state:
            assert rel_err(got[k] - state[k], ref[k] - state[k]) < 0.05, (k, rel_err(got[k] - state[k], ref[k] - state[k]))
    assert r.iter == a.iter and r.kliter == a.kliter

    # a checkpoint shaped like the reference's own (extra Lightning entries, criterion buffers absent): tensors load, the rest is ignored
    lit = {"epoch": 3, "global_step": 7, "pytorch-lightning_version": "1.1.8", "state_dict": raw["state_dict"], "optimizer_states": [],
           "lr_schedulers": []}
    p2 = str(tmp_path / "lightning.ckpt")
    torch.save(lit, p2)
    q = fresh()
    q.load_checkpoint(p2)
    assert q.iter == 7 and q.latent_transformer is True
    w = q.model.state_dict()["decoder.output.weight"].float().cpu()
    assert torch.equal(w, raw["state_dict"]["model.decoder.output.weight"])


@pytest.mark.parametrize("phase2", [False, True])
def test_eval_mode_forward_and_validation_step(phase2):
    """model.eval() (what Lightning's validation loop does, train_iq.py:133-157): dropout off and BatchNorm2d/1d from the running
    statistics, nothing updated — against the oracle's eval-mode forward (the mode the greedy-decode fixtures pin)."""
    from train_iq import SyntheticVocabulary, TrainIQ
    from oracle import iq_oracle as O
    z, cfg, state, batch = load_golden("tiny")
    t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg, attention_dropout=0.1, relu_dropout=0.1))
    t.model.load_state_dict(_full_state(t.model, state))
    t = t.to("cuda")
    if phase2:
        t.latent_transformer = True
        t.model.switch_GVT_train_mode(True)
    t.eval()
    b = {k: v.cuda() for k, v in batch.items()}
    before = {k: v.detach().clone() for k, v in t.model.state_dict().items()}
    P = O.clone_params(state, requires_grad=False)
    with torch.no_grad():
        out, z_logit, kld, (feats, recon), _ = O.iq_forward(P, cfg, batch["images"], batch["answers"], batch["posteriors"], batch["questions"],
                                                           phase2, batch["eps"], None, 0.0, False, None)
        ref_loss, ref_stats = O.calculate_losses(out, (feats, recon), kld, z_logit, batch["questions"], phase2, 0, O.default_hp())
    output, zl, k, image_recon = t(b)
    assert rel_err(output.cpu(), out) < 2e-4
    assert rel_err(image_recon[0].cpu(), feats) < 2e-4
    assert np.array_equal(output.argmax(-1).cpu().numpy(), out.argmax(-1).numpy())
    t.validation_step(b, 0)
    t.validation_step(b, 1)
    assert len(t.val_metrics["loss"]) == 2 and t.val_metrics["loss"][0] == t.val_metrics["loss"][1]      # no dropout noise
    assert abs(t.val_metrics["rec"][0] - ref_stats["rec"]) < 1e-4
    assert abs(t.val_metrics["loss"][0] - float(ref_loss)) < 1e-3 * max(1.0, abs(float(ref_loss)) * 0.05)      # phase 2: the KL term of this synthetic state is ~1e12
    assert "val_loss" in t.logged and "val_elbo" in t.logged
    after = t.model.state_dict()
    for kk, v in before.items():
        assert torch.equal(after[kk], v), kk                       # running statistics, num_batches_tracked, parameters untouched
    with pytest.raises(RuntimeError):                              # training through an eval-mode forward is refused, not silently wrong
        t(b)[0].sum().backward()
    t.train()
    t(b)[0].sum().backward()                                       # and train mode still works on the same model
    assert t.model.get_parameter("decoder.output.weight").grad is not None


def test_backward_refuses_stale_activations():
    """the engine keeps one set of saved activations per batch shape; a second forward of the same shape before backward (or a
    second backward) must raise instead of silently differentiating the wrong forward."""
    from train_iq import SyntheticVocabulary, TrainIQ
    z, cfg, state, batch = load_golden("tiny")
    t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg))
    t.model.load_state_dict(_full_state(t.model, state))
    t = t.to("cuda")
    b = {k: v.cuda() for k, v in batch.items()}
    out1, _, _, _ = t(b)
    out2, _, _, _ = t(b)                       # same shape: the engine now holds the activations of THIS forward
    with pytest.raises(RuntimeError, match="later forward"):
        out1.sum().backward()
    out2.sum().backward()                      # the latest forward differentiates fine ...
    out3, _, _, _ = t(b)
    out3.sum().backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="already consumed|later forward"):
        out3.sum().backward()                  # ... once


def test_iq_forward_reports_out_of_range_ids():
    from bltvqg_amd import _lib
    from train_iq import SyntheticVocabulary, TrainIQ
    z, cfg, state, batch = load_golden("tiny")
    t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg)).to("cuda")
    b = {k: v.cuda() for k, v in batch.items()}
    b["questions"] = b["questions"].clone()
    b["questions"][0, 2] = cfg.vocab_size + 3
    with pytest.raises(_lib.HipError, match="token id"):
        t(b)


def test_fused_step_then_torch_optimizer_never_reads_a_stale_weight_shadow():
    """the fused driver lets the engine's Adam keep the bf16 weight shadows current (trust_shadows); TrainIQ shares that engine
    with the autograd path, whose parameters a TORCH optimiser updates without telling the engine.  After fused step -> autograd step +
    torch Adam -> fused forward, the forward must see the torch optimiser's weights (bit-identical to a forced rebuild from fp32)."""
    from train_iq import SyntheticVocabulary, TrainIQ
    z, cfg, state, batch = load_golden("tiny")
    t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg, precision="bf16", num_pretraining_steps=0))
    t.model.load_state_dict(_full_state(t.model, state))
    t = t.to("cuda")
    b = {k: v.cuda() for k, v in batch.items()}
    t.fused_training_step(b)                              # engine Adam: writes the bf16 shadow itself, shadow generation == parameter generation
    t.last_stats()
    eng = t._last_engine
    opt = torch.optim.Adam(t.parameters(), lr=5e-2)       # a step large enough to change every bf16 weight
    output, z_logit, kld, recon = t(b)
    loss = t.calculate_losses(output, recon, kld, z_logit, b["questions"])[0]
    opt.zero_grad()
    loss.backward()
    opt.step()
    torch.cuda.synchronize()

    def fused_forward():
        eng.trust_shadows(True)                           # what DataParallelStep.run promises at the top of every fused step
        eng.forward(b["images"], b["answers"], b["posteriors"], b["questions"], b["eps"], True, 77)
        return eng.read(0).clone()
    got = fused_forward()
    eng.params_changed()                                  # forced rebuild of every shadow from the fp32 parameters
    want = fused_forward()
    assert torch.equal(got, want)
    assert float((got - output.detach()).abs().max()) > 1e-3      # and the torch step really moved the model


@pytest.mark.parametrize("fixture", ["ref300", "ref300l4"])
@pytest.mark.parametrize("phase2", [False, True])
def test_reference_default_widths_match_the_reference(phase2, fixture):
    """The reference's CLI default widths (train_iq.py:315-325: hidden 300 = 4 heads of 75, latent 300, FFN 600) through the padded engine
    layout (blt-vqg_amd/padded.py), fp32 engine, against the fixtures the reference itself produced (tests/golden/make_golden.py ref300: one
    layer; ref300l4: the CLI's own depth, 4 layers): loss within 1e-3 (north_star), argmax token ids bit-exact, gradients of the
    reference-shaped parameters <= 3e-3."""
    from train_iq import SyntheticVocabulary, TrainIQ
    z, cfg, state, batch = load_golden(fixture)
    assert (cfg.hidden_dim, cfg.latent_dim, cfg.pwffn_dim, cfg.num_heads) == (300, 300, 600, 4)
    tag = "p2" if phase2 else "p1"
    t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg))
    t.model.load_state_dict(_full_state(t.model, state))
    t = t.to("cuda")
    if phase2:
        t.latent_transformer = True
        t.model.switch_GVT_train_mode(True)
        t.kliter = int(z[tag + ".kliter"])
    b = {k: v.cuda() for k, v in batch.items()}
    output, z_logit, kld, recon = t(b)
    assert output.shape == (4, 20, cfg.vocab_size) and recon[0].shape == (4, 300) and recon[1].shape == (4, 300)
    idx = torch.from_numpy(z[tag + ".output_idx"]).cuda()
    assert rel_err(output.detach().reshape(-1)[idx].cpu(), z[tag + ".output_sample"]) < 2e-4
    assert np.array_equal(output.argmax(-1).cpu().numpy().astype(np.int32), z[tag + ".argmax"])
    assert rel_err(recon[0].detach().cpu(), z[tag + ".feats"]) < 2e-4 and rel_err(recon[1].detach().cpu(), z[tag + ".recon"]) < 2e-4
    loss = t.calculate_losses(output, recon, kld, z_logit, b["questions"])[0]
    assert abs(float(loss) - float(z[tag + ".loss"])) < 1e-3, (float(loss), float(z[tag + ".loss"]))
    loss.backward()
    names = [str(s_) for s_ in z[tag + ".grad_names"]]
    norms = z[tag + ".grad_norms"]
    class _P(object):
        def __getitem__(self, n):
            return t.model.get_parameter(n)
    params = _P()
    for n_, g in zip(names, norms):
        got = float(params[n_].grad.double().norm())
        assert abs(got - g) <= 3e-3 * max(g, 1e-6) + 2e-6, (n_, got, g)      # (+ noise floor: the bias in front of BatchNorm1d has an exactly-zero gradient)
    for k in z.files:
        if k.startswith(tag + ".grad.") and not k.endswith(("grad_names", "grad_norms")):
            n_ = k[len(tag) + 6:]
            assert rel_err(params[n_].grad.cpu(), z[k]) < 3e-3, n_
    if not phase2:          # BatchNorm running statistics reach the reference-shaped buffers
        for k in z.files:
            if k.startswith("p1.buf.") and "num_batches" not in k:
                assert rel_err(t.model.state_dict()[k[7:]].cpu(), z[k]) < 1e-4, k


@pytest.mark.parametrize("phase2", [False, True])
def test_reference_default_widths_greedy_decode_token_ids_bit_exact(phase2):
    """IQ.decode_greedy at the reference's default widths (padded engine layout) against the reference's own decode of the ref300 fixture:
    bit-exact token ids and top-6 indices, top-6 probabilities within 2e-3."""
    from train_iq import SyntheticVocabulary, TrainIQ
    z, cfg, state, batch = load_golden("ref300")
    tag = "dec2" if phase2 else "dec1"
    t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg))
    t.model.load_state_dict(_full_state(t.model, state))
    t = t.to("cuda")
    t.eval()
    t.model.switch_GVT_train_mode(phase2)
    sent, top_args, top_vals = t.model.decode_greedy(batch["images"].cuda(), batch["answers"].cuda(), max_decode_length=12, eps=batch["eps"].cuda())
    assert top_args.shape == (4, 13, 6) and len(sent) == 4
    assert np.array_equal(top_args.cpu().numpy().astype(np.int32), z[tag + ".top_idx"])
    assert np.allclose(top_vals.cpu().numpy(), z[tag + ".top_val"], rtol=2e-3, atol=1e-6)


def test_reference_default_widths_fused_steps_and_checkpoint(tmp_path):
    """Fused training steps on the padded engine: the pad entries of every weight stay exactly zero (zero gradient, zero Adam step), the
    nn.Parameters follow the engine's optimiser (state_dict gathers them), bf16 runs, and a checkpoint round-trips at reference shapes."""
    from train_iq import SyntheticVocabulary, TrainIQ
    z, cfg, state, batch = load_golden("ref300")
    for precision in ("fp32", "bf16"):
        t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg, precision=precision, num_pretraining_steps=1, attention_dropout=0.1, relu_dropout=0.1))
        t.model.load_state_dict(_full_state(t.model, state))
        t = t.to("cuda")
        b = {k: v.cuda() for k, v in batch.items()}
        before = {k: v.clone() for k, v in t.model.state_dict().items()}
        losses = []
        for _ in range(4):                                     # crosses the phase switch (latent nets start receiving gradients)
            t.fused_training_step(b)
            losses.append(t.last_stats()["loss"])
        assert all(np.isfinite(losses)), losses
        eng = t._last_engine
        eng.optimizer_wait()
        real = torch.zeros(eng.train_size, dtype=torch.bool, device="cuda")
        real[t.model._pad_index(real.device)[0]] = True
        assert float(eng.flat_train[~real].abs().max()) == 0.0          # pads: zero weights ...
        assert float(eng.flat_grad[~real].abs().max()) == 0.0           # ... zero gradients
        after = t.model.state_dict()
        assert after["decoder.output.weight"].shape == (cfg.vocab_size, 300)
        moved = sum(float((after[k].float() - before[k].float()).abs().max()) > 0 for k in before if "num_batches" not in k and "encoder_cnn.cnn.layer" not in k)
        assert moved > 50
        path = str(tmp_path / ("ref300_%s.ckpt" % precision))
        t.save_checkpoint(path)
        t2 = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg, precision=precision, num_pretraining_steps=1))
        t2.load_checkpoint(path)
        sd2 = t2.model.state_dict()
        for k in after:
            assert torch.equal(after[k].cpu().float(), sd2[k].cpu().float()), k


def test_fit_runs_the_conv_stack_one_batch_ahead_and_trains_the_same():
    """TrainIQ.fit hands every step the NEXT batch: its frozen ResNet-18 forward runs on the engine's conv stream underneath the current step
    (models/encoder_cnn.py:18-19).  Same batches, same seeds: the look-ahead run ends with the parameters / BatchNorm statistics of the
    inline run (to fp32-atomic order), and a batch other than the announced one is refused."""
    from train_iq import SyntheticVocabulary, TrainIQ
    import bltvqg_amd.synthetic as synthetic
    z, cfg, state, batch = load_golden("tiny")
    loader = []
    for i in range(4):
        b = synthetic.make_batch(4, cfg.vocab_size, cfg.latent_dim, seed=100 + i, image_hw=64)
        loader.append({k: v.cuda() for k, v in b.items()})
    finals = {}
    for look in (False, True):
        t = TrainIQ(SyntheticVocabulary(cfg.vocab_size), _args(cfg, num_pretraining_steps=2, no_prefetch=not look, seed=5))
        t.model.load_state_dict(_full_state(t.model, state))
        t = t.to("cuda")
        t.fit(loader, max_steps=6, log_every=0)
        eng = t._last_engine
        assert eng.prefetch_pending() == 0                         # the last step announced no next batch
        t.last_stats()
        finals[look] = {k: v.detach().float().cpu().clone() for k, v in t.model.state_dict().items()}
    # the bars of tests/test_prefetch_gpu.py, through TrainIQ.fit: the frozen stack's BatchNorm2d running statistics BIT-equal (the
    # stack sees the same batches in the same order, whichever stream runs it), everything the optimiser touched to fp32-atomic order
    # (1e-4: a one-step slip of the BatchNorm1d / look-ahead ordering would show as ~1e-2 in encoder_cnn.bn.running_*)
    for k in finals[False]:
        a, b_ = finals[False][k], finals[True][k]
        if "running_" in k and "encoder_cnn.cnn." in k:
            assert torch.equal(a, b_), k
        else:
            assert float((a - b_).abs().max()) <= 1e-4 * max(1.0, float(a.abs().max())), k
    t.fused_training_step(loader[0], next_batch=loader[1])
    with pytest.raises(RuntimeError, match="look-ahead"):
        t.fused_training_step(loader[2])
