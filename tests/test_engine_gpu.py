"""GPU parity of the whole HIP train step (engine) against the reference-generated golden fixtures and the CPU oracle.

fp32 engine (exact-fp32 MFMA): tight tolerances — this is the parity anchor ("loss within 1e-3 of the reference", bit-exact
argmax token ids).  bf16 engine: the stated bf16 tolerance (loss 2e-2 relative, logits 3e-2 absolute after LN scale).
"""
import numpy as np
import pytest
import torch

from helpers import load_golden, oracle_run, rel_err

pytestmark = pytest.mark.gpu


def _engine(cfg, B, hw, dtype, p_attn=0.0, p_relu=0.0):
    from bltvqg_amd.engine import StepEngine, make_config
    # --input_mode cat fixtures (run.sh): the context is the 3-token [<start>, category, <end>] row (train_iq.py:72-75)
    len_context = 3 if getattr(cfg, "input_mode", "ans") == "cat" else 5
    c = make_config(B, cfg.hidden_dim, cfg.pwffn_dim, cfg.latent_dim, cfg.emb_dim, cfg.num_layers, cfg.num_heads, cfg.vocab_size,
                    len_context=len_context, image_hw=(hw, hw), dtype=dtype, attention_dropout=p_attn, relu_dropout=p_relu)
    e = StepEngine(c)
    e.allocate()
    return e


def _run(e, batch, phase2, kl_w, seed=0):
    dev = "cuda"
    e.forward(batch["images"].to(dev), batch.get("context", batch["answers"]).to(dev), batch["posteriors"].to(dev), batch["questions"].to(dev),
              batch["eps"].to(dev) if phase2 else None, phase2, seed)
    out = dict(output=e.read(0).cpu(), feats=e.read(2).cpu(), recon=e.read(3).cpu())
    if phase2:
        out["z_logit"] = e.read(1).cpu()
    e.loss_backward(kl_w)
    out["stats"] = e.stats()
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("name", ["tiny", "tiny2"])
@pytest.mark.parametrize("phase2", [False, True])
def test_engine_fp32_matches_reference_golden(name, phase2):
    from oracle import iq_oracle as O
    z, cfg, state, batch = load_golden(name)
    tag = "p2" if phase2 else "p1"
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    e = _engine(cfg, B, hw, 0)
    e.load_state(state)
    kl_w = O.kl_weight(int(z[tag + ".kliter"]), 15000)
    r = _run(e, batch, phase2, kl_w)
    assert rel_err(r["feats"], z[tag + ".feats"]) < 2e-4
    assert rel_err(r["output"], z[tag + ".output"]) < 2e-4
    assert np.array_equal(r["output"].argmax(-1).numpy().astype(np.int32), z[tag + ".argmax"])     # bit-exact token ids
    assert rel_err(r["recon"], z[tag + ".recon"]) < 2e-4
    st = r["stats"]
    assert abs(st["rec"] - float(z[tag + ".loss_rec"])) < 1e-4
    assert abs(st["img"] - float(z[tag + ".loss_img"])) < 1e-4
    total = st["rec"] + 0.1 * st["img"]
    if phase2:
        assert rel_err(r["z_logit"], z[tag + ".z_logit"]) < 2e-4
        assert abs(st["kld"] - float(z[tag + ".kld"])) < 1e-3 * max(1.0, float(z[tag + ".kld"]))
        assert abs(st["aux"] - float(z[tag + ".loss_aux"])) < 1e-4
        total += 0.5 * kl_w * st["kld"] + 1.0 * st["aux"]
    assert abs(total - float(z[tag + ".loss"])) < 1e-3          # BASELINE north_star: loss within 1e-3 of the reference
    # every parameter gradient of the reference
    worst = ("", 0.0)
    for k in z.files:
        if not k.startswith(tag + ".grad."):
            continue
        n = k[len(tag) + 6:]
        ref = torch.from_numpy(z[k])
        got = e.grad_view(n).cpu()
        err = rel_err(got, ref)
        if n == "encoder_cnn.cnn.fc.bias" or float(ref.abs().max()) < 1e-7:
            # mathematically zero (a bias in front of BatchNorm1d cancels): the reference value itself is rounding noise
            assert float(got.abs().max()) < 1e-4, n
            continue
        if err > worst[1]:
            worst = (n, err)
        assert err < 3e-3, (n, err)
    # parameters without a reference gradient (phase 1: r_encoder, latent_*, z_classifier) must stay exactly zero
    have = {k[len(tag) + 6:] for k in z.files if k.startswith(tag + ".grad.")}
    for n in e.train_info:
        if n not in have:
            assert float(e.grad_view(n).abs().max()) == 0.0, n
    print("worst grad rel err", worst)
    # BatchNorm running statistics (train mode, encoder_cnn.py never calls eval())
    if not phase2:
        nb = 0
        for k in z.files:
            if k.startswith("p1.buf.") and not k.endswith("num_batches_tracked"):
                n = k[len("p1.buf."):]
                assert rel_err(e.view(n, 1).cpu(), z[k]) < 1e-4, n
                nb += 1
        assert nb >= 40


@pytest.mark.parametrize("name", ["tiny2"])
@pytest.mark.parametrize("phase2", [False, True])
def test_engine_bf16_within_stated_tolerance(name, phase2):
    from oracle import iq_oracle as O
    z, cfg, state, batch = load_golden(name)
    tag = "p2" if phase2 else "p1"
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    e = _engine(cfg, B, hw, 1)
    e.load_state(state)
    kl_w = O.kl_weight(int(z[tag + ".kliter"]), 15000)
    r = _run(e, batch, phase2, kl_w)
    ref_out = torch.from_numpy(z[tag + ".output"])
    st = r["stats"]
    print("bf16 %s: feats rel %.4f, output rel %.4f max-abs %.4f (|ref| max %.2f), rec %.5f vs %.5f, img %.5f vs %.5f" % (
        tag, rel_err(r["feats"], z[tag + ".feats"]), rel_err(r["output"], ref_out), float((r["output"] - ref_out).abs().max()),
        float(ref_out.abs().max()), st["rec"], float(z[tag + ".loss_rec"]), st["img"], float(z[tag + ".loss_img"])))
    # stated bf16 tolerance: logits 6 % relative (L2) / 8 % of the logit range max-abs, losses 2 % relative
    # (tiny2's 64x64 images leave 2x2x6 = 24 values per BatchNorm2d channel in layer4, which amplifies bf16 rounding of the
    # image feature; the 224x224 fixture below is the representative case)
    assert rel_err(r["feats"], z[tag + ".feats"]) < 1e-1
    assert rel_err(r["output"], ref_out) < 6e-2
    assert (r["output"] - ref_out).abs().max() < 8e-2 * max(1.0, float(ref_out.abs().max()))
    assert abs(st["rec"] - float(z[tag + ".loss_rec"])) < 2e-2 * float(z[tag + ".loss_rec"])
    assert abs(st["img"] - float(z[tag + ".loss_img"])) < 5e-2 * float(z[tag + ".loss_img"])
    errs = []
    for k in z.files:
        if k.startswith(tag + ".grad.") and "weight" in k and z[k].ndim == 2 and z[k].size > 2000:
            n = k[len(tag) + 6:]
            ref = torch.from_numpy(z[k])
            if float(ref.abs().max()) > 1e-6:
                errs.append((rel_err(e.grad_view(n).cpu(), ref), n))
    errs.sort()
    med, worst = errs[len(errs) // 2], errs[-1]
    print("bf16 %s: weight-gradient rel L2 err over %d matrices: median %.4f, worst %.4f (%s)" % (tag, len(errs), med[0], worst[0], worst[1]))
    # bf16 floor on this fixture (measured layer by layer against the fp32 engine): every bf16 conv stage adds ~0.25 % relative
    # error and the randomly initialised, frozen 20-conv stack passes it on undamped, so the image feature arrives ~5 % off;
    # with only 6 samples nothing averages out: typical gradient matrices agree to 7-12 %, the worst (small-norm, cancelling) ~25 %.
    # The fp32 engine (tests above) is the parity anchor; this test guards against bf16-specific regressions.
    assert med[0] < 0.15, med
    assert worst[0] < 0.35, worst


@pytest.mark.parametrize("name", ["small", "big", "runsh"])
def test_engine_small_cfg_fp32_matches_reference_golden(name):
    """BASELINE.json configs[0..1] model (2-layer, d_model 256, 224x224 images, V=8000) at B=8, configs[2..3] model (6-layer,
    d_model 512, 8 heads, F 2048) at B=4, and the launch the reference documents (run.sh:1-10: hidden / latent 1024, FFN 2048, 6 layers,
    8 heads of 128, --input_mode cat) at B=4: summary fixtures produced by the reference."""
    from oracle import iq_oracle as O
    z, cfg, state, batch = load_golden(name)
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    e = _engine(cfg, B, hw, 0)
    e.load_state(state)
    for phase2 in (False, True):
        tag = "p2" if phase2 else "p1"
        if phase2:
            e.load_state(state)       # running statistics were updated by the phase-1 pass
        kl_w = O.kl_weight(int(z[tag + ".kliter"]), 15000)
        r = _run(e, batch, phase2, kl_w)
        out = r["output"]
        idx = torch.from_numpy(z[tag + ".output_idx"])
        assert rel_err(out.reshape(-1)[idx], z[tag + ".output_sample"]) < 3e-4
        assert np.array_equal(out.argmax(-1).numpy().astype(np.int32), z[tag + ".argmax"])
        st = r["stats"]
        total = st["rec"] + 0.1 * st["img"] + (0.5 * kl_w * st["kld"] + st["aux"] if phase2 else 0.0)
        assert abs(total - float(z[tag + ".loss"])) < 1e-3
        names = [str(s) for s in z[tag + ".grad_names"]]
        for n, g in zip(names, z[tag + ".grad_norms"]):
            if n == "encoder_cnn.cnn.fc.bias":
                continue          # mathematically zero (cancelled by BatchNorm1d): rounding noise on both sides
            got = float(e.grad_view(n).double().norm())
            assert abs(got - g) <= 3e-3 * max(g, 1e-6) + 1e-6, (n, got, g)


@pytest.mark.parametrize("name", ["small", "big", "runsh"])
def test_engine_small_cfg_bf16_within_stated_tolerance(name):
    """bf16 engine on the 224x224 fixtures: loss within 2 % (relative) of the reference, sampled logits within 5 %."""
    from oracle import iq_oracle as O
    z, cfg, state, batch = load_golden(name)
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    e = _engine(cfg, B, hw, 1)
    for phase2 in (False, True):
        tag = "p2" if phase2 else "p1"
        e.load_state(state)
        kl_w = O.kl_weight(int(z[tag + ".kliter"]), 15000)
        r = _run(e, batch, phase2, kl_w)
        idx = torch.from_numpy(z[tag + ".output_idx"])
        err = rel_err(r["output"].reshape(-1)[idx], z[tag + ".output_sample"])
        st = r["stats"]
        total = st["rec"] + 0.1 * st["img"] + (0.5 * kl_w * st["kld"] + st["aux"] if phase2 else 0.0)
        print("bf16 " + name + " %s: logits rel %.4f, feats rel %.4f, loss %.4f vs %.4f" % (tag, err, rel_err(r["feats"], z[tag + ".feats"]), total,
                                                                                    float(z[tag + ".loss"])))
        assert err < 5e-2
        assert abs(total - float(z[tag + ".loss"])) < 2e-2 * float(z[tag + ".loss"])
        assert abs(st["rec"] - float(z[tag + ".loss_rec"])) < 1e-2 * float(z[tag + ".loss_rec"])


def test_engine_train_steps_match_oracle_adam():
    """3 full steps (forward, loss, backward, clip 5, Adam with Noam lr) vs the oracle's torch.optim.Adam run,
    crossing the pre-training -> latent phase switch."""
    from oracle import iq_oracle as O
    import bltvqg_amd.synthetic as synthetic
    z, cfg, state, batch0 = load_golden("tiny")
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    batches = [synthetic.make_batch(B, cfg.vocab_size, cfg.latent_dim, seed=100 + i, image_hw=hw) for i in range(4)]
    hp = O.default_hp(num_pretraining_steps=202)
    start = 200           # the phase switch happens at the 3rd step
    final, logs = O.train_steps(state, cfg, batches, hp, start_iter=start)
    e = _engine(cfg, B, hw, 0)
    e.load_state(state)
    kliter = 0
    lr_sum = 0.0
    for i, b in enumerate(batches):
        it = start + i
        phase2 = it >= hp.num_pretraining_steps
        kl_w = O.kl_weight(kliter, hp.full_kl_step)
        r = _run(e, b, phase2, kl_w)
        if phase2:
            kliter += 1
        lr = O.noam_lr(it, cfg.hidden_dim)
        lr_sum += lr
        e.optimizer_step(lr, 5.0)
        st = e.stats()
        assert abs(st["rec"] - logs[i]["rec"]) < 1e-3, "step %d rec %.6f vs %.6f (first read %.6f) %s" % (
            i, st["rec"], logs[i]["rec"], r["stats"]["rec"], str(st))
        assert abs(st["grad_norm"] - logs[i]["grad_norm"]) < 2e-3 * logs[i]["grad_norm"], (i, st["grad_norm"], logs[i]["grad_norm"])
    # Adam normalises every gradient element to O(1), so an element whose true gradient is below fp32 rounding noise can
    # move by +-lr per step in either implementation: compare the UPDATE in aggregate, and bound every element by sum(lr).
    for n in e.train_info:
        got, want, init = e.view(n, 0).cpu(), final[n], state[n]
        assert (got - want).abs().max() <= 2.5 * lr_sum + 1e-7, n
        if want.ndim == 2 and want.numel() > 1000 and float((want - init).abs().max()) > 0:
            assert rel_err(got - init, want - init) < 0.2, (n, rel_err(got - init, want - init))


def test_engine_dropout_is_reproducible_and_active():
    z, cfg, state, batch = load_golden("tiny")
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    e = _engine(cfg, B, hw, 0, 0.1, 0.1)
    e.load_state(state)
    a = _run(e, batch, True, 0.3, seed=5)
    ga = e.flat_grad.clone()
    e.load_state(state)
    b = _run(e, batch, True, 0.3, seed=5)
    assert torch.equal(a["output"], b["output"])
    assert (ga - e.flat_grad).abs().max() <= 1e-5 * ga.abs().max()        # float-atomic accumulation order only
    e.load_state(state)
    c = _run(e, batch, True, 0.3, seed=6)
    assert not torch.equal(a["output"], c["output"])


@pytest.mark.parametrize("dtype", [0, 1])
def test_overlapped_optimizer_matches_synchronous(dtype):
    """optimizer_step(overlap=True) runs clip + Adam on the engine's optimiser stream while the next forward's frozen CNN is already
    running; parameters, losses and gradient norms must follow the synchronous schedule (up to float-atomic accumulation order)."""
    import bltvqg_amd.synthetic as synthetic
    z, cfg, state, batch0 = load_golden("tiny")
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    batches = [synthetic.make_batch(B, cfg.vocab_size, cfg.latent_dim, seed=300 + i, image_hw=hw) for i in range(6)]

    def run(overlap):
        e = _engine(cfg, B, hw, dtype)
        e.load_state(state)
        logs = []
        for i, b in enumerate(batches):
            d = {k: v.cuda() for k, v in b.items()}
            phase2 = i >= 2
            e.forward(d["images"], d["answers"], d["posteriors"], d["questions"], d["eps"] if phase2 else None, phase2, 11 + i)
            e.loss_backward(0.25)
            e.optimizer_step(3e-4, 5.0, overlap=overlap)
            if i % 2 == 1:
                logs.append(e.stats())        # reading the statistics in between must also be ordered behind the pending update
        e.optimizer_wait()
        torch.cuda.synchronize()
        return e.flat_train.clone().cpu(), logs

    p_sync, l_sync = run(False)
    p_async, l_async = run(True)
    # Adam turns a gradient element at fp32 rounding-noise level into a +-lr move, and float-atomic accumulation order differs from run
    # to run: bound every element by the total step budget and compare the update in aggregate (as the oracle test above does)
    e0 = _engine(cfg, B, hw, dtype)
    e0.load_state(state)
    init = e0.flat_train.clone().cpu()
    lr_sum = 3e-4 * len(batches)
    assert float((p_sync - p_async).abs().max()) <= 2.5 * lr_sum, float((p_sync - p_async).abs().max())
    assert rel_err(p_async - init, p_sync - init) < 0.02, rel_err(p_async - init, p_sync - init)
    for a, b in zip(l_sync, l_async):
        for k in ("rec", "img", "kld", "aux", "grad_norm"):
            assert abs(a[k] - b[k]) <= 2e-4 * max(1.0, abs(a[k])), (k, a[k], b[k])


def test_full_size_properties_batch_permutation_and_gradient_linearity():
    """Size-independent properties at the BENCH size (BASELINE configs[1]: 2-layer d_model 256, batch 128, 224x224, bf16), where the
    oracle is too slow to be the checker:
      * permuting the batch permutes every per-sample output (train-mode BatchNorm statistics, the only cross-sample coupling, are
        permutation invariant) and leaves the losses unchanged;
      * the backward is linear in the output gradient: doubling d(output) doubles every parameter gradient (exact in bf16 / fp32 up
        to the float-atomic accumulation order)."""
    import bltvqg_amd.synthetic as synthetic
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.trainer import init_reference_style
    B, V, Z = 128, 8000, 256
    c = make_config(B, 256, 512, Z, 300, 2, 4, V, dtype=1, attention_dropout=0.0, relu_dropout=0.0)
    e = StepEngine(c)
    e.allocate()
    init_reference_style(e, seed=3)
    b = synthetic.make_batch(B, V, Z, seed=77)
    d = {k: v.cuda() for k, v in b.items()}
    eps = torch.randn(B, Z, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).cuda()

    def fwd(idx):
        e.forward(d["images"][idx], d["answers"][idx], d["posteriors"][idx], d["questions"][idx], eps[idx], True, 0)
        out, zl, feats = e.read(0).clone(), e.read(1).clone(), e.read(2).clone()
        e.loss_backward(0.5)
        st = e.stats()
        torch.cuda.synchronize()
        return out, zl, feats, st

    ident = torch.arange(B, device="cuda")
    o0, z0, f0, s0 = fwd(ident)
    o1, z1, f1, s1 = fwd(perm)
    assert torch.isfinite(o0).all() and torch.isfinite(f0).all()
    assert rel_err(f1.cpu(), f0[perm].cpu()) < 2e-2          # bf16 BatchNorm partial sums are formed in a different order
    assert rel_err(o1.cpu(), o0[perm].cpu()) < 2e-2
    assert rel_err(z1.cpu(), z0[perm].cpu()) < 2e-2
    for k in ("rec", "img", "kld", "aux"):
        assert abs(s0[k] - s1[k]) <= 2e-3 * max(1.0, abs(s0[k])), (k, s0[k], s1[k])
    assert s0["n_targets"] == s1["n_targets"]

    # linearity of the backward in d(output)
    e.forward(d["images"], d["answers"], d["posteriors"], d["questions"], eps, True, 0)
    g = torch.Generator(device="cuda").manual_seed(9)
    dout = torch.randn(B, c.len_target, V, device="cuda", generator=g) * 1e-3
    e.backward_external(d_output=dout)
    g1 = e.flat_grad.clone()
    dout2 = 2.0 * dout
    e.backward_external(d_output=dout2)
    g2 = e.flat_grad.clone()
    torch.cuda.synchronize()
    assert float(g1.abs().max()) > 0
    assert rel_err(g2.cpu(), 2.0 * g1.cpu()) < 1e-4


def test_ragged_last_batch_shares_the_adam_step_counters():
    """engines of different batch shapes share parameters AND Adam moments (allocate(share_from=...)); the bias-correction
    counters belong to those moments.  3 full steps on a B-sized engine + 1 step on a smaller one (the ragged last batch of an epoch:
    the reference DataLoader has no drop_last, utils/data_loader.py:178-206) must follow the oracle's single torch.optim.Adam."""
    from oracle import iq_oracle as O
    import bltvqg_amd.synthetic as synthetic
    z, cfg, state, batch0 = load_golden("tiny")
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    sizes = [B, B, B, B - 1]
    batches = [synthetic.make_batch(n, cfg.vocab_size, cfg.latent_dim, seed=900 + i, image_hw=hw) for i, n in enumerate(sizes)]
    hp = O.default_hp(num_pretraining_steps=0)
    final, logs = O.train_steps(state, cfg, batches, hp, start_iter=50)
    full = _engine(cfg, B, hw, 0)
    full.load_state(state)
    from bltvqg_amd.engine import StepEngine, make_config
    c = make_config(B - 1, cfg.hidden_dim, cfg.pwffn_dim, cfg.latent_dim, cfg.emb_dim, cfg.num_layers, cfg.num_heads, cfg.vocab_size,
                    image_hw=(hw, hw), dtype=0, attention_dropout=0.0, relu_dropout=0.0)
    ragged = StepEngine(c)
    ragged.allocate(share_from=full)
    lr_sum = 0.0
    for i, b in enumerate(batches):
        e = full if sizes[i] == B else ragged
        _run(e, b, True, O.kl_weight(i, hp.full_kl_step))
        lr = O.noam_lr(50 + i, cfg.hidden_dim)
        lr_sum += lr
        e.optimizer_step(lr, 5.0)
        assert full.adam_steps() == ragged.adam_steps() == (i + 1, i + 1)
    for n in full.train_info:
        got, want, init = full.view(n, 0).cpu(), final[n], state[n]
        assert (got - want).abs().max() <= 2.5 * lr_sum + 1e-7, n
        if want.ndim == 2 and want.numel() > 1000 and float((want - init).abs().max()) > 0:
            # a fresh counter on the ragged engine would scale its update by ~0.3 at t = 1: far outside this bound
            assert rel_err(got - init, want - init) < 0.2, (n, rel_err(got - init, want - init))


def test_out_of_range_token_ids_are_reported_not_dereferenced():
    """an id outside [0, V) must not index the embedding table / logits / gradient buffers; it is counted, treated as <pad>
    and surfaced as an error on the host (the reference raises a device-side index error)."""
    from bltvqg_amd import _lib
    z, cfg, state, batch = load_golden("tiny")
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    e = _engine(cfg, B, hw, 0)
    e.load_state(state)
    bad = {k: v.clone() for k, v in batch.items()}
    bad["questions"][1, 3] = cfg.vocab_size            # one past the end
    bad["answers"][0, 2] = -7
    bad["posteriors"][2, 5] = 10 ** 9
    d = {k: v.cuda() for k, v in bad.items()}
    e.forward(d["images"], d["answers"], d["posteriors"], d["questions"], d["eps"], True, 0)
    e.loss_backward(0.5)
    torch.cuda.synchronize()
    with pytest.raises(_lib.HipError, match="token id"):
        e.stats()
    st = e.stats(check_ids=False)
    assert st["bad_ids"] == 4.0                         # questions[1,3] is seen twice: as the shifted decoder input and as a target
    assert all(torch.isfinite(torch.tensor(st[k])) for k in ("rec", "img", "kld", "aux"))
    assert torch.isfinite(e.flat_grad).all()
    # the next clean batch clears the flag
    _run(e, batch, True, 0.5)
    assert e.stats()["bad_ids"] == 0.0


def test_trusted_weight_shadows_equal_a_full_refresh():
    """bf16 engine driven by DataParallelStep (which promises that only the engine's optimiser writes the parameters): the update writes
    the plain bf16 weight shadow itself and forward only derives the transposed copies.  After steps on both sides of the phase switch the
    forward must be BIT-identical to that of a fresh engine that loads the same parameters (and rebuilds every shadow from fp32)."""
    import bltvqg_amd.synthetic as synthetic
    from bltvqg_amd.trainer import DataParallelStep
    z, cfg, state, batch0 = load_golden("tiny")
    B, hw = int(z["meta_cfg"][7]), int(z["meta_cfg"][8])
    e = _engine(cfg, B, hw, 1)
    e.load_state(state)
    dp = DataParallelStep(e, None, overlap_optimizer=True)
    for i, phase2 in enumerate((False, False, True, True)):
        b = synthetic.make_batch(B, cfg.vocab_size, cfg.latent_dim, seed=700 + i, image_hw=hw)
        d = {k: v.cuda() for k, v in b.items()}
        dp.run(d["images"], d["answers"], d["posteriors"], d["questions"], d["eps"] if phase2 else None, phase2, seed=i, kl_weight=0.3, lr=1e-3)
    dp.finish()
    probe = {k: v.cuda() for k, v in batch0.items()}
    e.forward(probe["images"], probe["answers"], probe["posteriors"], probe["questions"], probe["eps"], True, 0)
    out1, zl1 = e.read(0).clone(), e.read(1).clone()
    e.loss_backward(0.3)
    g1 = e.flat_grad.clone()
    torch.cuda.synchronize()
    fresh = _engine(cfg, B, hw, 1)
    fresh.load_state({n: e.view(n, 0).clone() for n in e.train_info} | {n: e.view(n, 1).clone() for n in e.frozen_info})
    fresh.forward(probe["images"], probe["answers"], probe["posteriors"], probe["questions"], probe["eps"], True, 0)
    out2, zl2 = fresh.read(0), fresh.read(1)
    fresh.loss_backward(0.3)
    torch.cuda.synchronize()
    assert torch.equal(out1, out2) and torch.equal(zl1, zl2)                    # forward: same shadows, bit for bit
    assert rel_err(g1.cpu(), fresh.flat_grad.cpu()) < 1e-4                       # backward reads the TRANSPOSED shadows (float-atomic order only)
    # a write from outside (load_state) must invalidate the shortcut: perturb, reload, compare with the fresh engine again
    pert = {n: e.view(n, 0).clone() for n in e.train_info} | {n: e.view(n, 1).clone() for n in e.frozen_info}
    pert["decoder.output.weight"] = pert["decoder.output.weight"] * 1.5
    e.load_state(pert); fresh.load_state(pert)
    e.forward(probe["images"], probe["answers"], probe["posteriors"], probe["questions"], probe["eps"], True, 0)
    fresh.forward(probe["images"], probe["answers"], probe["posteriors"], probe["questions"], probe["eps"], True, 0)
    assert torch.equal(e.read(0), fresh.read(0))


@pytest.mark.parametrize("dtype", [0, 1])
@pytest.mark.parametrize("T", [13, 40])
@pytest.mark.parametrize("phase2", [False, True])
def test_incremental_greedy_decode_equals_the_prefix_redecode(dtype, T, phase2):
    """Round 4: step t of bltvqg_engine_decode_greedy computes row t of every sample and attends over the q|k|v rows of the
    earlier steps (the layer buffers are the key / value cache) instead of re-running the decoder over the whole prefix as the reference
    does (models/iq.py:134-141) and as round 2's form did (debug key 29 = 1).  Tokens, top-6 indices AND top-6 probabilities must be
    bit-identical between the two forms: T = 40 > 32 takes the VALU attention kernel in bf16 too, bf16 folds the LayerNorms (row statistics in
    64-column slots: independent of the rows per launch), phase 2 adds z to row 0."""
    import bltvqg_amd.synthetic as synthetic
    from bltvqg_amd.engine import StepEngine, make_config
    from bltvqg_amd.trainer import init_reference_style
    B, H, F, L, NH, V = 24, 128, 256, 3, 2, 503
    c = make_config(B, H, F, H, 40, L, NH, V, image_hw=(64, 64), len_target=T, dtype=dtype, attention_dropout=0.0, relu_dropout=0.0)
    e = StepEngine(c, "cuda:0")
    e.allocate()
    init_reference_style(e, seed=3)
    # a decisive model: scale the vocabulary projection up so that the argmax is not a coin toss between near-equal logits
    with torch.no_grad():
        e.view("decoder.output.weight", 0).mul_(30.0)
    e.params_changed()
    b = synthetic.make_batch(B, V, H, seed=11, image_hw=64)
    img, ans, eps = b["images"].cuda(), b["answers"].cuda(), b["eps"].cuda()
    outs = []
    try:
        for key in (1, 0):
            e.lib.bltvqg_debug_set(29, key)
            outs.append([x.clone() for x in e.decode_greedy(img, ans, eps, phase2)])
            torch.cuda.synchronize()
    finally:
        e.lib.bltvqg_debug_set(29, 0)
    full, inc = outs
    assert torch.equal(full[0], inc[0])
    assert torch.equal(full[1], inc[1])
    assert torch.equal(full[2], inc[2])
    # the sentences are not degenerate: several distinct tokens, and not every sample says the same thing
    assert int(torch.unique(inc[0]).numel()) > 5 and not bool((inc[0] == inc[0][0:1]).all())
