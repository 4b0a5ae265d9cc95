"""Fixture generator — runs ONLY in the build container (needs /root/reference).

Imports the real reference model (`/root/reference/models/*.py`) with three offline shims
(SURVEY.md §8c / Appendix A), loads deterministic synthetic weights into it, runs the reference
forward/backward in both training phases with dropout disabled and an injected eps, and stores
inputs + reference outputs as small .npz files next to this script.  No reference source is copied.

    python tests/golden/make_golden.py [config names]
"""
import importlib.util
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
REF = "/root/reference"

from synth import synth_state            # noqa: E402
import bltvqg_amd.synthetic as synthetic  # noqa: E402
from oracle import iq_oracle as O        # noqa: E402  (only for the name/shape spec)


# ---- shim 1: numpy alias removed in NumPy 1.24, used at transformer_layers.py:551
np.float = float


# ---- shim 2: stand-in torchvision.models.resnet18 (public topology + parameter names, random init)
class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(out + idt)


class _ResNet18(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for name, cout, stride in (("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2)):
            setattr(self, name, nn.Sequential(_BasicBlock(cin, cout, stride), _BasicBlock(cout, cout, 1)))
            cin = cout
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, 1000)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


def _install_shims():
    tv = types.ModuleType("torchvision")
    tvm = types.ModuleType("torchvision.models")
    tvm.resnet18 = lambda pretrained=False, **kw: _ResNet18()
    tv.models = tvm
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = tvm
    sys.modules["torchtext"] = types.ModuleType("torchtext")      # shim 3 (train_utils.py:6)
    sys.path.insert(0, REF)


def _vocab(V):
    spec = importlib.util.spec_from_file_location("ref_train_utils", os.path.join(REF, "utils/train_utils.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    v = m.Vocabulary()
    i = 0
    while len(v) < V:
        v.add_word("w%d" % i)
        i += 1
    return v


def build_reference(cfg):
    import models  # the reference package
    args = SimpleNamespace(emb_dim=cfg.emb_dim, hidden_dim=cfg.hidden_dim, latent_dim=cfg.latent_dim,
                           pwffn_dim=cfg.pwffn_dim, num_layers=cfg.num_layers, num_heads=cfg.num_heads,
                           device="cpu", emb_file=None, root_dir=".")
    model = models.IQ(False, _vocab(cfg.vocab_size), args)
    for mod in model.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
    return model


def run_reference(model, state, batch, phase2, hp, kliter, context_key="answers"):
    """One reference forward + loss (train_iq.py:81-103 formulas with torch's own criteria) + backward."""
    import models.transformer_layers as TL
    missing = model.load_state_dict(state, strict=False)
    assert not missing.unexpected_keys, missing.unexpected_keys
    model.train()
    model.switch_GVT_train_mode(phase2)
    model.zero_grad()
    eps = batch["eps"]
    real_randn = torch.randn
    TL.torch.randn = lambda *a, **k: eps.clone()     # inject eps at transformer_layers.py:45
    try:
        # (TrainIQ.forward, train_iq.py:72-75: the context is `answers`, or `answer_types_for_input` under --input_mode cat)
        output, z_logit, kld, (feats, recon) = model(batch["images"], batch[context_key], batch["posteriors"], batch["questions"])
    finally:
        TL.torch.randn = real_randn
    target = batch["questions"]
    crit = nn.CrossEntropyLoss(ignore_index=0)
    loss_rec = crit(output.reshape(-1, output.size(-1)), target.reshape(-1))
    loss_img = nn.MSELoss()(feats, recon)
    res = dict(output=output, feats=feats, recon=recon, loss_rec=loss_rec, loss_img=loss_img)
    if not phase2:
        loss = loss_rec + hp.image_recon_lambda * loss_img
    else:
        zl = z_logit.unsqueeze(1).repeat(1, output.size(1), 1)
        loss_aux = crit(zl.reshape(-1, zl.size(-1)), target.reshape(-1))
        import math
        w = min(math.tanh(6 * kliter / hp.full_kl_step - 3) + 1, 1)
        loss = loss_rec + hp.kl_ceiling * w * kld + hp.aux_ceiling * loss_aux + hp.image_recon_lambda * loss_img
        res.update(z_logit=z_logit, kld=kld, loss_aux=loss_aux)
    loss.backward()
    res["loss"] = loss
    grads = {}
    sd_names = dict(model.named_parameters())
    for n, p in sd_names.items():
        if p.grad is not None:
            grads[n] = p.grad.detach().clone()
    res["grads"] = grads
    res["buffers"] = {k: v.detach().clone() for k, v in model.state_dict().items() if O.is_buffer(k)}
    return res


CONFIGS = {
    # tiny: everything stored (all stage outputs and every gradient)
    "tiny": dict(cfg=SimpleNamespace(emb_dim=20, hidden_dim=64, latent_dim=64, pwffn_dim=128, num_layers=1, num_heads=4,
                                     vocab_size=97), B=4, hw=64, seed=11, full=True),
    # tiny2: 2 layers, d_head=32, odd vocab, B=6 — exercises multi-layer accumulation paths
    "tiny2": dict(cfg=SimpleNamespace(emb_dim=36, hidden_dim=128, latent_dim=32, pwffn_dim=96, num_layers=2, num_heads=4,
                                      vocab_size=203), B=6, hw=64, seed=12, full=True),
    # ref300: the reference's CLI default WIDTHS (train_iq.py:315-325: hidden 300 = 4 heads of 75, latent 300, FFN 600, emb 300) at one
    # layer and a small vocabulary: exercises the padded engine layout (blt-vqg_amd/padded.py); samples + gradient norms + selected tensors
    "ref300": dict(cfg=SimpleNamespace(emb_dim=300, hidden_dim=300, latent_dim=300, pwffn_dim=600, num_layers=1, num_heads=4,
                                       vocab_size=211), B=4, hw=64, seed=15, full=False, decode=True, keep=(
        "decoder.output.bias", "embedding.1.bias", "decoder.decoder.dec.0.multi_head_attention_dec.query_linear.weight",
        "answer_encoder.encoder.enc.0.layer_norm_mha.weight", "latent_layer.mean_logvar_posterior.0.weight", "encoder_cnn.bn.weight")),
    # ref300l4: the reference's CLI defaults at their own DEPTH too (train_iq.py:315-325: 4 layers of hidden 300 / FFN 600 / 4 heads)
    "ref300l4": dict(cfg=SimpleNamespace(emb_dim=300, hidden_dim=300, latent_dim=300, pwffn_dim=600, num_layers=4, num_heads=4,
                                         vocab_size=211), B=4, hw=64, seed=17, full=False, keep=(
        "decoder.output.bias", "embedding.1.bias", "decoder.decoder.dec.3.multi_head_attention_enc_dec.query_linear.weight",
        "answer_encoder.r_encoder.enc.0.layer_norm_mha.weight", "latent_layer.mean_logvar_prior.0.weight")),
    # runsh: the one launch the reference documents (run.sh:1-10): hidden 1024 / latent 1024 / FFN 2048, 6 layers, 8 heads of 128,
    # --input_mode cat (the 3-token [<start>, category, <end>] context), at B=4, 224x224: summaries only
    "runsh": dict(cfg=SimpleNamespace(emb_dim=300, hidden_dim=1024, latent_dim=1024, pwffn_dim=2048, num_layers=6, num_heads=8,
                                      vocab_size=8000), B=4, hw=224, seed=16, full=False, context="answer_types_for_input"),
    # small = BASELINE.json configs[0] model (2-layer, d_model 256) at B=8, 224x224: summaries only
    "small": dict(cfg=SimpleNamespace(emb_dim=300, hidden_dim=256, latent_dim=256, pwffn_dim=512, num_layers=2, num_heads=4,
                                      vocab_size=8000), B=8, hw=224, seed=13, full=False),
    # big = BASELINE.json configs[2..3] model (6-layer, d_model 512, 8 heads, F 2048) at B=4, 224x224: summaries only
    "big": dict(cfg=SimpleNamespace(emb_dim=300, hidden_dim=512, latent_dim=512, pwffn_dim=2048, num_layers=6, num_heads=8,
                                    vocab_size=8000), B=4, hw=224, seed=14, full=False),
}


def main():
    _install_shims()
    torch.manual_seed(0)
    np.random.seed(0)
    hp = O.default_hp()
    only = sys.argv[1:]                  # `python make_golden.py big` regenerates one fixture
    for name, c in CONFIGS.items():
        if only and name not in only:
            continue
        cfg = c["cfg"]
        spec = O.iq_spec(cfg)
        state = synth_state(spec, seed=c["seed"])
        batch = synthetic.make_batch(c["B"], cfg.vocab_size, cfg.latent_dim, seed=c["seed"], image_hw=c["hw"])
        model = build_reference(cfg)
        # alias-complete state for load_state_dict (embedding / latent_layer appear under several prefixes)
        full_state = {}
        for k in model.state_dict().keys():
            base = k
            for alias in ("answer_encoder.embedding.", "decoder.embedding."):
                if k.startswith(alias):
                    base = "embedding." + k[len(alias):]
            if k.startswith("answer_encoder.latent_layer."):
                base = k[len("answer_encoder."):]
            full_state[k] = state[base]
        assert set(spec.keys()) <= set(full_state.keys()), set(spec.keys()) - set(full_state.keys())
        out = {"meta_cfg": np.array([cfg.emb_dim, cfg.hidden_dim, cfg.latent_dim, cfg.pwffn_dim, cfg.num_layers,
                                     cfg.num_heads, cfg.vocab_size, c["B"], c["hw"], c["seed"]], dtype=np.int64),
               "meta_cat": np.array(1 if c.get("context") else 0, dtype=np.int64)}
        for phase2 in (False, True):
            tag = "p2" if phase2 else "p1"
            kliter = 5000 if phase2 else 0
            res = run_reference(model, full_state, batch, phase2, hp, kliter, c.get("context", "answers"))
            out[tag + ".kliter"] = np.array(kliter)
            for k in ("loss", "loss_rec", "loss_img", "kld", "loss_aux"):
                if k in res:
                    out["%s.%s" % (tag, k)] = res[k].detach().numpy().astype(np.float64)
            out[tag + ".feats"] = res["feats"].detach().numpy()
            out[tag + ".recon"] = res["recon"].detach().numpy()
            o = res["output"].detach()
            out[tag + ".argmax"] = o.argmax(-1).numpy().astype(np.int32)
            if c["full"]:
                out[tag + ".output"] = o.numpy()
                if phase2:
                    out[tag + ".z_logit"] = res["z_logit"].detach().numpy()
                for n, g in res["grads"].items():
                    if n.startswith("answer_encoder.embedding.") or n.startswith("decoder.embedding.") or \
                            n.startswith("answer_encoder.latent_layer."):
                        continue
                    out["%s.grad.%s" % (tag, n)] = g.numpy()
            else:
                gsel = torch.Generator().manual_seed(5)
                idx = torch.randint(0, o.numel(), (512,), generator=gsel)
                out[tag + ".output_idx"] = idx.numpy()
                out[tag + ".output_sample"] = o.reshape(-1)[idx].numpy()
                out[tag + ".output_sum"] = np.array(o.double().sum().item())
                if phase2:
                    out[tag + ".z_logit_sample"] = res["z_logit"].detach().reshape(-1)[idx % res["z_logit"].numel()].numpy()
                names, norms = [], []
                for n, g in sorted(res["grads"].items()):
                    if n.startswith("answer_encoder.embedding.") or n.startswith("decoder.embedding.") or \
                            n.startswith("answer_encoder.latent_layer."):
                        continue
                    names.append(n)
                    norms.append(g.double().norm().item())
                out[tag + ".grad_names"] = np.array(names)
                out[tag + ".grad_norms"] = np.array(norms)
                for n in c.get("keep", ()):
                    if n in res["grads"]:
                        out["%s.grad.%s" % (tag, n)] = res["grads"][n].numpy()
            if not phase2:
                for k, v in res["buffers"].items():
                    if c["full"] or k.startswith("encoder_cnn.bn.") or k.startswith("encoder_cnn.cnn.bn1."):
                        out["p1.buf." + k] = v.numpy()
            # the train-mode forward updated BN running stats inside the model; full_state is reloaded each run
        if c["full"] or c.get("decode"):
            # greedy decoding (iq.py:117-152) under model.eval(), latent off and on (eps injected)
            import models.transformer_layers as TL
            for phase2 in (False, True):
                model.load_state_dict(full_state, strict=False)
                model.eval()
                model.switch_GVT_train_mode(phase2)
                real_randn = torch.randn
                TL.torch.randn = lambda *a, **k: batch["eps"].clone()
                try:
                    with torch.no_grad():
                        _sent, targs, tvals = model.decode_greedy(batch["images"], batch["answers"], max_decode_length=12)
                finally:
                    TL.torch.randn = real_randn
                tag = "dec2" if phase2 else "dec1"
                out[tag + ".top_idx"] = targs.numpy().astype(np.int32)
                out[tag + ".top_val"] = tvals.numpy()
            model.train()
        if name == "small":
            # state_dict schema of the reference model (260 keys at L=2, aliases included) for tests/test_host_api.py
            with open(os.path.join(HERE, "state_keys_small.txt"), "w") as fh:
                fh.write("\n".join(model.state_dict().keys()) + "\n")
        path = os.path.join(HERE, "%s.npz" % name)
        np.savez_compressed(path, **out)
        print(name, "->", path, "%.1f KB" % (os.path.getsize(path) / 1024))
        for tag in ("p1", "p2"):
            print("  ", tag, "loss", float(out[tag + ".loss"]), "rec", float(out[tag + ".loss_rec"]), "img", float(out[tag + ".loss_img"]),
                  "kld", float(out.get(tag + ".kld", 0)), "aux", float(out.get(tag + ".loss_aux", 0)))


if __name__ == "__main__":
    main()
