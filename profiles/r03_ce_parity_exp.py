"""How much of the bf16 engine's loss error is the bf16 rounding of the STORED logits?  Recompute CE from fp32 logits of the engine's own
decoder output and compare all three with the CPU oracle (big cfg, B=256, dropout off)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bltvqg_amd.synthetic as synthetic
from bltvqg_amd.engine import StepEngine, make_config
from bltvqg_amd.trainer import init_reference_style
from oracle import iq_oracle as O
from types import SimpleNamespace
B = 256
cfg = dict(hidden_dim=512, pwffn_dim=2048, latent_dim=512, emb_dim=300, num_layers=6, num_heads=8, vocab_size=8000)
dev = torch.device("cuda", 0)
c = make_config(B, 512, 2048, 512, 300, 6, 8, 8000, dtype=1, attention_dropout=0.0, relu_dropout=0.0)
e = StepEngine(c, dev); e.allocate(); init_reference_style(e, seed=0)
batch = synthetic.make_batch(B, 8000, 512, seed=1234, image_hw=224)
d = {k: batch[k].to(dev) for k in ("images", "answers", "posteriors", "questions")}
e.forward(d["images"], d["answers"], d["posteriors"], d["questions"], batch["eps"].to(dev), True, 0)
dec = e.read(6)                                   # decoder outputs [B,T,H] (bf16 values, widened)
logits_bf16 = e.read(0)                           # stored bf16 logits, widened
zlog = e.read(1)
e.loss_backward(0.5)
st = e.stats()
W, bias = e.view("decoder.output.weight"), e.view("decoder.output.bias")
tgt = d["questions"].reshape(-1)
ce = torch.nn.CrossEntropyLoss(ignore_index=0)
lf = dec.reshape(-1, 512) @ W.t() + bias                                     # fp32 GEMM on fp32 weights
lb = dec.reshape(-1, 512).bfloat16().float() @ W.bfloat16().float().t() + bias   # fp32 accumulation of bf16 operands (what the MFMA computes)
print("engine rec %.6f | CE(stored bf16 logits) %.6f | CE(fp32 accum of bf16 operands) %.6f | CE(fp32 weights) %.6f" % (
    st["rec"], float(ce(logits_bf16.reshape(-1, 8000), tgt)), float(ce(lb, tgt)), float(ce(lf, tgt))))
ns = SimpleNamespace(num_regions=0, region_dim=0, **cfg)
state = {n: e.view(n, 0).detach().cpu().clone() for n in e.train_info}
state.update({n: e.view(n, 1).detach().cpu().clone() for n in e.frozen_info})
for k, shape in O.iq_spec(ns).items():
    if k not in state: state[k] = torch.zeros(shape, dtype=torch.long)
with torch.no_grad():
    out, z_logit, kld, recon, _ = O.iq_forward(O.clone_params(state, requires_grad=False), ns, batch["images"], batch["answers"], batch["posteriors"], batch["questions"], True, batch["eps"], None, 0.0, True, {})
    _, ost = O.calculate_losses(out, recon, kld, z_logit, batch["questions"], True, 100, O.default_hp())
print("oracle rec %.6f aux %.6f img %.6f | engine aux %.6f img %.6f" % (ost["rec"], ost["aux"], ost["img"], st["aux"], st["img"]))
# decoder output error vs oracle's logits: project oracle logits error
print("max |logit err| stored-vs-oracle %.4f, fp32accum-vs-oracle %.4f" % (float((logits_bf16.cpu() - out).abs().max()), float((lb.cpu().reshape(out.shape) - out).abs().max())))
