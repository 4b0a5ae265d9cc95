"""CPU oracle for the BLT-VQG training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) functional restatement of the reference
algorithm.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import it; the product package ``blt-vqg_amd`` never does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference
(`/root/reference/models/*.py`) in the build container and writes the fixtures under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function below
against them (forward tensors, losses and every parameter gradient, both phases).

All citations are relative to /root/reference/.  Parameters are passed as a flat
``dict[str, Tensor]`` keyed by the reference ``state_dict`` names.
"""
import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

PAD, SOQ, SOR, EOS, UNK, POS = 0, 1, 2, 3, 4, 5   # utils/train_utils.py:18-37

RESNET_LAYERS = [("layer1", 64, 1), ("layer2", 128, 2), ("layer3", 256, 2), ("layer4", 512, 2)]


# ----------------------------------------------------------------------------------
# parameter specification (names + shapes of the reference state_dict)
# ----------------------------------------------------------------------------------
def resnet18_spec():
    """torchvision 0.8.2 resnet18 parameter/buffer names (encoder_cnn.py:17), fc excluded."""
    spec = {}

    def bn(name, c):
        spec[name + ".weight"] = (c,)
        spec[name + ".bias"] = (c,)
        spec[name + ".running_mean"] = (c,)
        spec[name + ".running_var"] = (c,)
        spec[name + ".num_batches_tracked"] = ()

    spec["conv1.weight"] = (64, 3, 7, 7)
    bn("bn1", 64)
    cin = 64
    for lname, cout, stride in RESNET_LAYERS:
        for b in range(2):
            s = stride if b == 0 else 1
            pre = "%s.%d." % (lname, b)
            spec[pre + "conv1.weight"] = (cout, cin, 3, 3)
            bn(pre + "bn1", cout)
            spec[pre + "conv2.weight"] = (cout, cout, 3, 3)
            bn(pre + "bn2", cout)
            if s != 1 or cin != cout:
                spec[pre + "downsample.0.weight"] = (cout, cin, 1, 1)
                bn(pre + "downsample.1", cout)
            cin = cout
    return spec


def iq_spec(cfg):
    """Unique (un-aliased) state_dict entries of models.IQ (iq.py:25-48)."""
    H, F_, Z, E, L, V = cfg.hidden_dim, cfg.pwffn_dim, cfg.latent_dim, cfg.emb_dim, cfg.num_layers, cfg.vocab_size
    spec = {}
    spec["embedding.0.weight"] = (V, E)
    spec["embedding.1.weight"] = (H, E)
    spec["embedding.1.bias"] = (H,)
    if getattr(cfg, "num_regions", 0) > 0:       # BASELINE configs[4] / SURVEY A2': precomputed region features, no backbone
        spec["encoder_cnn.region_proj.weight"] = (H, cfg.region_dim)
        spec["encoder_cnn.region_proj.bias"] = (H,)
        if getattr(cfg, "region_pool", 0) in (1, "attention"):      # region-attention pooling (build-defined)
            spec["encoder_cnn.region_attn.weight"] = (1, H)
    else:
        for k, s in resnet18_spec().items():
            spec["encoder_cnn.cnn." + k] = s
        spec["encoder_cnn.cnn.fc.weight"] = (H, 512)
        spec["encoder_cnn.cnn.fc.bias"] = (H,)
    for k, s in (("weight", (H,)), ("bias", (H,)), ("running_mean", (H,)), ("running_var", (H,)), ("num_batches_tracked", ())):
        spec["encoder_cnn.bn." + k] = s
    for net, din in (("mean_logvar_prior", H), ("mean_logvar_posterior", 2 * H)):
        for idx, d in ((0, din), (3, 2 * Z), (6, 2 * Z)):
            spec["latent_layer.%s.%d.weight" % (net, idx)] = (2 * Z, d)
            spec["latent_layer.%s.%d.bias" % (net, idx)] = (2 * Z,)
    spec["latent_projection.weight"] = (H, Z)
    spec["latent_projection.bias"] = (H,)

    def mha(pre):
        for n in ("query", "key", "value", "output"):
            spec[pre + n + "_linear.weight"] = (H, H)

    def ffn(pre):
        spec[pre + "layers.0.weight"] = (F_, H)
        spec[pre + "layers.0.bias"] = (F_,)
        spec[pre + "layers.1.weight"] = (H, F_)
        spec[pre + "layers.1.bias"] = (H,)

    def ln(pre):
        spec[pre + ".weight"] = (H,)
        spec[pre + ".bias"] = (H,)

    for enc in ("encoder", "r_encoder"):
        for l in range(L):
            pre = "answer_encoder.%s.enc.%d." % (enc, l)
            mha(pre + "multi_head_attention.")
            ffn(pre + "positionwise_feed_forward.")
            ln(pre + "layer_norm_mha")
            ln(pre + "layer_norm_ffn")
        ln("answer_encoder.%s.layer_norm" % enc)
    for l in range(L):
        pre = "decoder.decoder.dec.%d." % l
        mha(pre + "multi_head_attention_dec.")
        mha(pre + "multi_head_attention_enc_dec.")
        ffn(pre + "positionwise_feed_forward.")
        ln(pre + "layer_norm_mha_dec")
        ln(pre + "layer_norm_mha_enc")
        ln(pre + "layer_norm_ffn")
    ln("decoder.decoder.layer_norm")
    spec["decoder.output.weight"] = (V, H)
    spec["decoder.output.bias"] = (V,)
    spec["decoder.z_classifier.weight"] = (V, H)
    spec["decoder.z_classifier.bias"] = (V,)
    spec["image_reconstructor.layers.fc0.weight"] = (F_, H)
    spec["image_reconstructor.layers.fc0.bias"] = (F_,)
    spec["image_reconstructor.layers.fc1.weight"] = (H, F_)
    spec["image_reconstructor.layers.fc1.bias"] = (H,)
    return spec


def is_frozen(name):
    """encoder_cnn.py:18-19 — every backbone parameter except the new fc is frozen."""
    return name.startswith("encoder_cnn.cnn.") and not name.startswith("encoder_cnn.cnn.fc.")


def is_buffer(name):
    return name.endswith("running_mean") or name.endswith("running_var") or name.endswith("num_batches_tracked")


# ----------------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------------
def timing_signal(length, channels, min_timescale=1.0, max_timescale=1.0e4):
    """transformer_layers.py:542-558 — [sin | cos] concatenated, float64 then cast."""
    position = np.arange(length)
    num_timescales = channels // 2
    log_inc = math.log(float(max_timescale) / float(min_timescale)) / (float(num_timescales) - 1)
    inv = min_timescale * np.exp(np.arange(num_timescales).astype(np.float64) * -log_inc)
    scaled = np.expand_dims(position, 1) * np.expand_dims(inv, 0)
    signal = np.concatenate([np.sin(scaled), np.cos(scaled)], axis=1)
    signal = np.pad(signal, [[0, 0], [0, channels % 2]], "constant", constant_values=[0.0, 0.0])
    return torch.from_numpy(signal).type(torch.FloatTensor)


def pad_mask(ids):
    """transformer_layers.py:12-13."""
    return ids.eq(PAD).unsqueeze(1)


def _drop(x, masks, key, p):
    """nn.Dropout with an explicit keep-mask (1 = keep); masks=None or p=0 -> identity."""
    if masks is None or p == 0.0:
        return x
    return x * masks[key].to(x.dtype) / (1.0 - p)


def mha(P, pre, q_in, k_in, v_in, mask, num_heads, masks=None, key=None, p_drop=0.0):
    """MultiHeadAttention.forward, transformer_layers.py:486-532 (bias-free linears)."""
    Q = F.linear(q_in, P[pre + "query_linear.weight"])
    K = F.linear(k_in, P[pre + "key_linear.weight"])
    V = F.linear(v_in, P[pre + "value_linear.weight"])
    B, Tq, Hd = Q.shape
    d = Hd // num_heads

    def split(x):
        return x.view(x.shape[0], x.shape[1], num_heads, d).permute(0, 2, 1, 3)

    Q, K, V = split(Q), split(K), split(V)
    Q = Q * (d ** -0.5)                                       # :499
    logits = torch.matmul(Q, K.permute(0, 1, 3, 2))           # :502
    if mask is not None:
        logits = logits.masked_fill(mask.unsqueeze(1), -1e18)  # :504-506
    w = F.softmax(logits, dim=-1)                             # :517
    w = _drop(w, masks, key, p_drop)                          # :520
    ctx = torch.matmul(w, V)                                  # :523
    ctx = ctx.permute(0, 2, 1, 3).contiguous().view(B, Tq, Hd)
    return F.linear(ctx, P[pre + "output_linear.weight"])     # :530


def ffn(P, pre, x, masks=None, key=None, p_drop=0.0):
    """PositionwiseFeedForward.forward, transformer_layers.py:400-408: ReLU+dropout after BOTH linears."""
    for i in range(2):
        x = F.linear(x, P[pre + "layers.%d.weight" % i], P[pre + "layers.%d.bias" % i])
        x = F.relu(x)
        x = _drop(x, masks, None if key is None else "%s.ffn%d" % (key, i), p_drop)
    return x


def layer_norm(P, pre, x):
    return F.layer_norm(x, (x.shape[-1],), P[pre + ".weight"], P[pre + ".bias"], 1e-5)


def encoder(P, pre, x, mask, cfg, masks=None, p_drop=0.0):
    """Encoder.forward transformer_layers.py:138-152 + EncoderLayer.forward 260-282."""
    x = x + timing_signal(x.shape[1], x.shape[2]).to(x.dtype).unsqueeze(0)
    for l in range(cfg.num_layers):
        lp = "%s.enc.%d." % (pre, l)
        xn = layer_norm(P, lp + "layer_norm_mha", x)
        y = mha(P, lp + "multi_head_attention.", xn, xn, xn, mask, cfg.num_heads, masks, lp + "attn", p_drop)
        x = x + y
        xn = layer_norm(P, lp + "layer_norm_ffn", x)
        y = ffn(P, lp + "positionwise_feed_forward.", xn, masks, lp[:-1], p_drop)
        x = x + y
    return layer_norm(P, pre + ".layer_norm", x)


def decoder(P, pre, x, enc_out, mask_src, mask_trg, cfg, masks=None, p_drop=0.0):
    """Decoder.forward transformer_layers.py:205-221 + DecoderLayer.forward 326-364."""
    T = x.shape[1]
    causal = torch.triu(torch.ones(1, T, T, dtype=torch.uint8), diagonal=1)
    dec_mask = torch.gt(mask_trg.to(torch.uint8) + causal, 0)     # :207
    x = x + timing_signal(T, x.shape[2]).to(x.dtype).unsqueeze(0)  # :214
    for l in range(cfg.num_layers):
        lp = "%s.dec.%d." % (pre, l)
        xn = layer_norm(P, lp + "layer_norm_mha_dec", x)
        y = mha(P, lp + "multi_head_attention_dec.", xn, xn, xn, dec_mask, cfg.num_heads, masks, lp + "attn_dec", p_drop)
        x = x + y
        xn = layer_norm(P, lp + "layer_norm_mha_enc", x)
        y = mha(P, lp + "multi_head_attention_enc_dec.", xn, enc_out, enc_out, mask_src, cfg.num_heads, masks,
                lp + "attn_enc", p_drop)
        x = x + y
        xn = layer_norm(P, lp + "layer_norm_ffn", x)
        y = ffn(P, lp + "positionwise_feed_forward.", xn, masks, lp[:-1], p_drop)
        x = x + y
    return layer_norm(P, pre + ".layer_norm", x)


def embed(P, ids):
    """IQ.embedder iq.py:72-78: Embedding(V,E,padding_idx=0) -> Linear(E,H)."""
    e = F.embedding(ids, P["embedding.0.weight"], padding_idx=PAD)
    return F.linear(e, P["embedding.1.weight"], P["embedding.1.bias"])


def _bn2d(P, pre, x, train, buffers_out):
    w, b = P[pre + ".weight"], P[pre + ".bias"]
    rm, rv = P[pre + ".running_mean"], P[pre + ".running_var"]
    if train:
        rm2, rv2 = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm2, rv2, w, b, True, 0.1, 1e-5)
        if buffers_out is not None:
            buffers_out[pre + ".running_mean"] = rm2
            buffers_out[pre + ".running_var"] = rv2
            buffers_out[pre + ".num_batches_tracked"] = P[pre + ".num_batches_tracked"] + 1
        return y
    return F.batch_norm(x, rm, rv, w, b, False, 0.1, 1e-5)


def resnet18_features(P, pre, x, train=True, buffers_out=None):
    """torchvision resnet18 forward up to (and including) global avg-pool; encoder_cnn.py:33.
    The module is never put in eval() by the reference, so BatchNorm2d uses batch statistics."""
    x = F.conv2d(x, P[pre + "conv1.weight"], None, 2, 3)
    x = F.relu(_bn2d(P, pre + "bn1", x, train, buffers_out))
    x = F.max_pool2d(x, 3, 2, 1)
    cin = 64
    for lname, cout, stride in RESNET_LAYERS:
        for b in range(2):
            s = stride if b == 0 else 1
            bp = "%s%s.%d." % (pre, lname, b)
            idt = x
            out = F.conv2d(x, P[bp + "conv1.weight"], None, s, 1)
            out = F.relu(_bn2d(P, bp + "bn1", out, train, buffers_out))
            out = F.conv2d(out, P[bp + "conv2.weight"], None, 1, 1)
            out = _bn2d(P, bp + "bn2", out, train, buffers_out)
            if (bp + "downsample.0.weight") in P:
                idt = F.conv2d(x, P[bp + "downsample.0.weight"], None, s, 0)
                idt = _bn2d(P, bp + "downsample.1", idt, train, buffers_out)
            x = F.relu(out + idt)
            cin = cout
    return x.mean(dim=(2, 3))


def encoder_cnn(P, images, train=True, buffers_out=None):
    """EncoderCNN.forward encoder_cnn.py:30-35: resnet -> fc(512->H) -> BatchNorm1d(momentum 0.01)."""
    if "encoder_cnn.region_proj.weight" in P:
        # Bottom-up path (BASELINE configs[4]).  NO reference symbol exists (SURVEY A2'): the build's definition, as the survey proposes,
        # is mean over the regions of Linear(D -> H)(x_r), then the same BatchNorm1d.  images: [B, regions, D].  Parity for this path is
        # against this restatement only ("parity unpinned").
        p = F.linear(images, P["encoder_cnn.region_proj.weight"], P["encoder_cnn.region_proj.bias"])      # [B, regions, H]
        if "encoder_cnn.region_attn.weight" in P:
            # Region-attention pooling (no reference symbol either — README.md:2 only names the model "Bottom-Up"): a learned
            # vector scores every projected region, s_r = w_a . tanh(p_r); alpha = softmax over the regions; feature = sum_r alpha_r p_r
            sc = F.linear(torch.tanh(p), P["encoder_cnn.region_attn.weight"]).squeeze(-1)              # [B, regions]
            f = (torch.softmax(sc, dim=1).unsqueeze(-1) * p).sum(dim=1)
        else:
            f = p.mean(dim=1)
    else:
        pooled = resnet18_features(P, "encoder_cnn.cnn.", images, train, buffers_out)
        f = F.linear(pooled, P["encoder_cnn.cnn.fc.weight"], P["encoder_cnn.cnn.fc.bias"])
    rm, rv = P["encoder_cnn.bn.running_mean"].clone(), P["encoder_cnn.bn.running_var"].clone()
    y = F.batch_norm(f, rm, rv, P["encoder_cnn.bn.weight"], P["encoder_cnn.bn.bias"], train, 0.01, 1e-5)
    if train and buffers_out is not None:
        buffers_out["encoder_cnn.bn.running_mean"] = rm
        buffers_out["encoder_cnn.bn.running_var"] = rv
        buffers_out["encoder_cnn.bn.num_batches_tracked"] = P["encoder_cnn.bn.num_batches_tracked"] + 1
    return y


def gaussian_kld(mu_q, lv_q, mu_p, lv_p):
    """transformer_layers.py:536-540."""
    return -0.5 * torch.sum(1 + (lv_q - lv_p) - (mu_p - mu_q) ** 2 / torch.exp(lv_p)
                            - torch.exp(lv_q) / torch.exp(lv_p), dim=-1)


def _latent_mlp(P, pre, x):
    h = F.linear(x, P[pre + ".0.weight"], P[pre + ".0.bias"])
    h = F.linear(F.relu(h), P[pre + ".3.weight"], P[pre + ".3.bias"])
    return F.linear(F.relu(h), P[pre + ".6.weight"], P[pre + ".6.bias"])


def latent(P, x, x_p, eps, Z):
    """Latent.forward transformer_layers.py:41-59 (eps injected instead of torch.randn at :45)."""
    mlv_p = _latent_mlp(P, "latent_layer.mean_logvar_prior", x)
    mu_p, lv_p = mlv_p[:, :Z], mlv_p[:, Z:]
    if x_p is None:
        return 0, eps * torch.exp(0.5 * lv_p) + mu_p, (None, None)
    mlv_q = _latent_mlp(P, "latent_layer.mean_logvar_posterior", torch.cat((x_p, x), dim=-1))
    mu_q, lv_q = mlv_q[:, :Z], mlv_q[:, Z:]
    kld = torch.mean(gaussian_kld(mu_q, lv_q, mu_p, lv_p))
    z = eps * torch.exp(0.5 * lv_q) + mu_q
    return kld, z, (mu_q, lv_q)


def iq_forward(P, cfg, images, answers, response, target, phase2, eps=None, masks=None, p_drop=0.0,
               bn_train=True, buffers_out=None, image_features=None):
    """IQ.forward iq.py:82-114 with GVTransformerEncoder.forward encoder_transformer.py:22-37 and
    GVTransformerDecoder.forward decoder_transformer.py:22-41 inlined.
    Returns (output, z_logit, kld, (image_features, reconstructed), extras)."""
    if image_features is None:
        image_features = encoder_cnn(P, images, bn_train, buffers_out)
    # --- answer encoder ---
    res_mask = pad_mask(response)
    r_out = encoder(P, "answer_encoder.r_encoder", embed(P, response), res_mask, cfg, masks, p_drop)
    src_mask = pad_mask(answers)
    enc = encoder(P, "answer_encoder.encoder", embed(P, answers), src_mask, cfg, masks, p_drop)
    row0 = enc[:, 0] + image_features                                 # encoder_transformer.py:32
    enc = torch.cat([row0.unsqueeze(1), enc[:, 1:]], dim=1)
    kld, z = None, None
    if phase2:
        kld, z, _post = latent(P, enc[:, 0], r_out[:, 0], eps, cfg.latent_dim)
        z = F.linear(z, P["latent_projection.weight"], P["latent_projection.bias"])   # iq.py:105-106
    # --- decoder ---
    B = target.shape[0]
    sos = torch.full((B, 1), SOQ, dtype=target.dtype)
    shifted = torch.cat((sos, target[:, :-1]), 1)                     # decoder_transformer.py:24-27
    trg_mask = pad_mask(shifted)
    temb = embed(P, shifted)
    t0 = temb[:, 0] + image_features                                  # :31
    z_logit = None
    if phase2:
        t0 = t0 + z                                                   # :34
        z_logit = F.linear(z + image_features, P["decoder.z_classifier.weight"], P["decoder.z_classifier.bias"])
    temb = torch.cat([t0.unsqueeze(1), temb[:, 1:]], dim=1)
    dec = decoder(P, "decoder.decoder", temb, enc, src_mask, trg_mask, cfg, masks, p_drop)
    output = F.linear(dec, P["decoder.output.weight"], P["decoder.output.bias"])      # :40
    # --- image reconstruction (iq.py:109-112, mlp.py:49-56) ---
    r_in = enc[:, 0] + z if phase2 else enc[:, 0]
    h = F.relu(F.linear(r_in, P["image_reconstructor.layers.fc0.weight"], P["image_reconstructor.layers.fc0.bias"]))
    recon = F.linear(h, P["image_reconstructor.layers.fc1.weight"], P["image_reconstructor.layers.fc1.bias"])
    extras = dict(encoder_outputs=enc, r_encoder_outputs=r_out, decoder_outputs=dec, z=z)
    return output, z_logit, kld, (image_features, recon), extras


def decode_greedy(P, cfg, images, answers, phase2, eps=None, max_decode_length=50, bn_train=False):
    """IQ.decode_greedy iq.py:117-152 + GVTransformerDecoder.inference_forward decoder_transformer.py:43-48: the decoder is re-run
    on the growing prefix (no KV cache); ys starts as a single <pad>.  Returns (tokens [B,L+1], top6 idx [B,L+1,6], top6 probs)."""
    with torch.no_grad():
        feats = encoder_cnn(P, images, bn_train, None)
        src_mask = pad_mask(answers)
        enc = encoder(P, "answer_encoder.encoder", embed(P, answers), src_mask, cfg)
        enc = torch.cat([(enc[:, 0] + feats).unsqueeze(1), enc[:, 1:]], dim=1)
        z = 0
        if phase2:
            _, z, _ = latent(P, enc[:, 0], None, eps, cfg.latent_dim)
            z = F.linear(z, P["latent_projection.weight"], P["latent_projection.bias"])
        B = answers.shape[0]
        ys = torch.full((B, 1), PAD, dtype=torch.long)
        toks, tidx, tval = [], [], []
        for _ in range(max_decode_length + 1):
            emb = embed(P, ys)
            emb = torch.cat([(emb[:, 0] + z + feats).unsqueeze(1), emb[:, 1:]], dim=1)
            dec = decoder(P, "decoder.decoder", emb, enc, src_mask, pad_mask(ys), cfg)
            logits = F.linear(dec, P["decoder.output.weight"], P["decoder.output.bias"])[:, -1]
            nxt = logits.argmax(dim=1)
            v, i = torch.topk(F.softmax(logits, -1), 6, dim=1)
            toks.append(nxt); tidx.append(i); tval.append(v)
            ys = torch.cat([ys, nxt.unsqueeze(1)], dim=1)
        return torch.stack(toks, 1), torch.stack(tidx, 1), torch.stack(tval, 1)


# ----------------------------------------------------------------------------------
# losses / schedules / optimiser (train_iq.py)
# ----------------------------------------------------------------------------------
def kl_weight(kliter, full_kl_step):
    """train_iq.py:96-97."""
    return min(math.tanh(6 * kliter / full_kl_step - 3) + 1, 1)


def noam_lr(step, hidden_dim, warmup_steps=4000):
    """TrainIQ.custom_optimizer train_iq.py:252-257 (lr is 0 at step 0)."""
    a1 = math.sqrt(1 / (step + 1))
    a2 = step * (warmup_steps ** -1.5)
    return math.sqrt(1 / hidden_dim) * min(a1, a2)


def calculate_losses(output, image_recon, kld, z_logit, target, phase2, kliter, hp):
    """TrainIQ.calculate_losses train_iq.py:81-103.  hp: full_kl_step, kl_ceiling, aux_ceiling, image_recon_lambda.
    Returns (loss tensor, dict of python floats)."""
    loss_rec = F.cross_entropy(output.reshape(-1, output.size(-1)), target.reshape(-1), ignore_index=PAD)
    loss_img = F.mse_loss(image_recon[0], image_recon[1])
    if not phase2:
        loss = loss_rec + hp.image_recon_lambda * loss_img
        elbo, aux, kl = loss_rec, 0.0, 0.0
    else:
        zl = z_logit.unsqueeze(1).repeat(1, output.size(1), 1)
        loss_aux = F.cross_entropy(zl.reshape(-1, zl.size(-1)), target.reshape(-1), ignore_index=PAD)
        w = kl_weight(kliter, hp.full_kl_step)
        aux = loss_aux.item()
        elbo = loss_rec + kld
        kl = kld.item()
        loss = loss_rec + hp.kl_ceiling * w * kld + hp.aux_ceiling * loss_aux + hp.image_recon_lambda * loss_img
    stats = dict(loss=loss.item(), rec=loss_rec.item(), img=loss_img.item(),
                 ppl=math.exp(min(loss_rec.item(), 100)), kld=kl, aux=aux, elbo=float(elbo.detach()) if torch.is_tensor(elbo) else float(elbo))
    return loss, stats


def default_hp(**kw):
    hp = SimpleNamespace(full_kl_step=15000, kl_ceiling=0.5, aux_ceiling=1.0, image_recon_lambda=0.1,
                         num_pretraining_steps=12000, clip=5.0)
    hp.__dict__.update(kw)
    return hp


def clone_params(state, requires_grad=True):
    P = {}
    for k, v in state.items():
        t = v.detach().clone()
        if requires_grad and t.is_floating_point() and not is_frozen(k) and not is_buffer(k):
            t.requires_grad_(True)
        P[k] = t
    return P


def trainable_names(P):
    return [k for k, v in P.items() if v.requires_grad]


def train_steps(state, cfg, batches, hp, start_iter=0, start_kliter=0, p_drop=0.0, masks_per_step=None):
    """Runs len(batches) full reference training steps (training_step train_iq.py:105-132 + Lightning's
    backward / clip_grad_norm_(5) / Adam.step, train_iq.py:260,372) on CPU.  Each batch is a dict with
    images, answers, posteriors, questions, eps.  Returns (final state dict, list of stat dicts)."""
    P = clone_params(state)
    names = trainable_names(P)
    opt = torch.optim.Adam([P[n] for n in names], lr=3e-5)
    it, kliter = start_iter, start_kliter
    phase2 = False
    logs = []
    for si, b in enumerate(batches):
        if it == hp.num_pretraining_steps:
            phase2 = True
        if it > hp.num_pretraining_steps:
            phase2 = True
        bufs = {}
        out, z_logit, kld, recon, _ = iq_forward(P, cfg, b["images"], b["answers"], b["posteriors"], b["questions"],
                                                 phase2, b.get("eps"), None if masks_per_step is None else masks_per_step[si],
                                                 p_drop, True, bufs)
        loss, stats = calculate_losses(out, recon, kld, z_logit, b["questions"], phase2, kliter, hp)
        if phase2:
            kliter += 1
        opt.zero_grad(set_to_none=True)
        loss.backward()
        used = [P[n] for n in names if P[n].grad is not None]
        gn = torch.nn.utils.clip_grad_norm_(used, hp.clip)
        stats["grad_norm"] = float(gn)
        # Lightning runs optimizer.step() with the lr set by the PREVIOUS training_step's custom_optimizer
        # call; custom_optimizer(self.iter) runs inside training_step before backward/step (train_iq.py:130),
        # so the step taken for iteration `it` uses noam_lr(it).
        lr = noam_lr(it, cfg.hidden_dim)
        for g in opt.param_groups:
            g["lr"] = lr
        stats["lr"] = lr
        opt.step()
        with torch.no_grad():
            for k, v in bufs.items():
                P[k] = v.detach()
        it += 1
        logs.append(stats)
    return {k: v.detach() for k, v in P.items()}, logs
