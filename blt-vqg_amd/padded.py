"""Padded parameter layout for model widths the HIP kernels do not tile directly.

The reference's CLI defaults (train_iq.py:315-325) are hidden_dim = latent_dim = 300, pwffn_dim = 600, num_heads = 4: heads of 75
features.  The kernels move activations in 16-byte vectors and the attention kernels tile the head width, so the engine is created with
PADDED widths — every head in a slot of round8(75) = 80 columns (hidden 320), latent 304 (FFN 600 is a multiple of 8 already) — and told the real head width
(bltvqg_config.head_dim_true) for the few places where the width enters the arithmetic (LayerNorm statistics, attention scale, MSE mean,
timing signal).  Everything else follows from ZEROS: pad rows / columns of every weight are zero, so pad activations are zero, so pad
gradients are zero (Adam leaves a zero parameter with zero gradient at zero).

This module only maps between the two layouts: the module's nn.Parameters keep the reference's shapes (state_dict / checkpoints
unchanged) in a flat buffer of their own; `train_index[i]` is where element i of that buffer lives in the engine's padded flat buffer.
"""
from collections import OrderedDict

import torch

from .engine import ParamInfo


def round_up(x, m):
    return (x + m - 1) // m * m


def needs_padding(hidden_dim, latent_dim, pwffn_dim, num_heads):
    return hidden_dim % 8 != 0 or latent_dim % 8 != 0 or pwffn_dim % 8 != 0 or (hidden_dim // num_heads) % 4 != 0


class PaddedLayout(object):
    def __init__(self, hidden_dim, latent_dim, pwffn_dim, num_heads):
        H, Z, F, NH = int(hidden_dim), int(latent_dim), int(pwffn_dim), int(num_heads)
        if H % NH != 0 or (H % 2) != 0:
            raise ValueError("hidden_dim (%d) must be even and divisible by num_heads (%d)" % (H, NH))
        self.H, self.Z, self.F, self.NH = H, Z, F, NH
        self.dh = H // NH
        self.dh_p = round_up(self.dh, 8)
        self.Hp, self.Zp, self.Fp = NH * self.dh_p, round_up(Z, 8), round_up(F, 8)
        i = torch.arange(H)
        mapH = (i // self.dh) * self.dh_p + i % self.dh
        j = torch.arange(2 * Z)
        self.maps = {
            "H": (mapH, self.Hp), "F": (torch.arange(F), self.Fp), "Z": (torch.arange(Z), self.Zp),
            "2Z": (torch.where(j < Z, j, j - Z + self.Zp), 2 * self.Zp),                 # [mean | logvar] halves of the latent nets
            "2H": (torch.cat([mapH, mapH + self.Hp]), 2 * self.Hp),                        # cat(x_p, x) input of the posterior net
        }

    # which logical width each dimension of a parameter has (None = not padded: vocabulary, embedding, CNN channels)
    def kinds(self, name, ndim):
        n = name
        if n.endswith(("query_linear.weight", "key_linear.weight", "value_linear.weight", "output_linear.weight")):
            return ("H", "H")
        if "positionwise_feed_forward.layers.0." in n or n.startswith("image_reconstructor.layers.fc0."):
            return ("F", "H") if ndim == 2 else ("F",)
        if "positionwise_feed_forward.layers.1." in n or n.startswith("image_reconstructor.layers.fc1."):
            return ("H", "F") if ndim == 2 else ("H",)
        if "layer_norm" in n:
            return ("H",)
        if n.startswith("embedding.1."):
            return ("H", None) if ndim == 2 else ("H",)
        if n in ("decoder.output.weight", "decoder.z_classifier.weight"):
            return (None, "H")
        if n.startswith("latent_projection."):
            return ("H", "Z") if ndim == 2 else ("H",)
        if n.startswith("latent_layer.mean_logvar_"):
            first_in = "2H" if "posterior" in n else "H"
            if ndim == 1:
                return ("2Z",)
            return ("2Z", first_in if n.endswith(".0.weight") else "2Z")
        if n in ("encoder_cnn.cnn.fc.weight", "encoder_cnn.region_proj.weight"):
            return ("H", None)
        if n == "encoder_cnn.region_attn.weight":
            return (None, "H")
        if n in ("encoder_cnn.cnn.fc.bias", "encoder_cnn.region_proj.bias", "encoder_cnn.bn.weight", "encoder_cnn.bn.bias",
                 "encoder_cnn.bn.running_mean", "encoder_cnn.bn.running_var"):
            return ("H",)
        return (None,) * ndim

    def true_shape(self, name, padded_shape):
        sizes = {"H": self.H, "F": self.F, "Z": self.Z, "2Z": 2 * self.Z, "2H": 2 * self.H}
        kinds = self.kinds(name, len(padded_shape))
        out = []
        for k, p in zip(kinds, padded_shape):
            if k is not None and self.maps[k][1] != p:
                raise ValueError("padded layout: %s has dimension %d where %d was expected" % (name, p, self.maps[k][1]))
            out.append(sizes[k] if k is not None else int(p))
        return tuple(out)

    def build(self, engine_infos):
        """engine_infos: OrderedDict name -> ParamInfo of the PADDED engine.  Returns (true infos with offsets into the module's own flat
        buffer, index LongTensor: position in that buffer -> position in the engine's flat buffer, size of the module buffer)."""
        infos, pieces, off = OrderedDict(), [], 0
        for name, pi in engine_infos.items():
            shape = self.true_shape(name, pi.shape)
            kinds = self.kinds(name, len(shape))
            numel = 1
            for d in shape:
                numel *= d
            if len(shape) in (1, 2) and any(k is not None for k in kinds):
                rmap = self.maps[kinds[0]][0] if kinds[0] is not None else torch.arange(shape[0])
                if len(shape) == 2:
                    cmap = self.maps[kinds[1]][0] if kinds[1] is not None else torch.arange(shape[1])
                    idx = (rmap[:, None] * pi.shape[1] + cmap[None, :]).reshape(-1)
                else:
                    idx = rmap
            else:
                idx = torch.arange(numel)
            pieces.append(idx + pi.offset)
            infos[name] = ParamInfo(name, off, numel, shape, pi.late)
            off += numel
        return infos, torch.cat(pieces).long(), off
