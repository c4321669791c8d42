"""Oracle: the fusion arithmetic of the reference, restated functionally (torch CPU fp32).

TEST INFRASTRUCTURE (see oracle/__init__.py).  Every function cites the reference lines it
follows.  Weights come in as a dict keyed by the reference's ``ERN.state_dict()`` names.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

PATCH_NUM = 13
BERT_HEADS = 8
MHA_HEADS = 8


def as_torch(sd: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v)
            for k, v in sd.items()}


def _lin(sd, prefix, x):
    return F.linear(x, sd[prefix + ".weight"], sd[prefix + ".bias"])


def _ln(sd, prefix, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], eps)


def combiner_simple(sd, prefix, image_features, text_features):
    """CombinerSimple.forward -- /root/reference/models/fusion_model.py:86-94 (eval: dropouts off)."""
    tp = F.relu(_lin(sd, prefix + ".text_projection_layer.0", text_features))
    ip = F.relu(_lin(sd, prefix + ".image_projection_layer.0", image_features))
    raw = torch.cat((tp, ip), dim=-1)                       # :90 text first, image second
    h = F.relu(_lin(sd, prefix + ".dynamic_scalar.0", raw))
    s = torch.sigmoid(_lin(sd, prefix + ".dynamic_scalar.3", h))
    out = s * text_features + (1 - s) * image_features      # :93
    return F.normalize(out, dim=-1)                         # :94 (eps 1e-12 clamp)


def _bn_eval(sd, prefix, x, channel_dim):
    """BatchNorm1d in eval mode; channel axis = 1 for 3-D input, = last for 2-D (torch semantics)."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    shape = [1] * x.dim()
    shape[channel_dim] = -1
    return (x - rm.view(shape)) / torch.sqrt(rv.view(shape) + 1e-5) * w.view(shape) + b.view(shape)


def visual_sr(sd, prefix, local):
    """VisualSR.forward -- fusion_model.py:141-154; BatchNorm1d(13) normalises over the *patch* axis."""
    assert local.shape[1] == PATCH_NUM
    raw_global = local.mean(dim=1)                                                   # :142
    l_emb = torch.tanh(_bn_eval(sd, prefix + ".embedding_local.1",
                                _lin(sd, prefix + ".embedding_local.0", local), 1))  # :145 (:117-123)
    g_emb = torch.tanh(_bn_eval(sd, prefix + ".embedding_global.1",
                                _lin(sd, prefix + ".embedding_global.0", raw_global), 1))
    common = l_emb * g_emb.unsqueeze(1)                                              # :149
    logits = _lin(sd, prefix + ".embedding_common", common).squeeze(2)
    w = torch.softmax(logits, dim=1)                                                 # :150
    new_global = (w.unsqueeze(2) * local).sum(dim=1)                                 # :153
    norm = torch.sqrt((new_global ** 2).sum(dim=-1, keepdim=True)) + 1e-8            # :136-139
    return new_global / norm


def _r(x, reduced):
    """bfloat16 rounding (RNE) carried as fp32: the operand rounding of the product's reduced-precision modes."""
    return x.bfloat16().float() if reduced else x


def bert_encode(sd, prefix, x0, n_type0, precision="fp32"):
    """HF BertModel(inputs_embeds=x0, token_type_ids=[0]*n_type0+[1]*rest, mask=1) -> last_hidden_state.

    ``precision`` "bf16" / "fp8" restate the product's reduced-precision modes (include/fern.h:fern_precision), in which the
    two BERT blocks follow the towers' bf16 recipe: bf16 operands on query/key/value, attention.output.dense,
    intermediate.dense and output.dense (activations AND weights rounded, fp32 sums), the packed projection stored as bf16,
    attention in the bf16 operand form (fp32 scores scaled after the product, un-normalised weights rounded to bf16 for P V,
    normaliser from the un-rounded weights, output stored as bf16), GELU output stored as bf16; residual stream and
    LayerNorm in fp32.

    Call site fusion_model.py:199-212; config fusion_model.py:162-170 (post-LN, eps 1e-12,
    GELU(erf), absolute positions, intermediate 3072, 8 heads).  The arithmetic itself lives in
    third-party ``transformers`` (pinned 4.30.2, environment.yml:156).
    """
    red = precision != "fp32"
    b, s, d = x0.shape
    hd = d // BERT_HEADS
    tok = torch.cat((torch.zeros(n_type0, dtype=torch.long), torch.ones(s - n_type0, dtype=torch.long)))
    x = x0 + sd[prefix + ".embeddings.token_type_embeddings.weight"][tok]
    x = x + sd[prefix + ".embeddings.position_embeddings.weight"][:s]
    x = _ln(sd, prefix + ".embeddings.LayerNorm", x, 1e-12)
    layer = 0
    while f"{prefix}.encoder.layer.{layer}.attention.self.query.weight" in sd:
        lp = f"{prefix}.encoder.layer.{layer}"

        def lin(name, inp):
            return F.linear(_r(inp, red), _r(sd[lp + name + ".weight"], red), sd[lp + name + ".bias"])

        q = _r(lin(".attention.self.query", x), red).view(b, s, BERT_HEADS, hd).transpose(1, 2)
        k = _r(lin(".attention.self.key", x), red).view(b, s, BERT_HEADS, hd).transpose(1, 2)
        v = _r(lin(".attention.self.value", x), red).view(b, s, BERT_HEADS, hd).transpose(1, 2)
        if red:
            att = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(hd))
            e = torch.exp(att - att.max(dim=-1, keepdim=True).values)
            ctx = _r(((_r(e, True) @ v) / e.sum(dim=-1, keepdim=True)).transpose(1, 2).reshape(b, s, d), True)
        else:
            p = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd), dim=-1)
            ctx = (p @ v).transpose(1, 2).reshape(b, s, d)
        x = _ln(sd, lp + ".attention.output.LayerNorm", lin(".attention.output.dense", ctx) + x, 1e-12)
        h = _r(F.gelu(lin(".intermediate.dense", x)), red)
        x = _ln(sd, lp + ".output.LayerNorm", lin(".output.dense", h) + x, 1e-12)
        layer += 1
    return x


def mha_forward(sd, prefix, query, key, value, heads=MHA_HEADS):
    """nn.MultiheadAttention(batch_first=True) eval forward -- constructed fusion_model.py:18-20."""
    d = query.shape[-1]
    hd = d // heads
    w, bias = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    q = F.linear(query, w[:d], bias[:d])
    k = F.linear(key, w[d:2 * d], bias[d:2 * d])
    v = F.linear(value, w[2 * d:], bias[2 * d:])
    b, sq, _ = q.shape
    sk = k.shape[1]
    q = q.view(b, sq, heads, hd).transpose(1, 2) * (1.0 / math.sqrt(hd))
    k = k.view(b, sk, heads, hd).transpose(1, 2)
    v = v.view(b, sk, heads, hd).transpose(1, 2)
    p = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
    o = (p @ v).transpose(1, 2).reshape(b, sq, d)
    return _lin(sd, prefix + ".out_proj", o)


def dvr_fuse(sd, ref_patch, text_seq, ref_global, text_global, prefix="DVR", precision="fp32"):
    """DVR_module.forward -- fusion_model.py:26-55 (= ERN mode="test", models/model.py:68-69).  ``precision``: see bert_encode
    (only the two BERT blocks change; cross attention, VisualSR and the combiners are fp32 in every mode)."""
    b, p, d = ref_patch.shape
    tl = prefix + ".transformer_layer"
    cls = sd.get(tl + ".cls_token")                  # absent in GPU-trained checkpoints -> zeros
    if cls is None:
        cls = torch.zeros(1, 1, d)
    x0 = torch.cat((cls.expand(b, -1, -1), ref_patch, text_seq), dim=1)             # :199-201
    hidden = bert_encode(sd, tl + ".bert_encoder.bert_model", x0, p + 1, precision)  # :202-212
    img = F.normalize(hidden[:, 1:p + 1], dim=2)                                    # :38,40
    txt = F.normalize(hidden[:, p + 1:], dim=2)                                     # :39,41
    cross = mha_forward(sd, prefix + ".MR_component", txt, img, img)[:, :p]         # :44-47
    patch_vision_mean = visual_sr(sd, prefix + ".SR_module", cross)                 # :48
    seq_text_mean = txt.mean(dim=1)                                                 # :49
    g = combiner_simple(sd, prefix + ".combiner_global", ref_global, text_global)   # :52
    l = combiner_simple(sd, prefix + ".combiner_local", patch_vision_mean, seq_text_mean)  # :53
    return combiner_simple(sd, prefix + ".combiner", g, l)                          # :54


def index_fuse(sd, tar_feats, tar_local):
    """ERN mode="index" -- models/model.py:64-66 (caller normalises tar_feats first, test_fiq.py:45)."""
    sr = visual_sr(sd, "SR_module", tar_local)
    return combiner_simple(sd, "Combiner_module", tar_feats, sr)


def combiner_clip4cir(sd, prefix, image_features, text_features):
    """CLIP4Cir Combiner.forward -- /root/reference/models/others/Combiner_Model.py:37-70 (eval: dropouts off)."""
    p = prefix + "." if prefix else ""
    tp = F.relu(F.linear(text_features, sd[p + "text_projection_layer.weight"], sd[p + "text_projection_layer.bias"]))
    ip = F.relu(F.linear(image_features, sd[p + "image_projection_layer.weight"], sd[p + "image_projection_layer.bias"]))
    raw = torch.cat((tp, ip), -1)                                                              # :56-58
    comb = F.relu(F.linear(raw, sd[p + "combiner_layer.weight"], sd[p + "combiner_layer.bias"]))
    h = F.relu(F.linear(raw, sd[p + "dynamic_scalar.0.weight"], sd[p + "dynamic_scalar.0.bias"]))
    s = torch.sigmoid(F.linear(h, sd[p + "dynamic_scalar.3.weight"], sd[p + "dynamic_scalar.3.bias"]))
    out = F.linear(comb, sd[p + "output_layer.weight"], sd[p + "output_layer.bias"]) + s * text_features + (1 - s) * image_features
    return F.normalize(out, dim=-1)                                                            # :69


def element_wise_sum(image_features, text_features):
    """utils.element_wise_sum -- /root/reference/utils/utils.py:133-140."""
    return F.normalize(image_features + text_features, dim=-1)


def batch_classification_loss(predicted, target):
    """BatchBasedClassificationLoss.forward -- /root/reference/losses/loss.py:10-14."""
    logits = 100 * predicted @ target.T
    return F.cross_entropy(logits, torch.arange(predicted.shape[0]))
