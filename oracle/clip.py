"""Oracle: CLIP ViT image tower and text tower (torch CPU fp32).  TEST INFRASTRUCTURE.

PARITY UNPINNED by the reference's own call path: ``clip_model.encode_image`` /
``encode_text`` are called at /root/reference/utils/utils.py:64 and
/root/reference/run/test/test_fiq.py:102-103 on an object built by third-party
open-clip-torch==2.20.0 (environment.yml:114) plus the authors' unreleased text encoder
(README.md:41); neither is in /root/reference.  The arithmetic restated here follows the one
in-tree statement of CLIP, /root/reference/models/others/modeling_clip.py (cited per step), the
open_clip "ViT-B-16" shape (exact-erf GELU), and open_clip's state-dict key names.
``tools/make_goldens.py`` executes that in-tree file on the same weights to pin this module.

Definition of the un-pinned ``encode_text(text, mode=, visual_emb=)``: one
text-tower pass; ``seq = ln_final(h) @ text_projection`` [B,77,D]; ``global = seq[b, argmax(text[b])]``;
default/"global" mode returns ``(global, seq)``, "seq" returns ``seq``; ``visual_emb`` is
shape-checked and ignored ("vanilla CLIP single branch", README.md:41).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def _ln(sd, prefix, x):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + ".weight"], sd[prefix + ".bias"], 1e-5)


def _r(x, reduced):
    """Operand rounding of the product's reduced encoder precisions (include/fern.h:fern_precision): round to nearest even
    to bfloat16, carried on as fp32 so that products are exact and sums accumulate in fp32, like the MFMA."""
    return x.bfloat16().float() if reduced else x


_INV448 = torch.tensor(1.0 / 448.0, dtype=torch.float32)


def _q8(x):
    """Per-row fp8 quantisation as the product does it (csrc/elem.hip:quantize_rows_fp8_kernel): scale = max|row| * (1/448)
    (1 for a zero row), q = e4m3fn(x * (1 / scale)), round to nearest even.  Returns (q as fp32, scale)."""
    am = x.abs().amax(dim=-1, keepdim=True)
    sc = torch.where(am > 0, am * _INV448, torch.ones_like(am))
    q = (x * (1.0 / sc)).to(torch.float8_e4m3fn).float()
    return q, sc


def mx8_quantize(x):
    """MX (block-scaled) e4m3fn quantisation as the product does it (csrc/elem.hip: quantize_mx8_kernel / layernorm_mx8_kernel;
    include/fern.h: fern_quantize_mx8): per (row, 32 consecutive k) one E8M0 byte e = the smallest power-of-two exponent with
    max|block| * 2^-(e-127) <= 448, read off the maximum's exponent and mantissa bits (448 = 1.75 * 2^8), clamped to [1, 253];
    q = e4m3fn(x * 2^(127-e)) -- exact scaling, round-to-nearest-even cast.  Returns (q [.., D] float8_e4m3fn, e [.., D/32] uint8)."""
    shp = x.shape
    blocks = x.float().reshape(*shp[:-1], shp[-1] // 32, 32)
    u = blocks.abs().amax(dim=-1).contiguous().view(torch.int32)
    e = ((u >> 23) - 8 + ((u & 0x7FFFFF) > 0x600000).int()).clamp(1, 253)
    inv = ((254 - e) << 23).view(torch.float32)
    q = (blocks * inv.unsqueeze(-1)).to(torch.float8_e4m3fn)
    return q.reshape(shp), e.to(torch.uint8)


def mx8_dequantize(q, e, dtype=torch.float32):
    """The values the block-scaled MFMA multiplies: q * 2^(e-127) per 32-k block (exact in fp32)."""
    shp = q.shape
    sc = torch.ldexp(torch.ones((), dtype=dtype), e.int() - 127)
    return (q.to(dtype).reshape(*shp[:-1], shp[-1] // 32, 32) * sc.unsqueeze(-1)).reshape(shp)


def _linear(x, w, b, prec):
    """prec None / "fp32": plain; "bf16": bf16-rounded operands; "fp8": e4m3fn operands with per-token and per-output-channel
    scales folded back after the fp32-accumulated product (the product's epilogue: acc * (sa * sw) + bias)."""
    if prec == "mx8":
        y = mx8_dequantize(*mx8_quantize(x)) @ mx8_dequantize(*mx8_quantize(w)).transpose(-1, -2)
        return y if b is None else y + b
    if prec == "fp8":
        qa, sa = _q8(x)
        qw, sw = _q8(w)
        y = (qa @ qw.transpose(-1, -2)) * (sa * sw.transpose(-1, -2))
        return y if b is None else y + b
    return F.linear(_r(x, prec == "bf16"), _r(w, prec == "bf16"), b)


def _attention(sd, prefix, x, heads, causal, prec=None, q_rows=None):
    """modeling_clip.py:272-336: q scaled by hd**-0.5 before QK^T, additive causal mask, softmax, PV.

    ``prec`` "bf16" / "fp8": the packed in-projection and the out-projection take reduced-precision operands and the
    attention itself runs in the product's bf16 operand form.  ``q_rows`` (last ViT block of the product: only the class row
    is consumed): queries, out-projection and everything after them are evaluated for those rows only and in fp32; K/V
    still come from every token (reduced-precision operands, fp32 result)."""
    b, s, w = x.shape
    hd = w // heads
    reduced = prec in ("bf16", "fp8", "mx8")
    wi, bi = sd[prefix + ".in_proj_weight"], sd[prefix + ".in_proj_bias"]
    if q_rows is None:
        qkv = _r(_linear(x, wi, bi, prec), reduced)       # reduced precision: the packed projection is STORED as bf16
        q, k, v = qkv.split(w, dim=-1)
        if reduced:
            # bf16 operand attention of the product (include/fern.h:fern_attention_bf16): fp32 scores from bf16 q, k, scaled
            # after the product; un-normalised weights rounded to bf16 for P V, normaliser from the un-rounded weights;
            # output stored as bf16.  (The product rounds exp(s - running max) tile by tile; here the row max is used, which
            # moves individual roundings by at most one bf16 ulp.)
            q = q.view(b, s, heads, hd).transpose(1, 2)
            k = k.view(b, s, heads, hd).transpose(1, 2)
            v = v.view(b, s, heads, hd).transpose(1, 2)
            att = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
            if causal:
                att = att + torch.full((s, s), float("-inf")).triu(1)
            e = torch.exp(att - att.max(dim=-1, keepdim=True).values)
            o = (_r(e, True) @ v) / e.sum(dim=-1, keepdim=True)
            # stored as bf16 -- in the block-scaled mode the kernel quantises the fp32 values itself (head_dim % 32 == 0)
            o = _r(o.transpose(1, 2).reshape(b, s, w), not (prec == "mx8" and hd % 32 == 0))
            return _linear(o, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"], prec)
    else:
        k, v = _linear(x, wi[w:], bi[w:], prec).split(w, dim=-1)
        q = F.linear(x[:, q_rows], wi[:w], bi[:w])
        s_q = q.shape[1]
        q = q.view(b, s_q, heads, hd).transpose(1, 2) * (hd ** -0.5)
        k = k.view(b, s, heads, hd).transpose(1, 2)
        v = v.view(b, s, heads, hd).transpose(1, 2)
        o = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(b, s_q, w)
        return F.linear(o, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])
    q = q.view(b, s, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = k.view(b, s, heads, hd).transpose(1, 2)
    v = v.view(b, s, heads, hd).transpose(1, 2)
    att = q @ k.transpose(-1, -2)
    if causal:                                            # modeling_clip.py:677-691
        att = att + torch.full((s, s), float("-inf")).triu(1)
    att = torch.softmax(att, dim=-1)
    o = (att @ v).transpose(1, 2).reshape(b, s, w)
    return F.linear(o, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])


def _block(sd, prefix, x, heads, causal, prec=None):
    """Pre-LN residual block, modeling_clip.py:354-401; MLP :339-351 with exact GELU."""
    reduced = prec in ("bf16", "fp8", "mx8")
    # the block-scaled mode keeps the residual stream itself in bf16 (csrc/api.hip: clip_block_mx8): one rounding per residual add
    stream_bf16 = prec == "mx8"
    x = _r(x + _attention(sd, prefix + ".attn", _ln(sd, prefix + ".ln_1", x), heads, causal, prec), stream_bf16)
    h = F.gelu(_linear(_ln(sd, prefix + ".ln_2", x), sd[prefix + ".mlp.c_fc.weight"], sd[prefix + ".mlp.c_fc.bias"], prec))
    # h is stored as bf16 -- except in the block-scaled mode, whose c_fc GEMM quantises its fp32 GELU output in the epilogue
    return _r(x + _linear(_r(h, reduced and prec != "mx8"), sd[prefix + ".mlp.c_proj.weight"], sd[prefix + ".mlp.c_proj.bias"], prec), stream_bf16)


def _block_cls(sd, prefix, x, heads, prec):
    """The last ViT block as the product evaluates it under a reduced precision: same arithmetic as ``_block`` restricted
    to the class row (the only row ln_post reads, modeling_clip.py:876-877) with fp32 operands, except the K/V projection
    of all tokens, which takes the reduced-precision operands."""
    c = x[:, :1] + _attention(sd, prefix + ".attn", _ln(sd, prefix + ".ln_1", x), heads, False, prec, q_rows=slice(0, 1))
    h = F.gelu(F.linear(_ln(sd, prefix + ".ln_2", c), sd[prefix + ".mlp.c_fc.weight"], sd[prefix + ".mlp.c_fc.bias"]))
    return c + F.linear(h, sd[prefix + ".mlp.c_proj.weight"], sd[prefix + ".mlp.c_proj.bias"])


def _bn(sd, prefix, x):
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"], sd[prefix + ".bias"],
                        training=False, eps=1e-5)


def encode_image_resnet(sd, cfg, images):
    """open_clip 2.20.0 ``ModifiedResNet`` (third party, NOT under /root/reference: environment.yml:114; selected by
    ``--clip-model-name RN50x4``, run/test/test_cirr.py:149).  PARITY UNPINNED: there is no statement of this tower in
    the reference tree and the package is not installed, so the published architecture is restated here:
    3-conv stem (3x3 s2, 3x3, 3x3; BN + ReLU each) + 2x2 avg-pool; Bottleneck blocks (1x1 -> 3x3 -> [avg-pool(stride)]
    -> 1x1 x4, BN after every conv, avg-pool + 1x1 + BN shortcut, ReLU after the add); AttentionPool2d (mean token
    prepended, learned positions, one multi-head attention read-out of the mean token, c_proj to embed_dim)."""
    x = F.relu(_bn(sd, "visual.bn1", F.conv2d(images, sd["visual.conv1.weight"], stride=2, padding=1)))
    x = F.relu(_bn(sd, "visual.bn2", F.conv2d(x, sd["visual.conv2.weight"], padding=1)))
    x = F.relu(_bn(sd, "visual.bn3", F.conv2d(x, sd["visual.conv3.weight"], padding=1)))
    x = F.avg_pool2d(x, 2)
    for li, nblocks in enumerate(cfg.r_layers):
        for bi in range(nblocks):
            p = f"visual.layer{li + 1}.{bi}"
            stride = 2 if (bi == 0 and li > 0) else 1
            out = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"])))
            out = F.relu(_bn(sd, p + ".bn2", F.conv2d(out, sd[p + ".conv2.weight"], padding=1)))
            if stride > 1:
                out = F.avg_pool2d(out, stride)
            out = _bn(sd, p + ".bn3", F.conv2d(out, sd[p + ".conv3.weight"]))
            identity = x
            if p + ".downsample.0.weight" in sd:
                identity = F.avg_pool2d(x, stride) if stride > 1 else x
                identity = _bn(sd, p + ".downsample.1", F.conv2d(identity, sd[p + ".downsample.0.weight"]))
            x = F.relu(out + identity)
    b, c, h, w = x.shape
    t = x.reshape(b, c, h * w).permute(0, 2, 1)                          # [b, HW, C]
    t = torch.cat((t.mean(dim=1, keepdim=True), t), dim=1) + sd["visual.attnpool.positional_embedding"]
    heads = cfg.r_heads
    hd = c // heads
    a = "visual.attnpool."
    q = F.linear(t[:, :1], sd[a + "q_proj.weight"], sd[a + "q_proj.bias"]).view(b, 1, heads, hd).transpose(1, 2) * (hd ** -0.5)
    k = F.linear(t, sd[a + "k_proj.weight"], sd[a + "k_proj.bias"]).view(b, -1, heads, hd).transpose(1, 2)
    v = F.linear(t, sd[a + "v_proj.weight"], sd[a + "v_proj.bias"]).view(b, -1, heads, hd).transpose(1, 2)
    o = (torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v).transpose(1, 2).reshape(b, c)
    return F.linear(o, sd[a + "c_proj.weight"], sd[a + "c_proj.bias"])


def encode_image(sd, cfg, images, precision="fp32"):
    """[b,3,H,W] f32 -> [b,embed_dim] un-normalised (call site utils/utils.py:64).

    ``precision="bf16"`` / ``"fp8"`` / ``"mx8"`` restate the product's reduced-precision modes (no reference counterpart;
    include/fern.h:fern_precision): the same arithmetic with the operands of the token-level block GEMMs rounded to
    bfloat16, quantised to e4m3fn with per-token / per-channel scales, or to e4m3fn with one E8M0 scale per 32-element block."""
    if precision not in ("fp32", "bf16", "fp8", "mx8"):
        raise ValueError(precision)
    bf16 = precision != "fp32"
    prec = None if precision == "fp32" else precision
    if getattr(cfg, "v_arch", "vit") == "resnet":
        if bf16:
            raise ValueError("reduced precisions are defined for the transformer towers only")
        return encode_image_resnet(sd, cfg, images)
    w = sd["visual.conv1.weight"]
    if precision == "mx8" and w[0].numel() % 128 == 0 and w[0].numel() <= 1280:
        # the block-scaled mode runs conv1 as a linear layer over patch rows ((channel, y, x) order) with block-scaled operands
        patches = F.unfold(images, kernel_size=cfg.patch_size, stride=cfg.patch_size).transpose(1, 2)    # [b, g*g, 3*P*P]
        x = _linear(patches, w.flatten(1), None, "mx8")
    elif precision == "bf16" and w[0].numel() % 32 == 0 and w[0].numel() <= 1280:
        # the bf16-operand modes (round 6) run conv1 as a linear layer over bf16-rounded patch rows and the bf16 copy of its weight
        patches = F.unfold(images, kernel_size=cfg.patch_size, stride=cfg.patch_size).transpose(1, 2)    # [b, g*g, 3*P*P]
        x = _linear(patches, w.flatten(1), None, "bf16")
    else:
        x = F.conv2d(images, w, stride=cfg.patch_size)                  # modeling_clip.py:180-196
        x = x.flatten(2).transpose(1, 2)                                # [b, g*g, width]
    cls = sd["visual.class_embedding"].expand(x.shape[0], 1, -1)
    x = torch.cat((cls, x), dim=1) + sd["visual.positional_embedding"]  # :197-200
    x = _ln(sd, "visual.ln_pre", x)                                     # :839,866
    if precision == "mx8" and cfg.v_layers > 1:
        x = _r(x, True)                                                 # ln_pre writes the mode's bf16 residual stream
    for i in range(cfg.v_layers - (1 if bf16 else 0)):
        x = _block(sd, f"visual.transformer.resblocks.{i}", x, cfg.v_heads, causal=False, prec=prec)
    if bf16:
        x = _block_cls(sd, f"visual.transformer.resblocks.{cfg.v_layers - 1}", x, cfg.v_heads, prec)
    pooled = _ln(sd, "visual.ln_post", x[:, 0])                         # :876-877
    return pooled @ sd["visual.proj"]                                   # :977,1076 (bias-free)


def text_hidden(sd, cfg, text, precision="fp32"):
    x = sd["token_embedding.weight"][text] + sd["positional_embedding"][: text.shape[1]]   # :204-232
    if precision == "mx8":
        x = _r(x, True)                                                 # the block-scaled mode's bf16 residual stream
    for i in range(cfg.t_layers):
        x = _block(sd, f"transformer.resblocks.{i}", x, cfg.t_heads, causal=True, prec=None if precision == "fp32" else precision)
    return _ln(sd, "ln_final", x)                                       # :750


def encode_text(sd, cfg, text, mode="global", visual_emb=None, precision="fp32"):
    """int64 [B,77] -> (global [B,D], seq [B,77,D]) or seq (call sites run/test/test_fiq.py:102-103)."""
    if visual_emb is not None and (visual_emb.dim() != 3 or visual_emb.shape[1] != text.shape[0]):
        raise ValueError("visual_emb must be [patch_num, B, D]")
    seq = text_hidden(sd, cfg, text, precision) @ sd["text_projection"]            # :978,1027 (bias-free)
    if mode == "seq":
        return seq
    pooled = seq[torch.arange(text.shape[0]), text.argmax(dim=-1)]      # :755-758 (EOT = largest id)
    return pooled, seq
