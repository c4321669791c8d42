"""Seeded synthetic weights and inputs for the encode -> fuse -> rank path.

There are no checkpoints or datasets for the reference, so every
parity test, golden fixture and benchmark runs on weights produced here.  Tensors are
named with the reference's own state-dict keys:

* fusion (``ERN.state_dict()``): /root/reference/models/model.py:8-20 and
  /root/reference/models/fusion_model.py:8-216 (key list in SURVEY.md Appendix B);
* CLIP towers: the open_clip 2.20.0 key layout the reference loads with
  ``clip_model.load_state_dict(saved["CLIP"])`` (/root/reference/run/test/test_fiq.py:141-143).

Each tensor is drawn from its own numpy Generator seeded with (seed, crc32(key)), so
values do not depend on generation order, on torch, or on the device.  Statistics are
deliberately *not* the default initialisation: BatchNorm running stats, biases and the
CLS token are non-trivial so that a wrong axis or a dropped term shows up in parity tests.
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass
from typing import Dict

import numpy as np

PATCH_NUM = 13          # /root/reference/models/fusion_model.py:106 (num_region)
BERT_INTER = 3072       # HF BertConfig default intermediate_size (fusion_model.py:162-170)
BERT_LAYERS = 2         # fusion_model.py:14
BERT_HEADS = 8          # fusion_model.py:164
BERT_MAX_POS = 512      # fusion_model.py:165


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(key.encode())])


_SHAPES_ONLY = False      # fusion_state_shapes(): the builders below then produce zero-stride placeholders of the right shape


def _normal(seed, key, shape, std=1.0, mean=0.0):
    if _SHAPES_ONLY:
        return np.broadcast_to(np.float32(0), shape)
    return (_rng(seed, key).standard_normal(shape, dtype=np.float32) * np.float32(std)
            + np.float32(mean)).astype(np.float32)


def _uniform(seed, key, shape, lo, hi):
    if _SHAPES_ONLY:
        return np.broadcast_to(np.float32(0), shape)
    return _rng(seed, key).uniform(lo, hi, size=shape).astype(np.float32)


def _linear(sd, seed, prefix, out_f, in_f, bias_std=0.1, gain=1.0):
    sd[prefix + ".weight"] = _normal(seed, prefix + ".weight", (out_f, in_f), gain / np.sqrt(in_f))
    sd[prefix + ".bias"] = _normal(seed, prefix + ".bias", (out_f,), bias_std)


def _layernorm(sd, seed, prefix, n):
    sd[prefix + ".weight"] = _normal(seed, prefix + ".weight", (n,), 0.1, 1.0)
    sd[prefix + ".bias"] = _normal(seed, prefix + ".bias", (n,), 0.1)


def _batchnorm(sd, seed, prefix, n):
    sd[prefix + ".weight"] = _normal(seed, prefix + ".weight", (n,), 0.1, 1.0)
    sd[prefix + ".bias"] = _normal(seed, prefix + ".bias", (n,), 0.1)
    sd[prefix + ".running_mean"] = _normal(seed, prefix + ".running_mean", (n,), 0.3)
    sd[prefix + ".running_var"] = _uniform(seed, prefix + ".running_var", (n,), 0.5, 1.5)
    sd[prefix + ".num_batches_tracked"] = np.array(7, dtype=np.int64)


def _visual_sr(sd, seed, prefix, d):
    _linear(sd, seed, prefix + ".embedding_local.0", d, d)
    _batchnorm(sd, seed, prefix + ".embedding_local.1", PATCH_NUM)
    _linear(sd, seed, prefix + ".embedding_global.0", d, d)
    _batchnorm(sd, seed, prefix + ".embedding_global.1", d)
    _linear(sd, seed, prefix + ".embedding_common", 1, d, gain=4.0)


def _combiner(sd, seed, prefix, d):
    _linear(sd, seed, prefix + ".dynamic_scalar.0", 8 * d, 8 * d)
    _linear(sd, seed, prefix + ".dynamic_scalar.3", 1, 8 * d, gain=2.0)
    _linear(sd, seed, prefix + ".text_projection_layer.0", 4 * d, d)
    _linear(sd, seed, prefix + ".image_projection_layer.0", 4 * d, d)


def fusion_state_dict(feature_dim: int, seed: int = 0, with_cls_token: bool = True) -> Dict[str, np.ndarray]:
    """Synthetic ``ERN.state_dict()`` (fusion part only) for ``feature_dim`` in {.., 512, 640}."""
    d = int(feature_dim)
    if d % BERT_HEADS:
        raise ValueError("feature_dim must be divisible by 8 heads")
    sd: Dict[str, np.ndarray] = {}
    tl = "DVR.transformer_layer"
    bm = tl + ".bert_encoder.bert_model"
    if with_cls_token:
        sd[tl + ".cls_token"] = _normal(seed, tl + ".cls_token", (1, 1, d), 0.5)
    sd[bm + ".embeddings.position_embeddings.weight"] = _normal(
        seed, bm + ".embeddings.position_embeddings.weight", (BERT_MAX_POS, d), 0.2)
    sd[bm + ".embeddings.token_type_embeddings.weight"] = _normal(
        seed, bm + ".embeddings.token_type_embeddings.weight", (2, d), 0.2)
    _layernorm(sd, seed, bm + ".embeddings.LayerNorm", d)
    for i in range(BERT_LAYERS):
        lp = f"{bm}.encoder.layer.{i}"
        for n in ("query", "key", "value"):
            _linear(sd, seed, f"{lp}.attention.self.{n}", d, d, gain=1.5)
        _linear(sd, seed, f"{lp}.attention.output.dense", d, d)
        _layernorm(sd, seed, f"{lp}.attention.output.LayerNorm", d)
        _linear(sd, seed, f"{lp}.intermediate.dense", BERT_INTER, d)
        _linear(sd, seed, f"{lp}.output.dense", d, BERT_INTER)
        _layernorm(sd, seed, f"{lp}.output.LayerNorm", d)
    _linear(sd, seed, bm + ".pooler.dense", d, d)
    _visual_sr(sd, seed, "DVR.SR_module", d)
    p = "DVR.MR_component"
    sd[p + ".in_proj_weight"] = _normal(seed, p + ".in_proj_weight", (3 * d, d), 3.0 / np.sqrt(d))
    sd[p + ".in_proj_bias"] = _normal(seed, p + ".in_proj_bias", (3 * d,), 0.1)
    _linear(sd, seed, p + ".out_proj", d, d, gain=4.0)
    for c in ("combiner_global", "combiner_local", "combiner"):
        _combiner(sd, seed, "DVR." + c, d)
    _visual_sr(sd, seed, "SR_module", d)
    _combiner(sd, seed, "Combiner_module", d)
    return sd


def fusion_state_shapes(feature_dim: int) -> Dict[str, tuple]:
    """Key -> shape of ``ERN.state_dict()``'s fusion part, without generating a single value (what ``load_state_dict`` checks
    a checkpoint's key set against)."""
    global _SHAPES_ONLY
    _SHAPES_ONLY = True
    try:
        return {k: tuple(v.shape) for k, v in fusion_state_dict(feature_dim, 0).items()}
    finally:
        _SHAPES_ONLY = False


def clip4cir_state_dict(clip_feature_dim: int, projection_dim: int, hidden_dim: int, seed: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic state dict of models/others/Combiner_Model.py:Combiner (its Linear inputs are 2 * clip_feature_dim wide)."""
    sd: Dict[str, np.ndarray] = {}
    w = 2 * clip_feature_dim
    _linear(sd, seed, "text_projection_layer", projection_dim, w)
    _linear(sd, seed, "image_projection_layer", projection_dim, w)
    _linear(sd, seed, "combiner_layer", hidden_dim, 2 * projection_dim)
    _linear(sd, seed, "output_layer", w, hidden_dim)
    _linear(sd, seed, "dynamic_scalar.0", hidden_dim, 2 * projection_dim)
    _linear(sd, seed, "dynamic_scalar.3", 1, hidden_dim, gain=2.0)
    return sd


@dataclass(frozen=True)
class ClipConfig:
    """Shape of a CLIP model in open_clip's vocabulary."""
    name: str
    embed_dim: int
    image_size: int
    patch_size: int
    v_width: int
    v_layers: int
    v_heads: int
    v_mlp: int
    context_length: int
    vocab_size: int
    t_width: int
    t_heads: int
    t_layers: int
    t_mlp: int
    # image tower architecture: "vit" (above) or "resnet" = open_clip ModifiedResNet (RN50x4: layers (4,6,10,6), width 80)
    v_arch: str = "vit"
    r_layers: tuple = (0, 0, 0, 0)
    r_width: int = 0
    r_heads: int = 0

    @property
    def grid(self) -> int:
        return self.image_size // self.patch_size

    @property
    def v_tokens(self) -> int:
        return self.grid * self.grid + 1


CLIP_CONFIGS = {
    # open_clip "ViT-B-16" (run/test/test_fiq.py:134 default --clip-model-name)
    "ViT-B-16": ClipConfig("ViT-B-16", 512, 224, 16, 768, 12, 12, 3072, 77, 49408, 512, 8, 12, 2048),
    # open_clip "RN50x4" (run/test/test_cirr.py:149 default): ModifiedResNet (4,6,10,6) width 80 @288 px, attention pool 40 heads
    "RN50x4": ClipConfig("RN50x4", 640, 288, 32, 0, 1, 0, 0, 77, 49408, 640, 10, 12, 2560, "resnet", (4, 6, 10, 6), 80, 40),
    "RN50x4-text": ClipConfig("RN50x4-text", 640, 288, 16, 768, 0, 12, 3072, 77, 49408, 640, 10, 12, 2560),
    "tiny-resnet": ClipConfig("tiny-resnet", 128, 64, 32, 0, 1, 0, 0, 77, 1000, 128, 4, 2, 512, "resnet", (2, 1, 1, 1), 32, 16),
    # small shapes for tests / fixtures (same arithmetic, seconds on CPU)
    "tiny": ClipConfig("tiny", 128, 64, 16, 128, 2, 4, 512, 77, 1000, 128, 4, 2, 512),
    "tiny-hd64": ClipConfig("tiny-hd64", 64, 48, 16, 192, 2, 3, 384, 77, 600, 128, 2, 2, 256),
    "tiny-w256": ClipConfig("tiny-w256", 64, 48, 16, 256, 3, 4, 512, 77, 600, 128, 2, 2, 256),     # widths % 128 == 0 (block-scaled fp8 mode)
    "tiny-hd48": ClipConfig("tiny-hd48", 64, 48, 16, 384, 2, 8, 768, 77, 600, 128, 2, 2, 256),     # head_dim 48: not a multiple of an MX block
}


def _conv_bn(sd, seed, conv, bn, cout, cin, k, gamma=1.0):
    sd[conv + ".weight"] = _normal(seed, conv + ".weight", (cout, cin, k, k), 1.4 / np.sqrt(cin * k * k))
    _batchnorm(sd, seed, bn, cout)
    sd[bn + ".weight"] = (sd[bn + ".weight"] * np.float32(gamma)).astype(np.float32)


def _resnet_state(sd, cfg: ClipConfig, seed: int) -> None:
    """open_clip ModifiedResNet key layout (visual.conv1..3 / bn1..3, visual.layerL.i.{conv1,bn1,...,downsample.0/1}, visual.attnpool.*)."""
    w = cfg.r_width
    _conv_bn(sd, seed, "visual.conv1", "visual.bn1", w // 2, 3, 3)
    _conv_bn(sd, seed, "visual.conv2", "visual.bn2", w // 2, w // 2, 3)
    _conv_bn(sd, seed, "visual.conv3", "visual.bn3", w, w // 2, 3)
    inplanes = w
    for li, nblocks in enumerate(cfg.r_layers):
        planes = w * (2 ** li)
        for bi in range(nblocks):
            p = f"visual.layer{li + 1}.{bi}"
            stride = 2 if (bi == 0 and li > 0) else 1
            _conv_bn(sd, seed, p + ".conv1", p + ".bn1", planes, inplanes, 1)
            _conv_bn(sd, seed, p + ".conv2", p + ".bn2", planes, planes, 3)
            _conv_bn(sd, seed, p + ".conv3", p + ".bn3", planes * 4, planes, 1, gamma=0.3)   # small residual branch: activations stay O(1)
            if stride > 1 or inplanes != planes * 4:
                _conv_bn(sd, seed, p + ".downsample.0", p + ".downsample.1", planes * 4, inplanes, 1)
            inplanes = planes * 4
    embed = w * 32
    tokens = (cfg.image_size // 32) ** 2 + 1
    sd["visual.attnpool.positional_embedding"] = _normal(seed, "visual.attnpool.positional_embedding", (tokens, embed), 0.5)
    for n in ("q_proj", "k_proj", "v_proj"):
        _linear(sd, seed, "visual.attnpool." + n, embed, embed, gain=1.5)
    _linear(sd, seed, "visual.attnpool.c_proj", cfg.embed_dim, embed)


def clip_state_dict(cfg: ClipConfig, seed: int = 0) -> Dict[str, np.ndarray]:
    """Synthetic open_clip-layout state dict (the dict stored under ``["CLIP"]``, test_fiq.py:143)."""
    sd: Dict[str, np.ndarray] = {}

    def block(prefix, width, mlp):
        _layernorm(sd, seed, prefix + ".ln_1", width)
        sd[prefix + ".attn.in_proj_weight"] = _normal(seed, prefix + ".attn.in_proj_weight",
                                                      (3 * width, width), 1.2 / np.sqrt(width))
        sd[prefix + ".attn.in_proj_bias"] = _normal(seed, prefix + ".attn.in_proj_bias", (3 * width,), 0.1)
        _linear(sd, seed, prefix + ".attn.out_proj", width, width, gain=0.7)
        _layernorm(sd, seed, prefix + ".ln_2", width)
        _linear(sd, seed, prefix + ".mlp.c_fc", mlp, width)
        _linear(sd, seed, prefix + ".mlp.c_proj", width, mlp, gain=0.7)

    if cfg.v_arch == "resnet":
        _resnet_state(sd, cfg, seed)
    elif cfg.v_layers > 0:
        vw = cfg.v_width
        sd["visual.class_embedding"] = _normal(seed, "visual.class_embedding", (vw,), 0.5)
        sd["visual.positional_embedding"] = _normal(seed, "visual.positional_embedding", (cfg.v_tokens, vw), 0.3)
        sd["visual.proj"] = _normal(seed, "visual.proj", (vw, cfg.embed_dim), 1.0 / np.sqrt(vw))
        k = 3 * cfg.patch_size * cfg.patch_size
        sd["visual.conv1.weight"] = _normal(seed, "visual.conv1.weight",
                                            (vw, 3, cfg.patch_size, cfg.patch_size), 1.0 / np.sqrt(k))
        _layernorm(sd, seed, "visual.ln_pre", vw)
        for i in range(cfg.v_layers):
            block(f"visual.transformer.resblocks.{i}", vw, cfg.v_mlp)
        _layernorm(sd, seed, "visual.ln_post", vw)
    tw = cfg.t_width
    sd["token_embedding.weight"] = _normal(seed, "token_embedding.weight", (cfg.vocab_size, tw), 0.5)
    sd["positional_embedding"] = _normal(seed, "positional_embedding", (cfg.context_length, tw), 0.3)
    for i in range(cfg.t_layers):
        block(f"transformer.resblocks.{i}", tw, cfg.t_mlp)
    _layernorm(sd, seed, "ln_final", tw)
    sd["text_projection"] = _normal(seed, "text_projection", (tw, cfg.embed_dim), 1.0 / np.sqrt(tw))
    sd["logit_scale"] = np.array(np.log(1 / 0.07), dtype=np.float32)
    return sd


# ----------------------------------------------------------------------------------------
# synthetic inputs (SURVEY.md section 8d "Synthetic inputs")
# ----------------------------------------------------------------------------------------

def images(n: int, cfg: ClipConfig, seed: int = 42) -> np.ndarray:
    """N(0,1) f32 [n,3,H,W]: the post-``Normalize`` domain of dataloader/dataset.py:73-87."""
    return _normal(seed, f"images/{cfg.image_size}", (n, 3, cfg.image_size, cfg.image_size))


def captions(n: int, cfg: ClipConfig, seed: int = 42, full_length: bool = True) -> np.ndarray:
    """int64 [n,77] token ids: SOT, random ids, EOT (= the largest id, so argmax pooling is defined), 0-pad."""
    rng = _rng(seed, "captions")
    sot, eot = cfg.vocab_size - 2, cfg.vocab_size - 1
    t = np.zeros((n, cfg.context_length), dtype=np.int64)
    for i in range(n):
        length = cfg.context_length - 2 if full_length else int(rng.integers(1, cfg.context_length - 1))
        t[i, 0] = sot
        t[i, 1:1 + length] = rng.integers(1, sot, size=length)
        t[i, 1 + length] = eot
    return t


def local_feats(n: int, d: int, seed: int = 42, tag: str = "local") -> np.ndarray:
    return _normal(seed, f"{tag}/{d}", (n, PATCH_NUM, d))


def global_feats(n: int, d: int, seed: int = 42, tag: str = "global") -> np.ndarray:
    return _normal(seed, f"{tag}/{d}", (n, d))


def unit_rows(n: int, d: int, seed: int = 42, tag: str = "unit") -> np.ndarray:
    x = _normal(seed, f"{tag}/{d}", (n, d))
    return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)
