"""Thin torch-facing wrapper over the libfern C ABI.

torch is used for what it is good at here -- device memory, streams, process groups -- and nothing
else: every op below hands raw ``data_ptr()``s of caller/torch-allocated buffers to a HIP kernel
sequence in libfern.so on torch's current stream.  There is no eager/CPU fallback: constructing an
engine without a ROCm device or without the library raises.
"""
from __future__ import annotations

import ctypes as C
import sys
from typing import Dict, Mapping, Optional

import numpy as np
import torch

from . import _lib
from .synth import ClipConfig

COMBINER_TARGET, COMBINER_DVR_GLOBAL, COMBINER_DVR_LOCAL, COMBINER_DVR_FINAL = 0, 1, 2, 3
SR_TARGET, SR_DVR = 0, 1
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_RELU, EPI_BIAS_RESIDUAL = 0, 1, 2, 3
PART_DVR, PART_TARGET_SR, PART_TARGET_COMBINER, PART_ALL = 1, 2, 4, 7
PREC_FP32, PREC_BF16, PREC_FP8, PREC_MX8, PREC_F32X3, PREC_MX8_MLP, PREC_MX8_IMG = 0, 1, 2, 3, 4, 5, 6
_PREC_NAMES = {"fp32": PREC_FP32, "bf16": PREC_BF16, "fp8": PREC_FP8, "mx8": PREC_MX8, "f32x3": PREC_F32X3, "mx8mlp": PREC_MX8_MLP, "mx8img": PREC_MX8_IMG}
PATCH_NUM = 13


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class PreparedGallery:
    """An fp32 gallery [N, D] together with its certified bf16 pre-filter copy (include/fern.h: fern_gallery_prepare): what
    `FernEngine.sim_topk` takes to run the ranking stage as one HBM-bound pass over the bf16 rows + exact fp32 rescoring of the few
    rows that can still be in the top-K.  Results are those of the fp32 gallery, bit for bit.  Build it once per gallery
    (`FernEngine.prepare_gallery`), like the reference builds its index once per evaluation (run/test/test_fiq.py:45-46); rebuild
    it when the gallery's contents change."""

    __slots__ = ("f32", "bf16", "meta")

    def __init__(self, f32: torch.Tensor, bf16: torch.Tensor, meta: torch.Tensor):
        self.f32, self.bf16, self.meta = f32, bf16, meta

    @property
    def shape(self):
        return self.f32.shape

    @property
    def dtype(self):
        return self.f32.dtype

    def float(self):
        return self.f32

    def data_ptr(self) -> int:
        return self.f32.data_ptr()

    def is_contiguous(self) -> bool:
        return True


class FernEngine:
    """One native context on one GPU.  Not thread-safe (one per device per process)."""

    def __init__(self, device="cuda:0"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("FernEngine needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"FernEngine runs on a GPU only, got device {device!r}")
        index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", index)
        h = C.c_void_p()
        _lib.check(self.lib.fern_ctx_create(index, C.byref(h)), "fern_ctx_create")
        self._h = h
        self.feature_dim: Optional[int] = None
        self.clip_cfg: Optional[ClipConfig] = None

    def fork(self) -> "FernEngine":
        """A context that shares this engine's finalised weights (no copy) but owns its workspace: use one per extra
        HIP stream.  Keep the parent alive (the fork holds a reference) and fork again after re-loading weights."""
        child = object.__new__(FernEngine)
        child.lib, child.device = self.lib, self.device
        child.feature_dim, child.clip_cfg = self.feature_dim, self.clip_cfg
        child._parent = self
        h = C.c_void_p()
        _lib.check(self.lib.fern_ctx_fork(self._h, C.byref(h)), "fern_ctx_fork")
        child._h = h
        return child

    def set_precision(self, precision) -> None:
        """Operand precision of the CLIP towers' token-level GEMMs: "fp32" (parity mode, default), "bf16", "fp8" (per-row
        scales), "mx8" (block-scaled fp8 on the scaled MFMA; "mx8mlp": only the image tower's MLP pair, "mx8img": the image tower's
        four GEMMs over the fp32 residual stream -- both with a bf16 text tower; bench.py's c5 default is "mx8img") -- or "f32x3": fp32 data, every large plain GEMM computed from three
        bf16 planes per operand (fp32-accurate, ~1.4x faster, not the bit-exact fma chain) -- include/fern.h:fern_precision."""
        prec = _PREC_NAMES[precision] if isinstance(precision, str) else int(precision)
        _lib.check(self.lib.fern_set_precision(self._h, prec), "fern_set_precision")

    @property
    def precision(self) -> str:
        return {v: k for k, v in _PREC_NAMES.items()}[self.lib.fern_get_precision(self._h)]

    def close(self):
        pipe = getattr(self, "_harness_pipe", None)      # forks of this context (run/_common.py keeps a query pipeline here) go first
        if pipe is not None:
            self._harness_pipe = None
            try:
                pipe.close()
            except Exception:
                pass
        if getattr(self, "_h", None):
            self.lib.fern_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        # At interpreter shutdown the HIP runtime (and a profiler's tool library) may already be finalised: calling
        # hipDeviceSynchronize / hipFree then can block forever.  The OS reclaims the context with the process.
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    # ---- tensors ------------------------------------------------------------------------------
    def _f32(self, t, shape=None) -> torch.Tensor:
        """The tensor as a contiguous fp32 tensor on this engine's device.  A tensor that already is one is passed through
        untouched (the hot path); anything else (host tensors as the reference's loaders yield them, other dtypes, numpy)
        is converted -- an H2D copy per call, which a caller on the hot path avoids by keeping its tensors on the device."""
        if not isinstance(t, torch.Tensor):
            t = torch.as_tensor(np.asarray(t))
        if not (t.device == self.device and t.dtype == torch.float32 and t.is_contiguous()):
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"expected shape {tuple(shape)}, got {tuple(t.shape)}")
        return t

    def _empty(self, *shape, dtype=torch.float32) -> torch.Tensor:
        return torch.empty(*shape, dtype=dtype, device=self.device)

    # ---- weights ------------------------------------------------------------------------------
    def load_tensors(self, state_dict: Mapping[str, object], prefix: str = "") -> None:
        """Push a state dict (reference key names; torch tensors or numpy arrays) to the native side."""
        for key, val in state_dict.items():
            if isinstance(val, torch.Tensor):
                val = val.detach().cpu().numpy()
            arr = np.asarray(val)
            if arr.dtype.kind == "f":
                arr, dt = np.ascontiguousarray(arr, dtype=np.float32), 0
            else:
                arr, dt = np.ascontiguousarray(arr, dtype=np.int64), 1
            shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
            _lib.check(self.lib.fern_load_tensor(self._h, (prefix + key).encode(), arr.ctypes.data_as(C.c_void_p), dt,
                                                 arr.ndim, shape), f"fern_load_tensor({key})")

    def finalize_fusion(self, feature_dim: int, parts: int = PART_ALL) -> None:
        _lib.check(self.lib.fern_finalize_fusion(self._h, int(feature_dim), int(parts)), "fern_finalize_fusion")
        self.feature_dim = int(feature_dim)

    def finalize_clip(self, cfg: ClipConfig) -> None:
        cc = _lib.ClipConfigC(cfg.embed_dim, cfg.image_size, cfg.patch_size, cfg.v_width, cfg.v_layers, cfg.v_heads,
                              cfg.v_mlp, cfg.context_length, cfg.vocab_size, cfg.t_width, cfg.t_heads, cfg.t_layers, cfg.t_mlp,
                              1 if cfg.v_arch == "resnet" else 0, (C.c_int * 4)(*cfg.r_layers), cfg.r_width, cfg.r_heads)
        _lib.check(self.lib.fern_finalize_clip(self._h, C.byref(cc)), "fern_finalize_clip")
        self.clip_cfg = cfg

    # ---- encoders -----------------------------------------------------------------------------
    def encode_image(self, images: torch.Tensor) -> torch.Tensor:
        cfg = self.clip_cfg
        if cfg is None:
            raise _lib.FernError("encode_image: CLIP weights not finalised")
        x = self._f32(images)
        if x.dim() != 4 or tuple(x.shape[1:]) != (3, cfg.image_size, cfg.image_size):
            raise ValueError(f"images must be [b,3,{cfg.image_size},{cfg.image_size}], got {tuple(x.shape)}")
        out = self._empty(x.shape[0], cfg.embed_dim)
        _lib.check(self.lib.fern_vit_encode_image(self._h, _ptr(x), _ptr(out), x.shape[0], _stream()), "fern_vit_encode_image")
        return out

    def encode_text(self, tokens: torch.Tensor, want_global=True, want_seq=True, visual_emb: Optional[torch.Tensor] = None):
        """tokens int64 [B,ctx] -> (global [B,D] | None, seq [B,ctx,D] | None).  `visual_emb` [13,B,D] is the reference's
        argument of that name (models/clip_model.py:23-31): shape-checked by the library, values unused.  Token ids outside
        the vocabulary raise IndexError like nn.Embedding when the tokens are on the host (as the reference's tokenizer leaves
        them, test_fiq.py:98); for device tokens the library flags them at the next sync / encode_text (no sync here)."""
        cfg = self.clip_cfg
        if cfg is None:
            raise _lib.FernError("encode_text: CLIP weights not finalised")
        if tokens.dim() != 2 or tokens.shape[1] != cfg.context_length:
            raise ValueError(f"text must be int64 [B,{cfg.context_length}], got {tuple(tokens.shape)}")
        if not tokens.is_cuda and tokens.numel() and (int(tokens.min()) < 0 or int(tokens.max()) >= cfg.vocab_size):
            raise IndexError(f"token id out of range [0, {cfg.vocab_size}): min {int(tokens.min())}, max {int(tokens.max())}")
        t = tokens.to(device=self.device, dtype=torch.int64).contiguous()
        b = t.shape[0]
        ve, ve_shape = None, None
        if visual_emb is not None:
            if visual_emb.dim() != 3:
                raise ValueError(f"visual_emb must be [{PATCH_NUM}, B, {cfg.embed_dim}], got {tuple(visual_emb.shape)}")
            ve = visual_emb if visual_emb.is_cuda else visual_emb.to(self.device)      # only its address and shape cross the ABI
            ve_shape = (C.c_int64 * 3)(*visual_emb.shape)
        g = self._empty(b, cfg.embed_dim) if want_global else None
        s = self._empty(b, cfg.context_length, cfg.embed_dim) if want_seq else None
        _lib.check(self.lib.fern_text_encode(self._h, _ptr(t), _ptr(ve), ve_shape, _ptr(g), _ptr(s), b, _stream()), "fern_text_encode")
        return g, s

    def encode_pair(self, images: torch.Tensor, tokens: torch.Tensor, want_seq: bool = True):
        """`encode_image(images)` and `encode_text(tokens)` of the SAME query batch in one pass (include/fern.h: fern_encode_pair): the towers
        are walked layer by layer and the text layer's GEMMs ride in the image layer's launches.  Returns (image [B,D], text global [B,D],
        text seq [B,ctx,D] | None) -- bit-identical to the two calls; in a mode or with a tower the library does not pair, it makes them."""
        cfg = self.clip_cfg
        if cfg is None:
            raise _lib.FernError("encode_pair: CLIP weights not finalised")
        if images.dim() != 4 or tuple(images.shape[1:]) != (3, cfg.image_size, cfg.image_size):
            raise ValueError(f"images must be [b,3,{cfg.image_size},{cfg.image_size}], got {tuple(images.shape)}")
        if tokens.dim() != 2 or tokens.shape[1] != cfg.context_length or tokens.shape[0] != images.shape[0]:
            raise ValueError(f"text must be int64 [{images.shape[0]},{cfg.context_length}], got {tuple(tokens.shape)}")
        if not tokens.is_cuda and tokens.numel() and (int(tokens.min()) < 0 or int(tokens.max()) >= cfg.vocab_size):
            raise IndexError(f"token id out of range [0, {cfg.vocab_size}): min {int(tokens.min())}, max {int(tokens.max())}")
        x = self._f32(images)
        t = tokens.to(device=self.device, dtype=torch.int64).contiguous()
        b = x.shape[0]
        out = self._empty(b, cfg.embed_dim)
        g = self._empty(b, cfg.embed_dim)
        s = self._empty(b, cfg.context_length, cfg.embed_dim) if want_seq else None
        _lib.check(self.lib.fern_encode_pair(self._h, _ptr(x), _ptr(t), _ptr(out), _ptr(g), _ptr(s), b, _stream()), "fern_encode_pair")
        return out, g, s

    # ---- fusion -------------------------------------------------------------------------------
    def _d(self) -> int:
        if self.feature_dim is None:
            raise _lib.FernError("fusion weights not finalised")
        return self.feature_dim

    def dvr_fuse(self, ref_global, ref_local, text_global, text_seq) -> torch.Tensor:
        d = self._d()
        rl = self._f32(ref_local)
        b = rl.shape[0]
        ts = self._f32(text_seq)
        if rl.dim() != 3 or rl.shape[1] != PATCH_NUM or rl.shape[2] != d:
            raise ValueError(f"ref_local_feats must be [B,{PATCH_NUM},{d}], got {tuple(rl.shape)}")
        if ts.dim() != 3 or ts.shape[0] != b or ts.shape[2] != d:
            raise ValueError(f"text_seq_feats must be [B,T,{d}], got {tuple(ts.shape)}")
        if ts.shape[1] < PATCH_NUM:      # fusion_model.py:47 keeps 13 text-query rows for BatchNorm1d(13)
            raise ValueError(f"text_seq_feats needs at least {PATCH_NUM} token rows, got {ts.shape[1]}")
        rg, tg = self._f32(ref_global, (b, d)), self._f32(text_global, (b, d))
        out = self._empty(b, d)
        _lib.check(self.lib.fern_dvr_fuse(self._h, _ptr(rg), _ptr(rl), _ptr(tg), _ptr(ts), _ptr(out), b, ts.shape[1],
                                          _stream()), "fern_dvr_fuse")
        return out

    def index_fuse(self, tar_feats, tar_local, normalize_input=False) -> torch.Tensor:
        d = self._d()
        tl = self._f32(tar_local)
        n = tl.shape[0]
        if tl.dim() != 3 or tl.shape[1] != PATCH_NUM or tl.shape[2] != d:
            raise ValueError(f"tar_local_feats must be [n,{PATCH_NUM},{d}], got {tuple(tl.shape)}")
        tf = self._f32(tar_feats, (n, d))
        out = self._empty(n, d)
        _lib.check(self.lib.fern_index_fuse(self._h, _ptr(tf), _ptr(tl), _ptr(out), n, int(bool(normalize_input)), _stream()),
                   "fern_index_fuse")
        return out

    def combiner(self, which: int, image, text) -> torch.Tensor:
        d = self._d()
        im = self._f32(image)
        if im.dim() != 2 or im.shape[1] != d:
            raise ValueError(f"image_features must be [n,{d}], got {tuple(im.shape)}")
        tx = self._f32(text, tuple(im.shape))
        out = self._empty(*im.shape)
        _lib.check(self.lib.fern_combiner(self._h, which, _ptr(im), _ptr(tx), _ptr(out), im.shape[0], _stream()), "fern_combiner")
        return out

    def visual_sr(self, which: int, local) -> torch.Tensor:
        d = self._d()
        x = self._f32(local)
        if x.dim() != 3 or x.shape[1] != PATCH_NUM or x.shape[2] != d:
            raise ValueError(f"local_feature must be [n,{PATCH_NUM},{d}], got {tuple(x.shape)}")
        out = self._empty(x.shape[0], d)
        _lib.check(self.lib.fern_visual_sr(self._h, which, _ptr(x), _ptr(out), x.shape[0], _stream()), "fern_visual_sr")
        return out

    def finalize_clip4cir(self) -> None:
        _lib.check(self.lib.fern_finalize_clip4cir(self._h), "fern_finalize_clip4cir")

    def combiner_clip4cir(self, image, text) -> torch.Tensor:
        im = self._f32(image)
        tx = self._f32(text, tuple(im.shape))
        out = torch.empty_like(im)
        _lib.check(self.lib.fern_combiner_clip4cir(self._h, _ptr(im), _ptr(tx), _ptr(out), im.shape[0], _stream()), "fern_combiner_clip4cir")
        return out

    def element_wise_sum(self, image, text) -> torch.Tensor:
        im = self._f32(image)
        tx = self._f32(text, tuple(im.shape))
        out = torch.empty_like(im)
        _lib.check(self.lib.fern_element_wise_sum(self._h, _ptr(im), _ptr(tx), _ptr(out), im.shape[0], im.shape[1], _stream()),
                   "fern_element_wise_sum")
        return out

    def l2_normalize(self, x) -> torch.Tensor:
        x = self._f32(x)
        out = torch.empty_like(x)
        _lib.check(self.lib.fern_l2_normalize(self._h, _ptr(x), _ptr(out), x.shape[0], x.shape[1], _stream()), "fern_l2_normalize")
        return out

    # ---- rank ---------------------------------------------------------------------------------
    def prepare_gallery(self, gallery, out: Optional[PreparedGallery] = None) -> PreparedGallery:
        """fp32 [N,D] -> PreparedGallery (the gallery itself, its bf16 copy, the three norms that certify the copy as a pre-filter).
        `out`: a PreparedGallery of the same shape to refill in place (a serving process's store)."""
        g = self._f32(gallery)
        if g.dim() != 2 or g.shape[1] % 4:
            raise ValueError(f"gallery must be [N,D] with D % 4 == 0, got {tuple(g.shape)}")
        if out is not None and tuple(out.bf16.shape) == tuple(g.shape):
            b16, meta = out.bf16, out.meta
        else:
            b16 = torch.empty(g.shape, dtype=torch.bfloat16, device=self.device)
            meta = torch.zeros(4, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.fern_gallery_prepare(self._h, _ptr(g), g.shape[0], g.shape[1], _ptr(b16), _ptr(meta), _stream()), "fern_gallery_prepare")
        return PreparedGallery(g, b16, meta)

    def sim_topk(self, q, gallery, k: int, idx_offset: int = 0, exclude_idx=None):
        """Exact cosine top-K of q [B,D] against an fp32 gallery [N,D] (run/test/test_fiq.py:49-50).  `gallery` is a tensor -- the
        fp32-MFMA sweep -- or a `PreparedGallery` -- bf16 pre-filter + exact rescoring, same scores and ordering bit for bit."""
        if isinstance(gallery, PreparedGallery):
            return self._sim_topk_prefiltered(q, gallery, k, idx_offset, exclude_idx)
        q, g = self._f32(q), self._f32(gallery)
        if q.dim() != 2 or g.dim() != 2 or q.shape[1] != g.shape[1]:
            raise ValueError(f"q [B,D] and gallery [N,D] must share D, got {tuple(q.shape)} and {tuple(g.shape)}")
        b = q.shape[0]
        scores = self._empty(b, k)
        idx = self._empty(b, k, dtype=torch.int32)
        ex = None
        if exclude_idx is not None:
            ex = torch.as_tensor(exclude_idx).to(device=self.device, dtype=torch.int32).contiguous()
            if tuple(ex.shape) != (b,):
                raise ValueError("exclude_idx must be [B]")
        _lib.check(self.lib.fern_sim_topk(self._h, _ptr(q), _ptr(g), b, g.shape[0], q.shape[1], int(k), _ptr(scores), _ptr(idx),
                                          int(idx_offset), _ptr(ex), _stream()), "fern_sim_topk")
        return scores, idx

    def sweep_bf16_scores(self, q, pg: "PreparedGallery", tile_max: bool = True):
        """The pre-filter's approximate scores [B,N] (bf16 operands, fp32 accumulation) and, with `tile_max`, the largest score of
        every 32 consecutive gallery rows [B, ceil(N/32)] -- what the dense form of the ranking stage selects on
        (include/fern.h: fern_sweep_bf16_scores)."""
        q = self._f32(q)
        n, d = pg.shape
        b = q.shape[0]
        scores = self._empty(b, n)
        nt = (n + 31) // 32
        tmax = self._empty(b, nt) if tile_max else None
        _lib.check(self.lib.fern_sweep_bf16_scores(self._h, _ptr(q), _ptr(pg.bf16), b, n, d, _ptr(scores), n, _ptr(tmax), nt, _stream()),
                   "fern_sweep_bf16_scores")
        return (scores, tmax) if tile_max else scores

    def set_rank_strategy(self, strategy) -> None:
        """Form of the PreparedGallery ranking stage: "auto" (cost model), "plain" (fp32 sweep), "lists", "dense" -- identical results
        (include/fern.h: fern_rank_strategy); a tuning / test knob."""
        code = {"auto": 0, "plain": 1, "lists": 2, "dense": 3}[strategy] if isinstance(strategy, str) else int(strategy)
        _lib.check(self.lib.fern_rank_set_strategy(self._h, code), "fern_rank_set_strategy")

    def _sim_topk_prefiltered(self, q, pg: PreparedGallery, k: int, idx_offset: int = 0, exclude_idx=None):
        q, g = self._f32(q), pg.f32
        if q.dim() != 2 or q.shape[1] != g.shape[1]:
            raise ValueError(f"q [B,D] and gallery [N,D] must share D, got {tuple(q.shape)} and {tuple(g.shape)}")
        b = q.shape[0]
        scores = self._empty(b, k)
        idx = self._empty(b, k, dtype=torch.int32)
        ex = None
        if exclude_idx is not None:
            ex = torch.as_tensor(exclude_idx).to(device=self.device, dtype=torch.int32).contiguous()
            if tuple(ex.shape) != (b,):
                raise ValueError("exclude_idx must be [B]")
        _lib.check(self.lib.fern_sim_topk_prefiltered(self._h, _ptr(q), _ptr(g), _ptr(pg.bf16), _ptr(pg.meta), b, g.shape[0], q.shape[1], int(k),
                                                      _ptr(scores), _ptr(idx), int(idx_offset), _ptr(ex), _stream()), "fern_sim_topk_prefiltered")
        return scores, idx

    def gallery_to_bf16(self, gallery) -> torch.Tensor:
        """fp32 [N,D] -> bf16 [N,D] (round to nearest even) for `sim_topk_bf16`."""
        g = self._f32(gallery)
        out = torch.empty(g.shape, dtype=torch.bfloat16, device=self.device)
        _lib.check(self.lib.fern_gallery_to_bf16(self._h, _ptr(g), _ptr(out), g.shape[0], g.shape[1], _stream()), "fern_gallery_to_bf16")
        return out

    def sim_topk_bf16(self, q, gallery_bf16: torch.Tensor, k: int, idx_offset: int = 0, exclude_idx=None):
        q = self._f32(q)
        g = gallery_bf16
        if g.dtype != torch.bfloat16 or g.dim() != 2 or not g.is_cuda or not g.is_contiguous() or g.shape[1] != q.shape[1]:
            raise ValueError("gallery must be a contiguous bf16 [N,D] device tensor sharing D with q")
        b = q.shape[0]
        scores = self._empty(b, k)
        idx = self._empty(b, k, dtype=torch.int32)
        ex = None
        if exclude_idx is not None:
            ex = torch.as_tensor(exclude_idx).to(device=self.device, dtype=torch.int32).contiguous()
        _lib.check(self.lib.fern_sim_topk_bf16(self._h, _ptr(q), _ptr(g), b, g.shape[0], q.shape[1], int(k), _ptr(scores), _ptr(idx),
                                               int(idx_offset), _ptr(ex), _stream()), "fern_sim_topk_bf16")
        return scores, idx

    def gather_scores(self, q, gallery, idx):
        q, g = self._f32(q), self._f32(gallery.f32 if isinstance(gallery, PreparedGallery) else gallery)
        ix = torch.as_tensor(idx).to(device=self.device, dtype=torch.int32).contiguous()
        out = self._empty(*ix.shape)
        _lib.check(self.lib.fern_gather_scores(self._h, _ptr(q), _ptr(g), _ptr(ix), _ptr(out), ix.shape[0], ix.shape[1],
                                               q.shape[1], _stream()), "fern_gather_scores")
        return out

    def topk_merge(self, scores, idx):
        s = self._f32(scores)
        ix = torch.as_tensor(idx).to(device=self.device, dtype=torch.int32).contiguous()
        r, b, k = s.shape
        os_, oi = self._empty(b, k), self._empty(b, k, dtype=torch.int32)
        _lib.check(self.lib.fern_topk_merge(self._h, _ptr(s), _ptr(ix), _ptr(os_), _ptr(oi), r, b, k, _stream()), "fern_topk_merge")
        return os_, oi

    # ---- building blocks ----------------------------------------------------------------------
    def gemm(self, a, w, bias=None, residual=None, epilogue=EPI_BIAS) -> torch.Tensor:
        a, w = self._f32(a), self._f32(w)
        m, k = a.shape
        n = w.shape[0]
        bias = None if bias is None else self._f32(bias, (n,))
        residual = None if residual is None else self._f32(residual, (m, n))
        out = self._empty(m, n)
        _lib.check(self.lib.fern_gemm(self._h, _ptr(a), k, _ptr(w), k, _ptr(bias), _ptr(residual), _ptr(out), n, m, n, k,
                                      int(epilogue), _stream()), "fern_gemm")
        return out

    def batch_classification_loss(self, predicted, target) -> torch.Tensor:
        """losses/loss.py:10-14: cross_entropy(100 * predicted @ target.T, arange(B)); returns a 0-dim device tensor."""
        p = self._f32(predicted)
        t = self._f32(target, tuple(p.shape))
        out = self._empty(1)
        _lib.check(self.lib.fern_batch_classification_loss(self._h, _ptr(p), _ptr(t), p.shape[0], p.shape[1], _ptr(out), _stream()),
                   "fern_batch_classification_loss")
        return out[0]

    def to_bf16(self, x) -> torch.Tensor:
        """fp32 [R,C] -> bf16 [R,C], round to nearest even (the conversion every bf16 operand of libfern goes through)."""
        return self.gallery_to_bf16(x)

    def gemm_bf16(self, a, w, bias=None, residual=None, epilogue=EPI_BIAS, out_bf16=False) -> torch.Tensor:
        """bf16 x bf16 -> fp32-accumulate GEMM; a [M,K] / w [N,K] are bf16 tensors (fp32 inputs are rounded first)."""
        a = a if a.dtype == torch.bfloat16 else self.to_bf16(a)
        w = w if w.dtype == torch.bfloat16 else self.to_bf16(w)
        a, w = a.to(self.device).contiguous(), w.to(self.device).contiguous()
        m, k = a.shape
        n = w.shape[0]
        bias = None if bias is None else self._f32(bias, (n,))
        residual = None if residual is None else self._f32(residual, (m, n))
        out = torch.empty(m, n, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=self.device)
        _lib.check(self.lib.fern_gemm_bf16(self._h, _ptr(a), k, _ptr(w), k, _ptr(bias), _ptr(residual), _ptr(out), n, m, n, k,
                                           int(epilogue), int(bool(out_bf16)), _stream()), "fern_gemm_bf16")
        return out

    def quantize_rows_fp8(self, x):
        """[R,C] fp32 or bf16 -> (fp8 e4m3fn bytes [R,C] as uint8, per-row scales [R]); scale = max|row| / 448."""
        x = x.to(self.device).contiguous() if x.dtype == torch.bfloat16 else self._f32(x)
        rows, d = x.shape
        y = torch.empty(rows, d, dtype=torch.uint8, device=self.device)
        sc = self._empty(rows)
        _lib.check(self.lib.fern_quantize_rows_fp8(self._h, _ptr(x), int(x.dtype == torch.bfloat16), d, _ptr(y), d, _ptr(sc), rows, d,
                                                   _stream()), "fern_quantize_rows_fp8")
        return y, sc

    def gemm_fp8(self, a8, sa, w8, sw, bias=None, residual=None, epilogue=EPI_BIAS, out_bf16=False) -> torch.Tensor:
        m, k = a8.shape
        n = w8.shape[0]
        bias = None if bias is None else self._f32(bias, (n,))
        residual = None if residual is None else self._f32(residual, (m, n))
        out = torch.empty(m, n, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=self.device)
        _lib.check(self.lib.fern_gemm_fp8(self._h, _ptr(a8), k, _ptr(sa), _ptr(w8), k, _ptr(sw), _ptr(bias), _ptr(residual), _ptr(out), n,
                                          m, n, k, int(epilogue), int(bool(out_bf16)), _stream()), "fern_gemm_fp8")
        return out

    def quantize_mx8(self, x):
        """[R,C] fp32 or bf16 (C % 128 == 0) -> (e4m3fn bytes [R,C] as uint8, E8M0 block scales as uint8 [C/128, R, 4]):
        scale byte of (row r, 32-k block b) = scales[b // 4, r, b % 4] (include/fern.h: fern_quantize_mx8)."""
        x = x.to(self.device).contiguous() if x.dtype == torch.bfloat16 else self._f32(x)
        rows, d = x.shape
        y = torch.empty(rows, d, dtype=torch.uint8, device=self.device)
        sc = torch.empty(d // 128, rows, 4, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.fern_quantize_mx8(self._h, _ptr(x), int(x.dtype == torch.bfloat16), d, _ptr(y), d, _ptr(sc), rows, rows, d,
                                              _stream()), "fern_quantize_mx8")
        return y, sc

    def gemm_mx8(self, a8, sa, w8, sw, bias=None, residual=None, epilogue=EPI_BIAS, out_bf16=False) -> torch.Tensor:
        """Block-scaled fp8 GEMM on v_mfma_scale_f32_32x32x64_f8f6f4; operands and scales as quantize_mx8 returns them."""
        m, k = a8.shape
        n = w8.shape[0]
        bias = None if bias is None else self._f32(bias, (n,))
        if residual is not None and out_bf16:      # the bf16 residual-stream form (include/fern.h: fern_gemm_mx8)
            if residual.dtype != torch.bfloat16 or tuple(residual.shape) != (m, n):
                raise ValueError("out_bf16 with a residual takes a bf16 [M, N] residual stream")
            residual = residual.to(self.device).contiguous()
        else:
            residual = None if residual is None else self._f32(residual, (m, n))
        out = torch.empty(m, n, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=self.device)
        _lib.check(self.lib.fern_gemm_mx8(self._h, _ptr(a8), k, _ptr(sa), sa.shape[1], _ptr(w8), k, _ptr(sw), sw.shape[1], _ptr(bias),
                                          _ptr(residual), _ptr(out), n, m, n, k, int(epilogue), int(bool(out_bf16)), _stream()), "fern_gemm_mx8")
        return out

    def gemm_mx8_quant(self, a8, sa, w8, sw, bias=None, epilogue=EPI_BIAS):
        """gemm_mx8 with the output quantised in the epilogue: returns (e4m3fn bytes [M,N], block scales [N/128, M, 4])."""
        m, k = a8.shape
        n = w8.shape[0]
        bias = None if bias is None else self._f32(bias, (n,))
        out = torch.empty(m, n, dtype=torch.uint8, device=self.device)
        sc = torch.empty(n // 128, m, 4, dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.fern_gemm_mx8_quant(self._h, _ptr(a8), k, _ptr(sa), sa.shape[1], _ptr(w8), k, _ptr(sw), sw.shape[1], _ptr(bias),
                                                _ptr(out), n, _ptr(sc), m, m, n, k, int(epilogue), _stream()), "fern_gemm_mx8_quant")
        return out, sc

    def layernorm(self, x, gamma, beta, eps: float, residual=None) -> torch.Tensor:
        x = self._f32(x)
        rows, d = x.shape
        residual = None if residual is None else self._f32(residual, (rows, d))
        gamma, beta = self._f32(gamma, (d,)), self._f32(beta, (d,))   # keep alive until the launch is queued
        out = torch.empty_like(x)
        _lib.check(self.lib.fern_layernorm(self._h, _ptr(x), _ptr(residual), _ptr(gamma), _ptr(beta), _ptr(out), rows, d,
                                           float(eps), _stream()), "fern_layernorm")
        return out

    def attention(self, q, k, v, heads: int, causal=False, scale=None) -> torch.Tensor:
        """q [B,Sq,W], k/v [B,Sk,W] -> [B,Sq,W] (W = heads * head_dim)."""
        q, k, v = self._f32(q), self._f32(k), self._f32(v)
        b, sq, w = q.shape
        sk = k.shape[1]
        hd = w // heads
        out = torch.empty_like(q)
        sc = float(scale) if scale is not None else hd ** -0.5
        _lib.check(self.lib.fern_attention(self._h, _ptr(q), w, _ptr(k), w, _ptr(v), w, _ptr(out), w, b, heads, hd, sq, sk,
                                           int(bool(causal)), sc, _stream()), "fern_attention")
        return out

    def attention_bf16(self, q, k, v, heads: int, causal=False, scale=None) -> torch.Tensor:
        """bf16 operand form of `attention`: q/k/v bf16 [B,S,W] (fp32 inputs are rounded first) -> bf16 [B,Sq,W]."""
        def b16(x):
            if x.dtype != torch.bfloat16:
                x3 = self._f32(x)
                return self.to_bf16(x3.reshape(-1, x3.shape[-1])).reshape(x3.shape)
            return x.to(self.device).contiguous()
        q, k, v = b16(q), b16(k), b16(v)
        b, sq, w = q.shape
        sk = k.shape[1]
        hd = w // heads
        out = torch.empty_like(q)
        sc = float(scale) if scale is not None else hd ** -0.5
        _lib.check(self.lib.fern_attention_bf16(self._h, _ptr(q), w, _ptr(k), w, _ptr(v), w, _ptr(out), w, b, heads, hd, sq, sk,
                                                int(bool(causal)), sc, _stream()), "fern_attention_bf16")
        return out

    # ---- profiling ----------------------------------------------------------------------------
    def prof_enable(self, on: bool) -> None:
        _lib.check(self.lib.fern_prof_enable(self._h, int(on)), "fern_prof_enable")

    def prof_collect(self) -> Dict[str, float]:
        st = _lib.ProfStats()
        _lib.check(self.lib.fern_prof_collect(self._h, C.byref(st)), "fern_prof_collect")
        return {f: getattr(st, f) for f, _ in st._fields_}

    def tuner_export(self) -> str:
        """The GEMM tuner's per-shape tile choices so far (text; a file of it named by FERN_GEMM_TILES pins them)."""
        n = self.lib.fern_tuner_export(None, 0)
        buf = C.create_string_buffer(int(n) + 1)
        self.lib.fern_tuner_export(buf, int(n) + 1)
        return buf.value.decode()

    def ws_generation(self) -> int:
        """Changes whenever this context has freed workspace memory an earlier call used (include/fern.h: fern_ws_generation):
        a hipGraph captured from this engine's calls is stale once the value moves."""
        return int(self.lib.fern_ws_generation(self._h))

    def tuner_import(self, text: str) -> None:
        """Adopt another process's `tuner_export()` for the shapes it lists (all tile choices are bit-identical: speed only)."""
        _lib.check(self.lib.fern_tuner_import(text.encode()), "fern_tuner_import")

    def tuner_set_concurrency(self, lanes: int) -> None:
        """Tell the GEMM tuner how many batches are kept in flight on separate streams (include/fern.h:
        fern_tuner_set_concurrency): shapes tuned afterwards are scored for pipeline throughput, not stand-alone latency."""
        _lib.check(self.lib.fern_tuner_set_concurrency(int(lanes)), "fern_tuner_set_concurrency")

    def tuner_force_config(self, family: str, cfg: int) -> None:
        """Force one tile configuration of a GEMM family ("f32", "f32x3", "bf16", "fp8", "mx8") process-wide; cfg < 0 releases it
        (include/fern.h: fern_tuner_force_config).  Results never depend on it."""
        _lib.check(self.lib.fern_tuner_force_config(family.encode(), int(cfg)), "fern_tuner_force_config")

    def sync(self) -> None:
        _lib.check(self.lib.fern_sync(self._h, _stream()), "fern_sync")
