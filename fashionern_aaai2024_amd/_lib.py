"""ctypes binding of libfern.so (include/fern.h).  No fallback: if the HIP library cannot be loaded the
import of this module raises, and every product entry point fails with it."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

c_void_p, c_int, c_i64, c_float = C.c_void_p, C.c_int, C.c_int64, C.c_float

# One hardware queue per pipeline lane: the ROCm runtime maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that
# share one run back to back -- with 8, four lanes of a ComposedQueryPipeline overlap where with 4 the fourth LOSES throughput (bench.py;
# profiles/r06_lanes_hwq.txt).  Effective only when set before the process's first HIP call; a value the user exported wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


class ClipConfigC(C.Structure):
    _fields_ = [(n, c_int) for n in (
        "embed_dim", "image_size", "patch_size", "v_width", "v_layers", "v_heads", "v_mlp",
        "context_length", "vocab_size", "t_width", "t_heads", "t_layers", "t_mlp", "v_arch")] + [
        ("r_layers", c_int * 4), ("r_width", c_int), ("r_heads", c_int)]


class ProfStats(C.Structure):
    _fields_ = [("gemm_ms", C.c_double), ("gemm_flops", C.c_double), ("gemm_launches", c_i64),
                ("attn_ms", C.c_double), ("attn_flops", C.c_double), ("attn_launches", c_i64),
                ("topk_ms", C.c_double), ("topk_launches", c_i64),
                ("sweep_ms", C.c_double), ("sweep_bytes", C.c_double), ("sweep_launches", c_i64),
                ("gemm_fp8_ms", C.c_double), ("gemm_fp8_flops", C.c_double), ("gemm_fp8_launches", c_i64),
                ("gemm_bf16_ms", C.c_double), ("gemm_bf16_flops", C.c_double), ("gemm_bf16_launches", c_i64),
                ("gemm_alg_bytes", C.c_double), ("gemm_dispatches", c_i64),
                ("gemm_mx8_ms", C.c_double), ("gemm_mx8_flops", C.c_double), ("gemm_mx8_launches", c_i64),
                ("gemm_mx8_bf16_flops", C.c_double)]


# name -> (restype, argtypes); must list every symbol include/fern.h declares (tests check this)
SIGNATURES = {
    "fern_abi_version": (c_int, []),
    "fern_last_error": (C.c_char_p, []),
    "fern_ctx_create": (c_int, [c_int, C.POINTER(c_void_p)]),
    "fern_ctx_fork": (c_int, [c_void_p, C.POINTER(c_void_p)]),
    "fern_ctx_destroy": (c_int, [c_void_p]),
    "fern_sync": (c_int, [c_void_p, c_void_p]),
    "fern_load_tensor": (c_int, [c_void_p, C.c_char_p, c_void_p, c_int, c_int, C.POINTER(c_i64)]),
    "fern_finalize_fusion": (c_int, [c_void_p, c_int, c_int]),
    "fern_finalize_clip": (c_int, [c_void_p, C.POINTER(ClipConfigC)]),
    "fern_vit_encode_image": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fern_text_encode": (c_int, [c_void_p, c_void_p, c_void_p, C.POINTER(c_i64), c_void_p, c_void_p, c_int, c_void_p]),
    "fern_encode_pair": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "fern_dvr_fuse": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fern_index_fuse": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "fern_combiner": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "fern_visual_sr": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_i64, c_void_p]),
    "fern_finalize_clip4cir": (c_int, [c_void_p]),
    "fern_combiner_clip4cir": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_void_p]),
    "fern_element_wise_sum": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "fern_l2_normalize": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "fern_resample_u8_horizontal": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "fern_resample_u8_vertical": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "fern_u8_to_normalized_chw": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_int, c_i64, c_void_p, c_int, c_int, c_int,
                                          C.POINTER(c_float), C.POINTER(c_float), c_void_p]),
    "fern_sim_topk": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_int, c_void_p, c_void_p, c_i64,
                              c_void_p, c_void_p]),
    "fern_gallery_to_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_i64, c_int, c_void_p]),
    "fern_sim_topk_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_int, c_void_p, c_void_p, c_i64, c_void_p, c_void_p]),
    "fern_gallery_prepare": (c_int, [c_void_p, c_void_p, c_i64, c_int, c_void_p, c_void_p, c_void_p]),
    "fern_sim_topk_prefiltered": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_int, c_void_p, c_void_p, c_i64,
                                          c_void_p, c_void_p]),
    "fern_rank_set_strategy": (c_int, [c_void_p, c_int]),
    "fern_sweep_bf16_scores": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_i64, c_int, c_void_p, c_i64, c_void_p, c_i64, c_void_p]),
    "fern_gather_scores": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fern_topk_merge": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fern_gemm": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64,
                          c_int, c_int, c_int, c_int, c_void_p]),
    "fern_batch_classification_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "fern_set_precision": (c_int, [c_void_p, c_int]),
    "fern_get_precision": (c_int, [c_void_p]),
    "fern_gemm_bf16": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64,
                               c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fern_quantize_rows_fp8": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_int, c_void_p]),
    "fern_gemm_fp8": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_void_p, c_i64,
                              c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fern_quantize_mx8": (c_int, [c_void_p, c_void_p, c_int, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_i64, c_int, c_void_p]),
    "fern_gemm_mx8": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_void_p, c_i64,
                              c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "fern_gemm_mx8_quant": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_void_p, c_i64,
                                    c_void_p, c_i64, c_int, c_int, c_int, c_int, c_void_p]),
    "fern_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_i64, c_int, c_float, c_void_p]),
    "fern_attention": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64,
                               c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "fern_attention_bf16": (c_int, [c_void_p, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64, c_void_p, c_i64,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "fern_prof_enable": (c_int, [c_void_p, c_int]),
    "fern_prof_collect": (c_int, [c_void_p, C.POINTER(ProfStats)]),
    "fern_tuner_export": (c_i64, [C.c_char_p, c_i64]),
    "fern_tuner_import": (c_int, [C.c_char_p]),
    "fern_tuner_set_concurrency": (c_int, [c_int]),
    "fern_tuner_force_config": (c_int, [C.c_char_p, c_int]),
    "fern_ws_generation": (C.c_uint64, [c_void_p]),
}

_lib = None


def lib_path() -> str:
    return _build.LIB


def load() -> C.CDLL:
    """Load libfern.so (building it first if it is missing/stale and hipcc exists).  Raises on failure."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.ensure_built()
    if not os.path.exists(path):
        raise RuntimeError(f"libfern.so not found at {path}: build it with `python -m fashionern_aaai2024_amd.build`")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library lacks a declared symbol
        fn.restype, fn.argtypes = res, args
    if lib.fern_abi_version() != 3:
        raise RuntimeError("libfern.so ABI version mismatch")
    _lib = lib
    return lib


class FernError(RuntimeError):
    pass


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().fern_last_error()
        raise FernError(f"{what} failed ({code}): {msg.decode() if msg else '?'}")
