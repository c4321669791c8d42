"""Tokenizer lookup for the harness (the reference calls ``open_clip.get_tokenizer(clip_model_name)``,
/root/reference/run/test/test_fiq.py:79).  open_clip and its BPE vocabulary are not available offline, so a
tokenizer must be registered by the caller (``register_tokenizer``) unless open_clip is importable."""
from __future__ import annotations

from typing import Callable, Dict

_REGISTRY: Dict[str, Callable] = {}


def register_tokenizer(name: str, fn: Callable) -> None:
    """fn(list[str] | str, context_length=77) -> int64 tensor [B, context_length]."""
    _REGISTRY[name] = fn


def get_tokenizer(name: str) -> Callable:
    if name in _REGISTRY:
        return _REGISTRY[name]
    try:
        import open_clip  # type: ignore
    except ImportError as e:
        raise RuntimeError(f"no tokenizer registered for {name!r} and open_clip is not installed; call "
                           "fashionern_aaai2024_amd.tokenizer.register_tokenizer(name, fn) first") from e
    return open_clip.get_tokenizer(name)
