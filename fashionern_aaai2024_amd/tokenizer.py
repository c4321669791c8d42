"""Tokenizer lookup for the harness (the reference calls ``open_clip.get_tokenizer(clip_model_name)``,
/root/reference/run/test/test_fiq.py:79).  Resolution order of ``get_tokenizer(name)``: a tokenizer registered with
``register_tokenizer``; open_clip's own if that package is importable; otherwise the built-in ``ClipBpeTokenizer`` when
``FERN_CLIP_BPE_VOCAB`` names a CLIP BPE merges file (``bpe_simple_vocab_16e6.txt.gz``, not shipped: no network here).

``ClipBpeTokenizer`` restates the published CLIP byte-level BPE (third party: openai/CLIP ``simple_tokenizer.py``, which
open_clip 2.20.0 re-exports; neither is under /root/reference): printable-byte alphabet, merges applied in rank order on
lower-cased, whitespace-collapsed text split by the CLIP pattern, ``<|startoftext|>`` / ``<|endoftext|>`` framing, zero padding
and truncation to ``context_length`` with the end token kept.  PARITY of the merge TABLE is UNPINNED (no vocabulary file, no
open_clip offline); how a table is APPLIED is pinned against an independent implementation, the `tokenizers` library's CLIP
pipeline behind transformers' CLIPTokenizer, on a shared synthetic table (tests/test_host_cpu.py: 315 strings, identical ids).
``ftfy`` text repair is not applied.
"""
from __future__ import annotations

import gzip
import html
import os
from functools import lru_cache
from typing import Callable, Dict, Iterable, List, Sequence, Tuple, Union

_REGISTRY: Dict[str, Callable] = {}


@lru_cache()
def _byte_alphabet() -> Dict[int, str]:
    """Every byte value as one printable unicode character.  ORDER MATTERS: the dict's insertion order is CLIP's vocabulary
    order of the 256 single-byte symbols -- first the 188 printable Latin-1 bytes ('!'..'~', 0xA1..0xAC, 0xAE..0xFF), mapped to
    themselves (so '!' is id 0, 'a' id 64), then the remaining 68 bytes in increasing byte order, mapped to code points 256,
    257, ... (ids 188..255).  ``list(_byte_alphabet().values())`` is therefore rows 0..255 of ``token_embedding``."""
    printable = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAD)) + list(range(0xAE, 0x100))
    table = {b: chr(b) for b in printable}
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + len(table) - len(printable))
    return table


def _clean(text: str) -> str:
    text = html.unescape(html.unescape(text)).strip()
    return " ".join(text.split()).strip()


class ClipBpeTokenizer:
    """``tok(texts, context_length=77) -> int64 [B, context_length]`` with CLIP's vocabulary layout: 256 byte symbols, the same
    256 with the end-of-word mark, one entry per merge, then the start and end tokens (49 408 entries for the 48 894 merges of
    the released file)."""

    WORD_END = "</w>"

    def __init__(self, merges: Union[str, Iterable[Tuple[str, str]]], context_length: int = 77, max_merges: int = 49152 - 256 - 2):
        import regex
        import torch
        self._torch = torch
        if isinstance(merges, str):
            opener = gzip.open if merges.endswith(".gz") else open
            with opener(merges, "rt", encoding="utf-8") as f:
                lines = f.read().split("\n")
            pairs = [tuple(line.split()) for line in lines[1:1 + max_merges]]      # first line is a header
            pairs = [p for p in pairs if len(p) == 2]
        else:
            pairs = [tuple(p) for p in merges]
        alphabet = list(_byte_alphabet().values())
        vocab = alphabet + [c + self.WORD_END for c in alphabet] + ["".join(p) for p in pairs] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}
        self.rank = {p: i for i, p in enumerate(pairs)}
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        self.context_length = int(context_length)
        self._cache: Dict[str, List[str]] = {}
        self._split = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+",
                                    regex.IGNORECASE)

    @property
    def vocab_size(self) -> int:
        return len(self.encoder)

    def _bpe(self, word: str) -> List[str]:
        """Merge adjacent symbols of one word, always the pair with the lowest rank first, until no ranked pair is left."""
        hit = self._cache.get(word)
        if hit is not None:
            return hit
        syms = list(word[:-1]) + [word[-1] + self.WORD_END]
        while len(syms) > 1:
            best, best_rank = -1, None
            for i in range(len(syms) - 1):
                r = self.rank.get((syms[i], syms[i + 1]))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = i, r
            if best_rank is None:
                break
            a, b = syms[best], syms[best + 1]
            merged, i = [], 0
            while i < len(syms):                       # merge EVERY occurrence of the chosen pair, left to right
                if i + 1 < len(syms) and syms[i] == a and syms[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(syms[i])
                    i += 1
            syms = merged
        self._cache[word] = syms
        return syms

    def encode(self, text: str) -> List[int]:
        alphabet = _byte_alphabet()
        ids: List[int] = []
        for piece in self._split.findall(_clean(text).lower()):
            word = "".join(alphabet[b] for b in piece.encode("utf-8"))
            ids.extend(self.encoder[s] for s in self._bpe(word))
        return ids

    def __call__(self, texts: Union[str, Sequence[str]], context_length: int = None):
        n = self.context_length if context_length is None else int(context_length)
        if isinstance(texts, str):
            texts = [texts]
        out = self._torch.zeros(len(texts), n, dtype=self._torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot] + self.encode(t) + [self.eot]
            if len(ids) > n:
                ids = ids[:n]
                ids[-1] = self.eot                       # truncated captions keep the end token (argmax pooling needs it)
            out[i, :len(ids)] = self._torch.tensor(ids)
        return out


def register_tokenizer(name: str, fn: Callable) -> None:
    """fn(list[str] | str, context_length=77) -> int64 tensor [B, context_length]."""
    _REGISTRY[name] = fn


def get_tokenizer(name: str) -> Callable:
    if name in _REGISTRY:
        return _REGISTRY[name]
    try:
        import open_clip  # type: ignore
        return open_clip.get_tokenizer(name)
    except ImportError:
        pass
    vocab = os.environ.get("FERN_CLIP_BPE_VOCAB")
    if vocab:
        tok = ClipBpeTokenizer(vocab)
        _REGISTRY[name] = tok
        return tok
    raise RuntimeError(f"no tokenizer registered for {name!r}, open_clip is not installed and FERN_CLIP_BPE_VOCAB is not set; call "
                       "fashionern_aaai2024_amd.tokenizer.register_tokenizer(name, fn) or point FERN_CLIP_BPE_VOCAB at "
                       "bpe_simple_vocab_16e6.txt.gz")
