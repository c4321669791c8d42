"""Loader of oracle/chain.c: the bit-exact fp32 fma-chain restatement of the cosine sweep.  TEST INFRASTRUCTURE ONLY.

`build()` compiles the C file with gcc into oracle/_build/ (git-ignored; it travels to the GPU box with the snapshot and is
rebuilt there if the host CPU lacks the instructions it was built for).  `chain_scores` / `chain_topk` are what the parity
tests call; `chain_topk` ranks with the path's tie rule (score descending, gallery index ascending)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "chain.c")
OUT = os.path.join(HERE, "_build", "libfern_oracle_chain.so")
_lib = None


def build(force: bool = False) -> str:
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    # -mfma only if this CPU has it (then fmaf is one instruction); without it glibc's software fmaf gives the same bits
    flags = ["-O2", "-ffp-contract=off", "-shared", "-fPIC"]
    try:
        if " fma " in open("/proc/cpuinfo").read():
            flags.append("-mfma")
    except OSError:
        pass
    subprocess.run(["gcc", *flags, SRC, "-o", OUT, "-lm"], check=True)
    return OUT


def _load():
    global _lib
    if _lib is None:
        try:
            _lib = C.CDLL(build())
            _lib.fern_oracle_chain_scores  # noqa: B018
        except (OSError, AttributeError):
            _lib = C.CDLL(build(force=True))
        _lib.fern_oracle_chain_scores.restype = None
        _lib.fern_oracle_chain_scores.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]
    return _lib


def chain_scores(q: np.ndarray, g: np.ndarray) -> np.ndarray:
    q = np.ascontiguousarray(q, dtype=np.float32)
    g = np.ascontiguousarray(g, dtype=np.float32)
    assert q.shape[1] == g.shape[1] and q.shape[1] % 8 == 0
    out = np.empty((q.shape[0], g.shape[0]), dtype=np.float32)
    _load().fern_oracle_chain_scores(q.ctypes.data, g.ctypes.data, out.ctypes.data, q.shape[0], g.shape[0], q.shape[1])
    return out


def chain_topk(q: np.ndarray, g: np.ndarray, k: int):
    """(scores [B,k] f32, idx [B,k] int32): top-k of the chain scores, score descending then index ascending."""
    s = chain_scores(q, g)
    order = np.argsort(-s, axis=1, kind="stable")[:, :k]          # stable: equal scores keep ascending index order
    return np.take_along_axis(s, order, axis=1), order.astype(np.int32)
