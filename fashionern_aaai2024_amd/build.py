"""Build libfern.so (HIP, gfx950 only) in-tree with hipcc.

The shared library is git-ignored but travels to the GPU box with the repo snapshot, so the
normal flow is: ``python -m fashionern_aaai2024_amd.build`` (or ``__graft_entry__.build()``)
in the dev container, then run on the GPU.  ``ensure_built()`` is what the ctypes loader calls;
it rebuilds only when a source is newer than the library.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libfern.so")
SOURCES = ["api.hip", "gemm.hip", "gemm_bf16.hip", "attn.hip", "elem.hip", "topk.hip", "image.hip", "sweep_bf16.hip"]
HEADERS = ["kernels.h", "gemm_epilogue.h", os.path.join("..", "..", "include", "fern.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DFERN_BUILD"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libfern.so cannot be built (ROCm toolchain required)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = LIB + ".tmp"
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr}")
    os.replace(tmp, LIB)
    if verbose:
        print(f"built {LIB}", file=sys.stderr)
    return LIB


def ensure_built() -> str:
    """Return the library path, building it if hipcc is available and it is missing/stale."""
    if _stale():
        try:
            build_lib()
        except RuntimeError:
            if not os.path.exists(LIB):
                raise
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
