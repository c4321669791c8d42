"""Build libfern.so (HIP, gfx950 only) in-tree with hipcc.

The shared library is git-ignored but travels to the GPU box with the repo snapshot, so the
normal flow is: ``python -m fashionern_aaai2024_amd.build`` (or ``__graft_entry__.build()``)
in the dev container, then run on the GPU.  ``ensure_built()`` is what the ctypes loader calls;
it rebuilds only when a source is newer than the library.
"""
from __future__ import annotations

import json
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libfern.so")
SOURCES = ["api.hip", "gemm.hip", "gemm_bf16.hip", "gemm_pp.hip", "attn.hip", "elem.hip", "topk.hip", "image.hip", "sweep_bf16.hip"]
HEADERS = ["kernels.h", "gemm_epilogue.h", "gemm_pp.h", os.path.join("..", "..", "include", "fern.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-DFERN_BUILD"]


RESOURCE_REPORT = os.path.join(CSRC, "build", "resource_usage.txt")


def parse_resource_usage(stderr: str) -> dict:
    """`-Rpass-analysis=kernel-resource-usage` remarks -> {kernel: {field: value}}."""
    out: dict = {}
    cur = None
    for line in stderr.splitlines():
        if "remark:" not in line:
            continue
        body = line.split("remark:", 1)[1].split("[-Rpass-analysis", 1)[0].strip()
        if body.startswith("Function Name:"):
            cur = body.split(":", 1)[1].strip()
            out[cur] = {}
        elif cur is not None and ":" in body:
            k, v = body.rsplit(":", 1)
            out[cur][k.strip()] = v.strip()
    return out


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

    usage: dict = {}

    import re

    def deps_time(path: str, seen=None) -> float:
        """Newest mtime of `path` and of every file it #include "..."s, transitively."""
        seen = set() if seen is None else seen
        path = os.path.normpath(path)
        if path in seen or not os.path.exists(path):
            return 0.0
        seen.add(path)
        t = os.path.getmtime(path)
        for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(path).read(), flags=re.M):
            t = max(t, deps_time(os.path.join(os.path.dirname(path), inc), seen))
        return t

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        side = obj + ".usage.json"      # the resource report of the object, so an object that is reused still gets checked for spills
        newest = deps_time(os.path.join(CSRC, src))
        if not force and os.path.exists(obj) and os.path.exists(side) and os.path.getmtime(obj) > newest:
            usage[src] = json.load(open(side))
            return obj
        cmd = [hipcc, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        usage[src] = parse_resource_usage(r.stderr)
        json.dump(usage[src], open(side, "w"))
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    # Every kernel's registers / scratch / LDS as the compiler reports them.  A kernel that SPILLS is a build error: scratch
    # traffic inside an MFMA loop (or an epilogue that drags the whole kernel's occupancy down) is a silent 2x, and it has
    # happened twice -- an epilogue added to a 128-VGPR GEMM, a 16-wave attention kernel -- without any test noticing.
    with open(RESOURCE_REPORT, "w") as f:
        for src in SOURCES:
            for name, u in sorted(usage.get(src, {}).items()):
                f.write(f"{src} {name} vgpr={u.get('VGPRs', '?')} sgpr={u.get('TotalSGPRs', '?')} scratch={u.get('ScratchSize [bytes/lane]', '?')} "
                        f"vgpr_spill={u.get('VGPRs Spill', '?')} sgpr_spill={u.get('SGPRs Spill', '?')} lds={u.get('LDS Size [bytes/block]', '?')} "
                        f"occupancy={u.get('Occupancy [waves/SIMD]', '?')}\n")
    # (a "VGPRs Spill" count with ScratchSize 0 is a spill into the wave's AGPR half of the register file -- v_accvgpr moves, no memory
    # traffic: a 256-thread kernel that keeps 192 row values in registers uses it on purpose; what is an error is SCRATCH memory)
    spilled = [(src, n, u) for src in SOURCES for n, u in usage.get(src, {}).items() if int(u.get("ScratchSize [bytes/lane]", 0))]
    if spilled and not os.environ.get("FERN_ALLOW_SPILLS"):
        lines = "\n".join(f"  {src}: {n}: {u.get('VGPRs Spill')} VGPRs spilled, {u.get('ScratchSize [bytes/lane]')} B/lane scratch" for src, n, u in spilled)
        raise RuntimeError("kernels spill registers (restructure them, or set FERN_ALLOW_SPILLS=1 to build anyway):\n" + lines)
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
            _hipcc()
        except RuntimeError:
            if os.path.exists(LIB):
                return LIB      # no toolchain on this machine: use the library that travelled with the tree
            raise
        build_lib()             # a failing compile (or a kernel that spills) is an error, never a silent fall-back to the old library
    return LIB


if __name__ == "__main__":
    build_lib(force="--force" in sys.argv)
