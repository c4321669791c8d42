"""CPU: libfern.so builds, loads through ctypes and exports every symbol include/fern.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re
import subprocess

from fashionern_aaai2024_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "fern.h")).read()
    return sorted(set(re.findall(r"FERN_API\s+[\w\s\*]+?\b(fern_\w+)\s*\(", hdr)))


def test_library_builds_and_exports_every_declared_symbol():
    path = build.ensure_built()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libfern.so does not export {n}"
    assert sorted(_lib.SIGNATURES) == names, "ctypes signature table and fern.h disagree"


def test_ctypes_loader_and_error_reporting_without_gpu():
    lib = _lib.load()
    assert lib.fern_abi_version() == 3
    # argument validation happens before any HIP call
    assert lib.fern_finalize_fusion(None, 512, 7) == -1
    assert b"ctx is NULL" in lib.fern_last_error()
    assert lib.fern_load_tensor(None, b"x", None, 0, 0, None) == -1
    # the block-scaled fp8 entry points validate before touching the device too
    assert lib.fern_quantize_mx8(None, None, 0, 0, None, 0, None, 0, 0, 128, None) == -1
    assert lib.fern_gemm_mx8(None, None, 0, None, 0, None, 0, None, 0, None, None, None, 0, 1, 1, 128, 0, 0, None) == -1
    assert lib.fern_gemm_mx8_quant(None, None, 0, None, 0, None, 0, None, 0, None, None, 0, None, 0, 1, 128, 128, 0, None) == -1
    assert b"fern_gemm_mx8_quant" in lib.fern_last_error()


def test_code_object_targets_gfx950_only():
    out = subprocess.run(["strings", "-n", "6", build.LIB], capture_output=True, text=True).stdout
    assert "gfx950" in out
    assert "gfx90a" not in out and "gfx942" not in out and "sm_" not in out
