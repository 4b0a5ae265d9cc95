"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/bltvqg_hip.h declares, and the
ctypes signature table agrees with the header (argument counts)."""
import os
import re

import pytest

import bltvqg_amd  # noqa: F401
from bltvqg_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "bltvqg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(bltvqg_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(1), m.group(2).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        out[name] = n
    return out


def test_library_is_built_and_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libbltvqg_hip.so not built (make -C blt-vqg_amd/csrc)"
    lib = _lib.load()
    decl = _header_functions()
    assert len(decl) >= 45
    for name in decl:
        assert hasattr(lib, name), "missing export " + name


def test_ctypes_table_matches_header():
    decl = _header_functions()
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, n in decl.items():
        assert len(_lib.SIGNATURES[name][1]) == n, (name, n, len(_lib.SIGNATURES[name][1]))


def test_config_struct_layout():
    import ctypes
    assert ctypes.sizeof(_lib.Config) == 14 * 4 + 5 * 4 + 2 * 4      # + num_regions, region_dim


def test_argument_validation_without_gpu():
    """Argument checks run before any HIP call, so they can be exercised on a machine without a GPU."""
    lib = _lib.load()
    assert lib.bltvqg_version() >= 100
    rc = lib.bltvqg_layernorm_fwd(0, None, None, None, None, None, None, 4, 12, 1e-5, None)
    assert rc < 0 and b"layernorm_fwd" in lib.bltvqg_last_error_string()
    rc = lib.bltvqg_gemm(7, None, 0, 0, None, 0, 0, None, 0, 1, 1, 1, None, 0, 0.0, 0, 0, None, 0, 1.0, None, 0, 0, 0, 0, 0, None)
    assert rc < 0 and b"dtype" in lib.bltvqg_last_error_string()
    with pytest.raises(_lib.HipError):
        _lib.check(rc, "gemm")
    cfg = _lib.Config(batch=4, hidden_dim=60, pwffn_dim=128, latent_dim=64, emb_dim=20, num_layers=1, num_heads=4, vocab_size=97,
                      len_context=5, len_posterior=21, len_target=20, image_h=64, image_w=64, dtype=0)
    import ctypes
    assert not lib.bltvqg_engine_create(ctypes.byref(cfg))     # hidden_dim % 8 != 0 is rejected
