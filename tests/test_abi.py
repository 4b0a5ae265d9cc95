"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol include/bltvqg_hip.h declares, and the
ctypes signature table agrees with the header (argument counts)."""
import os
import re

import pytest

import bltvqg_amd  # noqa: F401
from bltvqg_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


HEADER = os.path.join(ROOT, "include", "bltvqg_hip.h")


def _header_functions(path=None):
    src = open(path or HEADER).read()
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


def test_experiment_surface_is_not_in_the_product_library():
    """The timing aids, the hardware-id probe and the fused operators that were measured and not adopted live in
    include/bltvqg_hip_experiments.h and the -DBLT_EXPERIMENTS build only: the shipped library exports none of them,
    the product header declares none of them."""
    lib = _lib.load()
    decl = _header_functions()
    exp = _header_functions(os.path.join(os.path.dirname(HEADER), "bltvqg_hip_experiments.h"))
    assert set(exp) == set(_lib.EXPERIMENT_SIGNATURES), set(exp) ^ set(_lib.EXPERIMENT_SIGNATURES)
    for name, n in exp.items():
        assert name not in decl
        assert not hasattr(lib, name), "experiment entry point exported by the product library: " + name
        assert len(_lib.EXPERIMENT_SIGNATURES[name][1]) == n, (name, n)
    e = _lib.load_experiments()
    if e is not None:      # when built: it exports both surfaces
        for name in list(exp) + list(decl):
            assert hasattr(e, name), name


def test_ctypes_table_matches_header():
    decl = _header_functions()
    assert set(decl) == set(_lib.SIGNATURES), set(decl) ^ set(_lib.SIGNATURES)
    for name, n in decl.items():
        assert len(_lib.SIGNATURES[name][1]) == n, (name, n, len(_lib.SIGNATURES[name][1]))


def test_config_struct_layout():
    import ctypes
    assert ctypes.sizeof(_lib.Config) == 14 * 4 + 5 * 4 + 3 * 4 + 4      # + num_regions, region_dim, region_pool; + head_dim_true


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
    # bltvqg_gemm_ex goes through the same operand validation as bltvqg_gemm: a pitch below round8(K), a rowtab without its
    # index and a dropout probability of 1 are argument errors, not out-of-bounds LDS-DMA reads on the device
    buf = (ctypes.c_char * 4096)()
    a16 = ctypes.c_void_p((ctypes.addressof(buf) + 15) // 16 * 16)

    def gemm_ex(lda=64, ldb=64, rowtab=None, drop_p=0.0, tile=(0, 0)):
        return lib.bltvqg_gemm_ex(a16, lda, a16, ldb, a16, 64, 256, 64, 64, None, rowtab, None, 64, 0, drop_p, 0, 0, None, 0, 1.0, None, 0, None, 0, 0,
                                  tile[0], tile[1], None)
    for kw, what in ((dict(lda=56), b"lda"), (dict(ldb=56), b"ldb"), (dict(rowtab=a16), b"rowtab"), (dict(drop_p=1.0), b"dropout"),
                     (dict(tile=(64, 0)), b"tile")):
        rc = gemm_ex(**kw)
        assert rc < 0 and what in lib.bltvqg_last_error_string(), (kw, lib.bltvqg_last_error_string())


def test_host_paths_under_address_sanitizer():
    """`make asan` (csrc/Makefile): the C-ABI shim, the error path and the engine's host logic (parameter layout, workspace carve, bucket
    table) built with AddressSanitizer + UBSan, exercised WITHOUT a GPU: argument validation of the operator entry points, engine
    descriptors of every BASELINE configuration created / queried / destroyed.  (GPU ASan is not available on this pool.)"""
    import subprocess
    import sys
    csrc = os.path.join(ROOT, "blt-vqg_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asan", "-j8"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lib = os.path.join(ROOT, "blt-vqg_amd", "libbltvqg_hip_asan.so")
    assert os.path.exists(lib)
    rt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/clang", "-print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    assert os.path.exists(rt), rt
    code = r'''
import ctypes, sys
sys.path.insert(0, %r)
import bltvqg_amd
from bltvqg_amd import _lib
from bltvqg_amd.engine import StepEngine, make_config
lib = _lib.load()
assert lib.bltvqg_version() >= 100
assert lib.bltvqg_layernorm_fwd(0, None, None, None, None, None, None, 4, 12, 1e-5, None) < 0
assert lib.bltvqg_gemm(7, None, 0, 0, None, 0, 0, None, 0, 1, 1, 1, None, 0, 0.0, 0, 0, None, 0, 1.0, None, 0, 0, 0, 0, 0, None) < 0
assert lib.bltvqg_engine_forward(None, None, None, None, None, None, 0, 0, None) < 0
assert b"null engine" in lib.bltvqg_last_error_string()
assert lib.bltvqg_engine_profile_enable(None, 1) < 0 and lib.bltvqg_engine_share_optimizer_state(None, None) < 0
bad = _lib.Config(batch=4, hidden_dim=60, pwffn_dim=128, latent_dim=64, emb_dim=20, num_layers=1, num_heads=4, vocab_size=97,
                  len_context=5, len_posterior=21, len_target=20, image_h=64, image_w=64, dtype=0)
assert not lib.bltvqg_engine_create(ctypes.byref(bad))
for kw in (dict(batch=128, hidden_dim=256, pwffn_dim=512, latent_dim=256, num_layers=2, num_heads=4),
           dict(batch=256, hidden_dim=512, pwffn_dim=2048, latent_dim=512, num_layers=6, num_heads=8),
           dict(batch=64, hidden_dim=512, pwffn_dim=2048, latent_dim=512, num_layers=6, num_heads=8, num_regions=36, region_dim=2048)):
    for dtype in (0, 1):
        e = StepEngine(make_config(emb_dim=300, vocab_size=8000, dtype=dtype, **kw), "cpu")
        assert e.train_size > 0 and e.workspace_bytes > 0 and len(e.buckets()) >= 5
        assert sum(n for _, n, _ in e.buckets()) == e.train_size
        assert e.adam_steps() == (0, 0)
        e2 = StepEngine(make_config(emb_dim=300, vocab_size=8000, dtype=dtype, **dict(kw, batch=kw["batch"] - 1)), "cpu")
        _lib.check(lib.bltvqg_engine_share_optimizer_state(e2.h, e.h), "share")
        e.set_adam_steps(3, 1)
        assert e2.adam_steps() == (3, 1)
        del e                                   # the shared counters outlive the primary engine
        assert e2.adam_steps() == (3, 1)
        del e2
print("asan-host-ok")
''' % ROOT
    env = dict(os.environ, LD_PRELOAD=rt, BLTVQG_LIB=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0 and "asan-host-ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
    assert "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-3000:]
