"""ctypes binding of libbltvqg_hip.so (C ABI declared in include/bltvqg_hip.h).

There is deliberately NO fallback: if the HIP library is missing the import of the product path fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BLTVQG_LIB: load another build of the SAME library (A/B timing of kernel changes on one GPU box); there is still no fallback
LIB_PATH = os.environ.get("BLTVQG_LIB") or os.path.join(_HERE, "libbltvqg_hip.so")

F32, BF16 = 0, 1

P = ctypes.c_void_p
I = ctypes.c_int
L = ctypes.c_int64
F = ctypes.c_float
U64 = ctypes.c_uint64
U32 = ctypes.c_uint32
S = ctypes.c_char_p


class Config(ctypes.Structure):
    """struct bltvqg_config (include/bltvqg_hip.h)."""
    _fields_ = [
        ("batch", ctypes.c_int32), ("hidden_dim", ctypes.c_int32), ("pwffn_dim", ctypes.c_int32), ("latent_dim", ctypes.c_int32),
        ("emb_dim", ctypes.c_int32), ("num_layers", ctypes.c_int32), ("num_heads", ctypes.c_int32), ("vocab_size", ctypes.c_int32),
        ("len_context", ctypes.c_int32), ("len_posterior", ctypes.c_int32), ("len_target", ctypes.c_int32),
        ("image_h", ctypes.c_int32), ("image_w", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("attention_dropout", ctypes.c_float), ("relu_dropout", ctypes.c_float),
        ("kl_ceiling", ctypes.c_float), ("aux_ceiling", ctypes.c_float), ("image_recon_lambda", ctypes.c_float),
        ("num_regions", ctypes.c_int32), ("region_dim", ctypes.c_int32), ("region_pool", ctypes.c_int32),
        ("head_dim_true", ctypes.c_int32),
    ]


class LinearDesc(ctypes.Structure):
    """struct bltvqg_linear_desc (include/bltvqg_hip_experiments.h): one problem of bltvqg_linear_pair."""
    _fields_ = [
        ("A", ctypes.c_void_p), ("lda", ctypes.c_int32), ("W", ctypes.c_void_p), ("ldw", ctypes.c_int32), ("C", ctypes.c_void_p), ("ldc", ctypes.c_int32),
        ("M", ctypes.c_int32), ("bias", ctypes.c_void_p), ("maskY", ctypes.c_void_p), ("ldm", ctypes.c_int32), ("C2", ctypes.c_void_p), ("ldc2", ctypes.c_int32),
        ("R", ctypes.c_void_p), ("ldr", ctypes.c_int32), ("stream_id", ctypes.c_uint32), ("fold_s", ctypes.c_void_p), ("fold_c", ctypes.c_void_p),
        ("row_stat", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p), ("out_stat", ctypes.c_void_p),
    ]


# name -> (restype, argtypes) ; must list every function the header declares (tests/test_abi.py checks this)
SIGNATURES = {
    "bltvqg_version": (I, []),
    "bltvqg_last_error_string": (S, []),
    "bltvqg_debug_set": (None, [I, I]),
    "bltvqg_debug_get": (I, [I]),
    "bltvqg_build_has_ablations": (I, []),
    "bltvqg_gemm": (I, [I, P, I, I, P, I, I, P, I, I, I, I, P, I, F, U64, U32, P, I, F, P, I, I, I, I, I, P]),
    "bltvqg_gemm_ex": (I, [P, I, P, I, P, I, I, I, I, P, P, P, I, I, F, U64, U32, P, I, F, P, I, P, I, I, I, I, P]),
    "bltvqg_gemm_rowstat": (I, [P, I, P, I, P, I, I, I, I, P, I, F, U64, U32, P, I, P, I, P, I, I, I, P]),
    "bltvqg_gemm_rowstat_parts": (I, [I, I, I, I]),
    "bltvqg_ln_fold_prepare": (I, [P, I, I, P, P, P, P, P, P, P]),
    "bltvqg_linear_ln_folded": (I, [P, I, P, I, P, I, I, I, I, P, P, P, I, I, P, P, F, I, F, U64, U32, I, I, P]),
    "bltvqg_linear_wgrad_group": (I, [I, P, P, P, P, P, P, P, P, P, P, P, L, P]),
    "bltvqg_linear_wgrad": (I, [I, P, I, P, I, P, I, P, I, I, I, I, P]),
    "bltvqg_conv2d": (I, [I, P, P, P, I, I, I, I, I, I, I, I, I, P, P, P]),
    "bltvqg_conv2d_stat_rows": (I, [I, I, I, I, I, I, I, I]),
    "bltvqg_pp_pixels": (L, [I, I, I]),
    "bltvqg_pp_guard_front": (I, []),
    "bltvqg_pp_guard_tail": (I, []),
    "bltvqg_conv3x3_pp": (I, [P, P, P, I, I, I, I, I, P, P, P]),
    "bltvqg_conv3x3_pp_bn_relu_in": (I, [P, P, P, P, P, I, I, I, I, I, P, P, P]),
    "bltvqg_conv3x3_pp_stat_rows": (I, [I, I, I]),
    "bltvqg_conv2d_pp": (I, [I, P, P, P, I, I, I, I, I, I, I, I, I, I, I, P, P, P]),
    "bltvqg_conv2d_pp_stat_rows": (I, [I, I, I, I, I, I, I, I, I, I, I]),
    "bltvqg_bn_apply_pp": (I, [I, P, P, P, P, P, I, I, I, I, I, P]),
    "bltvqg_bn_relu_maxpool_pp": (I, [I, P, P, P, P, I, I, I, I, P]),
    "bltvqg_avgpool_pp": (I, [I, P, P, I, I, I, I, P]),
    "bltvqg_img_pack": (I, [I, P, P, I, I, I, I, I, I, I, I, I, P]),
    "bltvqg_conv_pack_w": (I, [I, P, P, I, I, I, I, I, I, P]),
    "bltvqg_conv_stem": (I, [I, P, P, P, I, I, I, I, I, I, P, P, P]),
    "bltvqg_conv_stem_stat_rows": (I, [I, I, I, I]),
    "bltvqg_conv_stem_pool": (I, [P, P, P, P, I, I, I, I, I, P, P, P]),
    "bltvqg_conv_stem_pool_ok": (I, [I, I, I, I, I, I]),
    "bltvqg_conv_stem_pool_stat_rows": (I, [I, I, I]),
    "bltvqg_layernorm_fwd": (I, [I, P, P, P, P, P, P, L, I, F, P]),
    "bltvqg_layernorm_bwd": (I, [I, P, P, P, P, P, P, P, P, P, L, I, P]),
    "bltvqg_bn_scratch_doubles": (I, [I]),
    "bltvqg_bn_finalize": (I, [P, P, I, I, L, P, P, F, F, P, P, P, P, P, P]),
    "bltvqg_bn_apply": (I, [I, P, P, P, P, P, L, I, I, P]),
    "bltvqg_bn_relu_maxpool": (I, [I, P, P, P, P, I, I, I, I, P]),
    "bltvqg_avgpool": (I, [I, P, P, I, I, I, P]),
    "bltvqg_bn1d_fwd": (I, [I, P, P, P, P, P, P, P, P, I, I, F, F, P]),
    "bltvqg_bn1d_bwd": (I, [I, P, P, P, P, P, P, P, P, I, I, P]),
    "bltvqg_attn_fwd": (I, [I, P, I, P, I, P, I, P, I, P, I, I, I, I, I, I, F, F, U64, U32, P]),
    "bltvqg_attn_fwd_rows": (I, [I, P, I, I, P, I, P, I, I, P, I, P, I, I, I, I, I, I, F, P]),
    "bltvqg_attn_bwd": (I, [I, P, I, P, I, P, I, P, I, P, I, P, I, P, I, P, I, I, I, I, I, I, F, F, U64, U32, P]),
    "bltvqg_embed_gather": (I, [I, P, P, P, L, I, I, P]),
    "bltvqg_embed_scatter": (I, [I, P, I, P, P, L, I, I, P]),
    "bltvqg_ce_fwd_bwd": (I, [I, P, I, P, L, I, P, F, P, I, P]),
    "bltvqg_bow_ce_fwd_bwd": (I, [I, P, I, P, I, I, I, P, F, P, P, P]),
    "bltvqg_mse_fwd_bwd": (I, [I, P, P, L, F, P, P, P, P]),
    "bltvqg_latent_fwd": (I, [I, P, P, P, P, P, I, I, I, P]),
    "bltvqg_latent_bwd": (I, [I, P, P, P, P, F, P, P, I, I, I, P]),
    "bltvqg_sumsq": (I, [P, L, P, P]),
    "bltvqg_adam_step": (I, [P, P, P, P, L, P, F, F, F, F, F, I, P]),
    "bltvqg_dropout_mask": (I, [U64, U32, L, I, I, F, P, P]),
    "bltvqg_cast": (I, [I, P, I, I, P, I, L, I, P]),
    "bltvqg_image_store_u8": (I, [P, P, L, P]),
    "bltvqg_batch_rows": (I, [P, P, P, P, I, L, P, I, I, I, P, P, P, P, P, P]),
    "bltvqg_batch_images": (I, [P, L, I, P, L, P, P, P, I, I, I, ctypes.POINTER(ctypes.c_float), P, P, P]),
    "bltvqg_batch_images_packed": (I, [P, L, I, P, L, P, P, P, I, I, I, ctypes.POINTER(ctypes.c_float), I, P, I, I, I, I, P]),
    "bltvqg_engine_image_input": (I, [P, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                      ctypes.POINTER(ctypes.c_int)]),
    "bltvqg_engine_prefetch_images": (I, [P, P, P]),
    "bltvqg_engine_prefetch_pending": (I, [P]),
    "bltvqg_engine_set_prefetch_split": (I, [P, I]),
    "bltvqg_engine_set_cu_masks": (I, [P, P, P, P, I, I]),
    "bltvqg_engine_chain_stream": (I, [P, ctypes.POINTER(ctypes.c_void_p)]),
    "bltvqg_engine_conv_stream": (I, [P, ctypes.POINTER(ctypes.c_void_p)]),
    "bltvqg_engine_conv_stream_wait": (I, [P, P]),
    "bltvqg_engine_side_stream": (I, [P, I, ctypes.POINTER(ctypes.c_void_p)]),
    "bltvqg_engine_adopt_conv_stream": (I, [P, P]),
    "bltvqg_engine_create": (P, [ctypes.POINTER(Config)]),
    "bltvqg_engine_destroy": (None, [P]),
    "bltvqg_engine_num_params": (I, [P, I]),
    "bltvqg_engine_param_info": (I, [P, I, I, ctypes.c_char_p, I, ctypes.POINTER(L), ctypes.POINTER(L),
                                     ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "bltvqg_engine_flat_size": (L, [P, I]),
    "bltvqg_engine_late_offset": (L, [P]),
    "bltvqg_engine_workspace_bytes": (L, [P]),
    "bltvqg_engine_bind": (I, [P, P, P, P, P, P, P, L]),
    "bltvqg_engine_invalidate_frozen": (None, [P]),
    "bltvqg_engine_trust_shadows": (I, [P, I]),
    "bltvqg_engine_invalidate_params": (None, [P]),
    "bltvqg_engine_forward": (I, [P, P, P, P, P, P, I, U64, P]),
    "bltvqg_engine_decode_greedy": (I, [P, P, P, P, I, I, P, P, P, P]),
    "bltvqg_engine_set_bn_train": (I, [P, I]),
    "bltvqg_engine_loss_backward": (I, [P, F, P]),
    "bltvqg_engine_backward_external": (I, [P, P, P, F, P, P, P]),
    "bltvqg_engine_optimizer_step": (I, [P, F, F, F, F, F, P]),
    "bltvqg_engine_optimizer_step_async": (I, [P, F, F, F, F, F, P]),
    "bltvqg_engine_optimizer_wait": (I, [P, P]),
    "bltvqg_engine_adam_steps": (I, [P, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_int32)]),
    "bltvqg_engine_set_adam_steps": (I, [P, ctypes.c_int32, ctypes.c_int32]),
    "bltvqg_engine_share_optimizer_state": (I, [P, P]),
    "bltvqg_engine_read": (I, [P, I, P, P]),
    "bltvqg_engine_dropout_stream_id": (U32, [I, I, I]),
    "bltvqg_engine_profile_enable": (I, [P, I]),
    "bltvqg_engine_profile_read": (I, [P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)]),
    "bltvqg_engine_profile_read_class": (I, [P, I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double)]),
    "bltvqg_engine_profile_read_streams": (I, [P, I, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]),
    "bltvqg_engine_phase_stamps": (I, [P, ctypes.POINTER(ctypes.c_float)]),
    "bltvqg_engine_set_bucket_flush": (I, [P, I]),
    "bltvqg_engine_num_buckets": (I, [P]),
    "bltvqg_engine_bucket_info": (I, [P, I, ctypes.POINTER(L), ctypes.POINTER(L), ctypes.POINTER(ctypes.c_int32)]),
    "bltvqg_engine_bucket_wait": (I, [P, I, P]),
}

# include/bltvqg_hip_experiments.h: entry points that exist only in the experiments build (make -C blt-vqg_amd/csrc experiments ->
# libbltvqg_hip_exp.so, -DBLT_EXPERIMENTS).  Never loaded by the product; tests load it NEXT TO the product library (load_experiments()).
EXPERIMENT_SIGNATURES = {
    "bltvqg_gemm_repeat": (I, [I, P, I, P, I, P, I, I, I, I, P, I, P, I, I, P]),
    "bltvqg_gemm_rotate": (I, [I, P, I, I, L, P, I, I, L, P, I, I, L, I, I, I, I, I, P]),
    "bltvqg_layernorm_linear": (I, [P, I, P, P, F, P, P, P, P, I, P, I, F, U64, U32, P, I, P, I, I, I, I, P]),
    "bltvqg_linear_layernorm": (I, [P, I, P, I, P, I, F, U64, U32, P, I, P, I, P, I, P, P, F, P, P, P, I, I, I, P]),
    "bltvqg_attn_out_fwd": (I, [P, I, P, I, P, I, P, I, P, I, P, I, P, I, P, I, I, I, I, I, I, F, F, U64, U32, P]),
    "bltvqg_hw_id_probe": (I, [P, I, I, P]),
    "bltvqg_linear_pair": (I, [P, P, I, I, I, F, U64, F, I, I, F, I, I, P]),
}
EXP_LIB_PATH = os.environ.get("BLTVQG_EXP_LIB") or os.path.join(_HERE, "libbltvqg_hip_exp.so")
_exp_lib = None


def load_experiments():
    """The experiments build as a SECOND library in this process (its own copy of the kernels and of the process-wide switches), or None
    when it has not been built.  It carries every product entry point too, so an operator test can run wholly inside it."""
    global _exp_lib
    if _exp_lib is not None:
        return _exp_lib
    if not os.path.exists(EXP_LIB_PATH):
        return None
    import torch  # noqa: F401  (one HIP runtime per process: torch's, see load())
    lib = ctypes.CDLL(EXP_LIB_PATH)
    for table in (SIGNATURES, EXPERIMENT_SIGNATURES):
        for name, (res, args) in table.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
    _exp_lib = lib
    return lib


_lib = None


class HipError(RuntimeError):
    pass


def load():
    """Loads the shared library (once).  Raises if it has not been built — there is no CPU fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # torch first: PyTorch-ROCm ships its own libamdhip64.so; a process must end up with ONE HIP runtime, the one that owns torch's device
    # memory.  Loaded before torch, this library would bind the system runtime (/opt/rocm/lib) and every device pointer torch hands over
    # would be foreign to it ("no ROCm-capable device is detected" on the first hipMemset).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "%s is missing: build it with `make -C blt-vqg_amd/csrc` (or __graft_entry__.build()). "
            "The BLT-VQG hot path has no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().bltvqg_last_error_string()
        raise HipError("%s failed (%d): %s" % (what or "libbltvqg_hip call", rc, (msg or b"").decode("utf-8", "replace")))


def ptr(t):
    """Device (or host) pointer of a tensor, None -> NULL."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
