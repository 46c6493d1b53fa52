"""ctypes binding of libllada_hip.so (the C ABI declared in include/llada_hip.h).

The library is the product: there is no Python or CPU fallback.  ``load()`` raises if the shared
object has not been built (``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C llada_amd/csrc``).
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LLADA_HIP_LIB") or os.path.join(_HERE, "libllada_hip.so")   # override: A/B timing of two builds

ABI_VERSION = 4     # LLADA_ABI_VERSION of include/llada_hip.h this binding was written for
OK = 0
ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_WORKSPACE = -1, -2, -3, -4

REMASK = {"low_confidence": 0, "high_confidence": 1, "low_entropy": 2, "high_entropy": 3, "random": 4}
FLAG_LOGITS_EOS_INF, FLAG_CONF_EOS_EOT_INF, FLAG_CONF_FLOAT32 = 1, 2, 4
EPI_PLAIN, EPI_RESIDUAL, EPI_ROPE_QKV, EPI_SWIGLU = 0, 1, 2, 3
GEMM_W_TILED = 0x100
STATIC_REMASK_MAX_ROWS = 8192

vp = C.c_void_p


class Config(C.Structure):
    _fields_ = [("d_model", C.c_int32), ("n_layers", C.c_int32), ("n_heads", C.c_int32),
                ("mlp_hidden", C.c_int32), ("vocab_size", C.c_int32), ("rms_eps", C.c_float)]


class Weights(C.Structure):
    _fields_ = [("packed", C.c_int32), ("wte", vp), ("ln_f", vp), ("lm_head", vp),
                ("attn_norm", C.POINTER(vp)), ("wqkv", C.POINTER(vp)), ("wo", C.POINTER(vp)),
                ("ff_norm", C.POINTER(vp)), ("wgu", C.POINTER(vp)), ("wdown", C.POINTER(vp))]


class RemaskArgs(C.Structure):
    _fields_ = [("logits", vp), ("uncond_logits", vp), ("cfg_scale", C.c_double), ("row_pos", vp),
                ("n_rows", C.c_int32), ("vocab", C.c_int32), ("x", vp), ("B", C.c_int32), ("L", C.c_int32),
                ("block_end", C.c_int32), ("mask_id", C.c_int64), ("remasking", C.c_int32), ("flags", C.c_uint32),
                ("eos_id", C.c_int32), ("eot_id", C.c_int32), ("temperature", C.c_float), ("noise", vp),
                ("rand_scores", vp), ("seed", C.c_uint64), ("step", C.c_uint32), ("row_offset", C.c_uint32),
                ("x0", vp), ("confidence", vp), ("workspace", vp), ("workspace_bytes", C.c_size_t)]


# name -> (restype, argtypes); every symbol include/llada_hip.h declares
SIGNATURES = {
    "llada_abi_version": (C.c_int, []),
    "llada_last_error": (C.c_char_p, []),
    "llada_forward_workspace_bytes": (C.c_size_t, [C.POINTER(Config), C.c_int32, C.c_int32, C.c_int32]),
    "llada_forward": (C.c_int, [C.POINTER(Config), C.POINTER(Weights), vp, vp, vp, vp, C.c_int32, C.c_int32, vp,
                                C.c_int32, vp, vp, C.c_size_t, vp]),
    "llada_forward_positions": (C.c_int, [C.POINTER(Config), C.POINTER(Weights), vp, vp, vp, vp, vp, C.c_int32, C.c_int32, vp,
                                          C.c_int32, vp, vp, C.c_size_t, vp]),
    "llada_embed_bf16": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int32, C.c_int64, vp]),
    "llada_gather_rows_bf16": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int32, C.c_int64, vp]),
    "llada_rmsnorm_bf16": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int32, C.c_float, vp, vp]),
    "llada_gemm_residual_rmsnorm_bf16": (C.c_int, [vp, vp, vp, vp, vp, C.c_float, C.c_int64, C.c_int32, C.c_int32,
                                                   C.c_int32, C.c_int32, vp, C.c_size_t, vp]),
    "llada_gemm_bf16": (C.c_int, [vp, vp, vp, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp,
                                  C.c_int32, vp, vp, C.c_int32, vp, C.c_size_t, vp]),
    "llada_gemm_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "llada_rope_bf16": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]),
    "llada_swiglu_bf16": (C.c_int, [vp, vp, C.c_int64, C.c_int32, vp]),
    "llada_attn_bidir_bf16": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp]),
    "llada_attn_bidir_queries_bf16": (C.c_int, [vp, C.c_int64, C.c_int32, vp, vp, vp, C.c_int32, C.c_int32, C.c_int32, vp,
                                                C.c_int32, vp, C.c_size_t, vp]),
    "llada_attn_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "llada_kv_cache_bytes": (C.c_size_t, [C.POINTER(Config), C.c_int32, C.c_int32]),
    "llada_forward_fill_cache": (C.c_int, [C.POINTER(Config), C.POINTER(Weights), vp, vp, vp, vp, C.c_int32, C.c_int32, vp,
                                           C.c_int32, vp, vp, C.c_size_t, vp, vp]),
    "llada_forward_block_workspace_bytes": (C.c_size_t, [C.POINTER(Config), C.c_int32, C.c_int32, C.c_int32]),
    "llada_forward_block": (C.c_int, [C.POINTER(Config), C.POINTER(Weights), vp, vp, vp, vp, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, vp, C.c_int32, vp, C.c_int32, vp, vp, C.c_size_t, vp]),
    "llada_num_transfer_tokens": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                            vp, vp]),
    "llada_add_gumbel_noise": (C.c_int, [vp, vp, C.c_double, vp, C.c_int64, C.c_int32, C.c_int32, C.c_uint64, C.c_uint32,
                                         C.c_uint32, vp]),
    "llada_rows_workspace_bytes": (C.c_size_t, [C.c_int32]),
    "llada_remask_scores": (C.c_int, [C.POINTER(RemaskArgs), vp]),
    "llada_remask_scores_static": (C.c_int, [C.POINTER(RemaskArgs), vp]),
    "llada_topk_transfer": (C.c_int, [vp, vp, vp, C.c_int32, C.c_int32, vp, C.c_int32, C.c_int32, vp]),
    "llada_ll_mask": (C.c_int, [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, vp, vp, C.c_uint64, C.c_uint32,
                                C.c_uint32, vp]),
    "llada_ll_masked_rows": (C.c_int, [vp, C.c_int32, C.c_int32, C.c_int32, C.c_int64, vp, vp, vp]),
    "llada_masked_ce": (C.c_int, [vp, vp, C.c_double, vp, C.c_int32, C.c_int32, vp, vp, C.c_int32, C.c_int32,
                                  C.c_int64, vp, C.c_int32, vp, vp, vp, C.c_size_t, vp]),
}
# test hooks outside the public header
_EXTRA = {"llada_rows_force_splits": (None, [C.c_int]), "llada_gemm_force_tile_rows": (None, [C.c_int]), "llada_gemm_use_pipelined": (None, [C.c_int]),
          "llada_gemm_force_splits": (None, [C.c_int]), "llada_gemm_force_tile_cols": (None, [C.c_int]), "llada_gemm_force_small_m": (None, [C.c_int]), "llada_gemm_stream_grid": (None, [C.c_int]), "llada_attn_force_variant": (None, [C.c_int]), "llada_attn_force_key_splits": (None, [C.c_int]), "llada_gemm_assume_tiled": (None, [C.c_int]), "llada_gemm_tile_walk": (None, [C.c_int, C.c_int]), "llada_gemm_last_plan": (None, [C.POINTER(C.c_int32)]), "llada_gemm_describe_plan": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_size_t, C.POINTER(C.c_int32)]), "llada_rmsnorm_use_wave_kernel": (None, [C.c_int]),
          "llada_gemm_profile_begin": (C.c_int, [C.c_int, C.c_int, C.c_longlong]),
          "llada_gemm_profile_end": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int)])}

_lib = None


class LLaDAHipError(RuntimeError):
    pass


def build(verbose=False):
    """Compile the HIP sources for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout[-4000:])
        print(res.stderr[-4000:])
    if res.returncode != 0:
        raise LLaDAHipError("building libllada_hip.so failed")
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LLaDAHipError(f"{LIB_PATH} is missing: build it first (make -C llada_amd/csrc); there is no fallback path")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in {**SIGNATURES, **_EXTRA}.items():
        if name in _EXTRA and not hasattr(lib, name):
            continue            # a diagnostic hook an older build (LLADA_HIP_LIB, A/B timing) does not have
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.llada_abi_version() != ABI_VERSION:
        raise LLaDAHipError("libllada_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    """Turn a C status into the exception the reference would raise for the same mistake."""
    if rc == OK:
        return
    msg = load().llada_last_error().decode()
    if rc == ERR_INVALID:
        raise AssertionError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise LLaDAHipError(msg)


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
