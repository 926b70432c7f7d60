"""ctypes binding of libantmmf_hip.so (C ABI: include/antmmf_hip.h).

The library is the product: if it cannot be loaded this module raises -- there is no eager / CPU
fallback for the hot path.  `ANTMMF_HIP_LIB` may point at another build of the same ABI (the unit
tests use it to run the kernels' CPU lane emulation, tests/emu, whose `antmmf_backend()` is 0 and
which only accepts host tensors).
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libantmmf_hip.so"))

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU_ERF, ACT_QUICK_GELU, ACT_RELU = 0, 1, 2, 3
ACT_IDS = {None: 0, "none": 0, "gelu": 1, "gelu_erf": 1, "quick_gelu": 2, "relu": 3}

P, I, L, F, U64 = c_void_p, c_int, c_int64, c_float, c_uint64
_SIGNATURES = {
    "antmmf_backend": [],
    "antmmf_abi_version": [],
    "antmmf_layernorm_fwd": [P, P, P, P, P, P, L, I, F, I, P],
    "antmmf_layernorm_bwd": [P, P, P, P, P, P, P, P, P, L, I, I, P],
    "antmmf_act_layernorm_fwd": [P, P, P, P, P, P, L, I, F, I, I, P],
    "antmmf_act_layernorm_bwd": [P, P, P, P, P, P, P, P, P, P, L, I, I, I, P, L, P],
    "antmmf_layernorm_bwd_renorm": [P, P, P, P, P, P, P, P, P, P, P, P, L, I, I, P],
    "antmmf_act_fwd": [P, P, L, I, I, P],
    "antmmf_act_bwd": [P, P, P, L, I, I, P],
    "antmmf_l2norm_fwd": [P, P, P, L, I, F, I, I, P],
    "antmmf_l2norm_bwd": [P, P, P, P, L, I, I, I, P],
    "antmmf_colsum": [P, P, L, I, L, I, P],
    "antmmf_transpose_bf16": [P, P, I, I, P],
    "antmmf_transpose_bf16_batched": [P, P, P, I, L, P],
    "antmmf_cast_f32_bf16": [P, P, L, P],
    "antmmf_split_hi_lo_bf16": [P, L, I, I, P, P, I, I, P],
    "antmmf_patchify": [P, P, I, I, I, I, I, I, F, F, I, P],
    "antmmf_assemble_tokens": [P, P, P, P, P, L, I, I, P],
    "antmmf_split_tokens": [P, P, L, I, I, P],
    "antmmf_embed_gather": [P, P, P, P, P, P, P, L, I, I, I, P],
    "antmmf_embed_scatter_add": [P, P, P, P, L, I, I, I, P],
    "antmmf_embed_scatter_add_sorted": [P, P, P, P, L, L, I, P],
    "antmmf_adamw_step": [P, P, P, P, P, L, F, F, F, F, F, I, F, P],
    "antmmf_adamw_step_scaled": [P, P, P, P, P, L, F, F, F, F, F, I, F, P, P],
    "antmmf_sumsq": [P, P, L, P],
    "antmmf_gemm_bf16": [P, P, P, I, I, I, L, L, L, I, I, I, F, P, I, P, L, P, L, P, L, I, I, P],
    "antmmf_gemm_bf16_ws": [P, P, P, I, I, I, L, L, L, I, I, I, F, P, I, P, L, P, L, P, L, I, I, P, L, P],
    "antmmf_gemm_wgrad_bf16": [P, P, P, L, I, I, L, L, L, I, P, L, P],
    "antmmf_gemm_wgrad_bf16_seg": [P, P, P, I, I, L, I, L, L, L, I, P, L, P],
    "antmmf_attention_fwd": [P, P, P, P, P, P, I, I, I, I, L, L, L, L, F, F, U64, P],
    "antmmf_attention_bwd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, L, L, L, L, L, L, L, L, F, F, U64, P],
    "antmmf_attention_fwd_hd": [P, P, P, P, P, P, I, I, I, I, I, L, L, L, L, F, F, U64, P],
    "antmmf_attention_key_importance": [P, P, P, P, P, I, I, I, I, L, L, F, F, U64, F, P],
    "antmmf_attention_bwd_hd": [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, L, L, L, L, L, L, L, L, F, F, U64, P],
    "antmmf_gemm_bf16_gated_colsum_ok": [I, I, I, L, L],
    "antmmf_gemm_bf16_gated_colsum": [P, P, P, I, I, I, L, L, L, P, L, P, P],
    "antmmf_attention_bwd_sums_ok": [I, I, I, F],
    "antmmf_attention_bwd_sums": [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, L, L, L, L, L, L, L, L, F, P],
    "antmmf_milnce_fwd": [P, P, I, I, I, I, I, P, P, P],
    "antmmf_milnce_bwd": [P, P, P, P, I, I, I, I, I, P, P, I, P],
    "antmmf_softmax_ce_fwd": [P, I, I, I, P, F, P, P, P],
    "antmmf_softmax_ce_bwd": [P, P, P, I, I, I, P, F, P, P, I, P],
    "antmmf_moco_fwd": [P, P, I, I, I, F, P, P, P, P],
    "antmmf_moco_bwd": [P, P, P, P, P, I, I, I, F, P, P, I, P],
    "antmmf_ema_update": [P, P, P, L, F, P],
    "antmmf_dropout_add": [P, P, P, L, F, U64, I, P],
    "antmmf_wti_reduce_fwd": [P, I, I, I, I, P, P, P, P, P, P, P, P, P],
    "antmmf_wti_reduce_bwd": [P, I, I, I, I, P, P, P, P, P, P, P, P, P, P, I, P],
    "antmmf_rank_rows": [P, L, I, I, P, P, P, P],
    "antmmf_token_weight_fwd": [P, P, P, P, P, I, I, I, P],
    "antmmf_token_weight_bwd": [P, P, P, P, P, P, P, P, L, I, I, I, P],
    "antmmf_pair_dots": [P, P, P, I, I, I, P],
    "antmmf_pair_wsum": [P, P, P, I, I, I, P],
    "antmmf_pair_outer": [P, P, P, I, I, I, P],
    "antmmf_tis_keep": [P, F, P, I, I, P],
    "antmmf_negnce_fwd": [P, P, I, I, I, F, F, P, P, P, P, P],
    "antmmf_negnce_bwd": [P, P, P, P, I, I, I, F, F, P, I, P],
    "antmmf_resize_bicubic_u8": [P, L, P, I, I, I, I, I, I, P, P, P, P, I, P],
    "antmmf_frames_bilinear_norm": [P, I, I, I, I, L, L, L, L, P, I, I, L, L, L, P, P, I, P, P],
    "antmmf_frames_bilinear_aa_norm": [P, I, I, I, I, L, L, L, L, P, P, I, I, L, L, L, P, P, I, P, P],
}

# exported by the LAB library only (libantmmf_hip_lab.so, include/antmmf_hip_lab.h; also by the test-side emulator): the sub-LN fold and the A/B switch.
# Bound when present; the product library has none of them.
_LAB_SIGNATURES = {
    "antmmf_ffn_prepare_w2": [P, P, P, P, P, P, P, I, I, P],
    "antmmf_ffn_fc1_fwd": [P, P, P, P, P, P, I, I, I, L, L, L, I, F, P, L, P],
    "antmmf_ffn_fc2_fwd": [P, P, P, P, P, P, P, I, I, I, L, L, L, L, P],
    "antmmf_ffn_bwd_rows": [P, P, P, P, P, P, P, P, P, P, I, I, I, L, L, L, L, P],
    "antmmf_ffn_fc2_dgrad": [P, P, P, P, P, P, P, I, I, I, L, L, L, L, P, L, P],
    "antmmf_ffn_wgrad_post": [P, P, P, P, P, P, P, P, P, I, I, P],
    "antmmf_debug_set_gemm_variant": [I],
}
_LAB_PROBES = ("antmmf_debug_gemm_cell_launches", "antmmf_debug_attn_fused_launches")   # read-only counters of the lab build (restype long; bound by the tests that read them)
LAB_LIB = os.path.join(os.path.dirname(DEFAULT_LIB), "libantmmf_hip_lab.so")


class HipLibraryError(RuntimeError):
    pass


_lib = None
_backend = None


def lib_path():
    return os.environ.get("ANTMMF_HIP_LIB") or DEFAULT_LIB


def load():
    """Load (once) and return the ctypes handle; raises HipLibraryError if the extension is missing."""
    global _lib, _backend
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.isfile(path):
        raise HipLibraryError(
            f"libantmmf_hip.so not found at {path}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"(or `make -C ant-multi-modal-framework_amd/csrc`). The HIP path has no fallback.")
    try:
        lib = ctypes.CDLL(path)
    except OSError as e:  # pragma: no cover
        raise HipLibraryError(f"cannot load {path}: {e}") from e
    for name, args in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{path} does not export {name}") from e
        fn.argtypes = args
        fn.restype = c_int
    lab = 0
    for name, args in _LAB_SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes = args
            fn.restype = c_int
            lab += 1
    lib.antmmf_is_lab = lab == len(_LAB_SIGNATURES)
    be = lib.antmmf_backend()
    if be == 0 and not (os.environ.get("PYTEST_CURRENT_TEST") or os.environ.get("ANTMMF_ALLOW_EMULATOR")):
        # the CPU lane emulator is test infrastructure: the product path never runs on it (ANTMMF_HIP_LIB could point it there)
        raise HipLibraryError(f"{path} is the CPU lane EMULATOR build of the kernels (tests/emu); it is only accepted under pytest "
                              f"(or with ANTMMF_ALLOW_EMULATOR=1 for debugging). The product path needs the gfx950 library.")
    _lib = lib
    _backend = be
    return lib


def backend():
    """1 = gfx950 device library, 0 = CPU lane emulator."""
    load()
    return _backend


def is_lab():
    """True when the loaded library is the measurement build (A/B switches, sub-LN fold): never on the product path."""
    return bool(load().antmmf_is_lab)


def reset_for_tests():
    global _lib, _backend
    _lib = None
    _backend = None
