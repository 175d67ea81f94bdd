"""ctypes binding of libsdlt_kernels.so (include/sdlt_kernels.h).

The product path has NO fallback: if the HIP library is missing or a kernel returns an error this
module raises.  torch is imported first so the library binds to the libamdhip64 that PyTorch-ROCm
already loaded (one HIP runtime per process -> shared streams and graph capture).
"""
import ctypes as C
import os

import torch  # noqa: F401  (must be loaded before the HIP library, see above)

_HERE = os.path.dirname(os.path.abspath(__file__))
# SDLT_KERNEL_LIB: load an alternative build of the SAME C-ABI (kernel A/B experiments); never a fallback
LIB_PATH = os.environ.get("SDLT_KERNEL_LIB") or os.path.join(_HERE, "libsdlt_kernels.so")

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmParams(C.Structure):
    _fields_ = [
        ("X", vp), ("ldx", i64), ("W", vp), ("ldw", i64),
        ("M", i32), ("N", i32), ("K", i32),
        ("X2", vp), ("ldx2", i64), ("W2", vp), ("ldw2", i64),
        ("K2", i32), ("mode", i32),
        ("Hin", i32), ("Win", i32), ("Cin", i32), ("Hout", i32), ("Wout", i32),
        ("stride", i32), ("ups", i32), ("flip", i32), ("tr", i32),
        ("zero", vp),
        ("Adown", vp), ("ld_adown", i64), ("Bup", vp), ("ld_bup", i64),
        ("T_out", vp), ("ld_t", i64),
        ("lora_R", i32), ("lora_scale", f32), ("alpha", f32),
        ("bias", vp),
        ("rowbias", vp), ("ld_rowbias", i64), ("rows_per_batch", i32),
        ("R", vp), ("ldr", i64),
        ("C", vp), ("ldc", i64),
        ("out_fp32", i32), ("tile", i32),
        ("Ct", vp), ("ldct", i64),
        ("splitk", i32), ("ws_cnt_len", i32), ("ws_slab", vp), ("ws_slab_bytes", i64), ("stages", i32), ("accumulate", i32), ("ws_cnt", vp),
        ("lora_group_n", i32), ("lora_group_k", i32),
        ("batch", vp), ("n_batch", i32), ("throughput_hint", i32),
        ("epi_op", i32), ("epi_act", i32), ("epi_out", vp), ("ld_epi_out", i64), ("epi_in", vp), ("ld_epi_in", i64),
        ("col_scale", vp),
        ("ln_c1", vp), ("ln_stats", vp), ("ln_adapter", vp), ("ln_eps", f32), ("ln_nparts", i32), ("ln_parts", vp),
    ]


class LnFoldDesc(C.Structure):
    _fields_ = [("A32", vp), ("lda", i64), ("gamma", vp), ("beta", vp), ("Ag", vp), ("ldag", i64), ("consts", vp), ("rank", i32), ("K", i32)]


class GemmBatchItem(C.Structure):
    _fields_ = [("X", vp), ("W", vp), ("Adown", vp), ("Bup", vp), ("T_out", vp), ("C", vp), ("Ct", vp), ("bias", vp), ("col_scale", vp)]


class SplitsumDesc(C.Structure):
    _fields_ = [("s0", vp), ("s1", vp), ("ld32", i64), ("out0", vp), ("ldo0", i64), ("out1", vp), ("ldo1", i64),
                ("nsplit", i32), ("B", i32), ("Nk", i32), ("Nkp", i32), ("C", i32), ("acc0", i32), ("nblocks", i32), ("pad_", i32)]


class DoraDesc(C.Structure):
    _fields_ = [("W", vp), ("ldw", i64), ("A", vp), ("lda", i64), ("B", vp), ("ldb", i64), ("mag", vp), ("scale", vp), ("Bt", vp), ("ldbt", i64),
                ("B32", vp), ("ldb32", i64), ("N", i32), ("K", i32), ("Rp", i32), ("rank", i32), ("s", f32), ("pad_", i32)]


class DoraWtDesc(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("ld", i64), ("scale", vp), ("rows", i32), ("cols", i32), ("period", i32), ("nvalid", i32)]


class DoraGradDesc(C.Structure):
    _fields_ = [("dY", vp), ("lddy", i64), ("Y", vp), ("ldy", i64), ("bias", vp), ("mag", vp), ("scale", vp), ("gmag", vp), ("gB", vp),
                ("ws_off", i64), ("M", i32), ("N", i32), ("rank", i32), ("splits", i32), ("grad_scale", f32), ("accumulate", i32)]


class LoraGradDesc(C.Structure):
    _fields_ = [
        ("P", vp), ("ldp", i64), ("Q", vp), ("ldq", i64), ("out", vp),
        ("M", i32), ("Cw", i32), ("R", i32), ("Rp", i32),
        ("rank_major", i32), ("accumulate", i32),
        ("conv", i32), ("Hin", i32), ("Win", i32), ("Cin", i32), ("Hout", i32), ("Wout", i32), ("stride", i32),
        ("zero", vp),
        ("first_block", i32), ("pad_", i32),
    ]


class AttnParams(C.Structure):
    _fields_ = [
        ("Q", vp), ("ldq", i64), ("K", vp), ("ldk", i64), ("V", vp), ("ldv", i64),
        ("Kt", vp), ("ldkt", i64), ("Vt", vp), ("ldvt", i64), ("Qt", vp), ("ldqt", i64), ("dOt", vp), ("lddot", i64),
        ("O", vp), ("ldo", i64), ("L", vp),
        ("dO", vp), ("lddo", i64), ("D", vp),
        ("dQ", vp), ("lddq", i64), ("dK", vp), ("lddk", i64), ("dV", vp), ("lddv", i64),
        ("dK32", vp), ("dV32", vp), ("ld32", i64),
        ("B", i32), ("H", i32), ("Nq", i32), ("Nk", i32), ("Nqp", i32), ("Nkp", i32), ("d", i32),
        ("scale", f32), ("qsplit", i32), ("causal", i32), ("accumulate_dq", i32), ("accumulate_dk", i32), ("defer_splitsum", i32), ("d_ready", i32),
    ]


class GroupNormParams(C.Structure):
    _fields_ = [
        ("x1", vp), ("ldx1", i64), ("C1", i32),
        ("x2", vp), ("ldx2", i64),
        ("B", i32), ("HW", i32), ("C", i32),
        ("gamma", vp), ("beta", vp), ("eps", f32),
        ("silu", i32),
        ("y", vp), ("ldy", i64),
        ("stats", vp),
        ("dy", vp), ("lddy", i64),
        ("dres", vp), ("lddres", i64),
        ("dx", vp), ("lddx", i64),
        ("bstats", vp),
        ("ws", vp), ("ws_floats", i64),
        ("cnt", vp), ("cnt_len", i32), ("pad_", i32),
        ("colsum_ws", vp),
    ]


class ColsumFinishDesc(C.Structure):
    _fields_ = [("ws", vp), ("out32", vp), ("out16", vp), ("nsplit", i32), ("n", i32)]


class WskGemmParams(C.Structure):
    _fields_ = [("X", vp), ("ldx", i64), ("W", vp), ("ldw", i64), ("bias", vp), ("R", vp), ("ldr", i64), ("Y", vp), ("ldy", i64),
                ("Adown", vp), ("ld_adown", i64), ("Bup", vp), ("ld_bup", i64), ("T_out", vp), ("ld_t", i64), ("col_scale", vp), ("Y0", vp), ("ldy0", i64),
                ("ln_c1", vp), ("ln_stats", vp), ("ln_adapter", vp), ("ln_parts", vp), ("dotD", vp), ("pf_next_w", vp),
                ("M", i32), ("N", i32), ("K", i32), ("lora_group_k", i32), ("lora_rp", i32), ("dot_nq", i32), ("lora_scale", f32), ("ln_eps", f32),
                ("pf_next_n", i32), ("pf_next_k", i32), ("pf_steps", i32), ("pad_", i32)]


class StripParams(C.Structure):
    _fields_ = [("X", vp), ("ldx", i64), ("W", vp), ("ldw", i64), ("bias", vp), ("R", vp), ("ldr", i64), ("Y", vp), ("ldy", i64),
                ("Y2", vp), ("ldy2", i64), ("Z", vp), ("ldz", i64), ("c1", vp), ("c2", vp), ("stats", vp),
                ("B", i32), ("T", i32), ("Tp", i32), ("N", i32), ("K", i32), ("act", i32), ("ln", i32), ("eps", f32),
                ("splitk", i32), ("cnt_len", i32), ("ws", vp), ("ws_bytes", i64), ("cnt", vp), ("P", vp), ("ldp", i64)]


class TaGroup(C.Structure):
    _fields_ = [("S", vp), ("Wh", vp), ("Ww", vp), ("ch", vp), ("cw", vp), ("dS", vp), ("dSt", vp), ("dheat", vp), ("h", i32), ("w", i32)]


class LnSlabsParams(C.Structure):
    _fields_ = [("x", vp), ("ldx", i64), ("dy32", vp), ("lddy32", i64), ("gamma", vp), ("stats", vp), ("dres", vp), ("lddres", i64),
                ("dx", vp), ("lddx", i64), ("nslab", i32), ("M", i32), ("C", i32), ("pad_", i32)]


class TaParams(C.Structure):
    _fields_ = [("g", TaGroup * 4), ("mask", vp), ("tok_w", vp), ("tok_cnt", vp), ("ti_onehot", vp), ("has_ti", vp), ("ws", vp), ("ws_floats", i64),
                ("loss", vp), ("ngroups", i32), ("B", i32), ("n_tok", i32), ("n_layers", i32), ("mH", i32), ("mW", i32), ("weight", f32),
                ("max_px", i32), ("max_tmp", i32), ("max_w", i32), ("pad_", i32)]


class ShadowDesc(C.Structure):
    _fields_ = [("offset", i64), ("src_ld", i64), ("rows", i32), ("cols", i32), ("dst", vp), ("ld", i64), ("dstT", vp), ("ldT", i64)]


# every symbol include/sdlt_kernels.h declares (tests/test_capi_symbols.py cross-checks this list with the header)
SYMBOLS = {
    "sdlt_last_error": (C.c_char_p, []),
    "sdlt_abi_version": (i32, []),
    "sdlt_struct_size": (i32, [i32]),
    "sdlt_add2d": (i32, [vp, i64, vp, i64, vp, i64, i32, i32, vp]),
    "sdlt_gemm_bf16": (i32, [C.POINTER(GemmParams), vp]),
    "sdlt_lora_grad_grouped": (i32, [vp, vp, i32, i32, i32, vp]),
    "sdlt_lora_grad_block_cols": (i32, []),
    "sdlt_attn_fwd": (i32, [C.POINTER(AttnParams), vp]),
    "sdlt_attn_bwd": (i32, [C.POINTER(AttnParams), vp]),
    "sdlt_attn_splitsum_batch": (i32, [vp, vp, vp, i32, vp]),
    "sdlt_groupnorm_ws_floats": (i32, [i32, i32, i32]),
    "sdlt_colsum_finish_batch": (i32, [vp, i32, i32, vp]),
    "sdlt_groupnorm_fwd": (i32, [C.POINTER(GroupNormParams), vp]),
    "sdlt_groupnorm_bwd": (i32, [C.POINTER(GroupNormParams), vp]),
    "sdlt_layernorm_fwd": (i32, [vp, i64, i32, i32, vp, vp, f32, vp, i64, vp, vp]),
    "sdlt_layernorm_bwd": (i32, [vp, i64, vp, i64, i32, i32, vp, vp, vp, i64, vp, i64, vp]),
    "sdlt_layernorm_bwd_y": (i32, [vp, i64, vp, i64, i32, i32, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp]),
    "sdlt_ln_fold_adapters": (i32, [vp, i32, vp]),
    "sdlt_layernorm_bwd_slabs": (i32, [vp, i64, vp, i64, i32, i32, i32, vp, vp, vp, i64, vp, i64, vp]),
    "sdlt_geglu_fwd": (i32, [vp, i64, i32, i32, vp, i64, vp]),
    "sdlt_geglu_bwd": (i32, [vp, i64, vp, i64, i32, i32, vp, i64, vp]),
    "sdlt_map_bf16": (i32, [i32, vp, vp, vp, i64, vp]),
    "sdlt_timestep_embedding": (i32, [vp, i32, i32, vp, i64, vp]),
    "sdlt_add_noise_nhwc": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp]),
    "sdlt_masked_mse_fwd_bwd": (i32, [vp, i64, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, i32, f32, vp, i32, vp, vp, vp]),
    "sdlt_adamw_fused": (i32, [vp, vp, vp, vp, i64, vp, vp, vp]),
    "sdlt_prodigy_step": (i32, [vp, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, vp]),
    "sdlt_wgrad_transpose": (i32, [vp, i64, i32, i32, vp, i64, i32, vp, vp]),
    "sdlt_wgrad_im2col_t": (i32, [vp, i64, i32, i32, i32, i32, i32, i32, vp, i64, i32, vp]),
    "sdlt_wgrad_transpose_batch": (i32, [vp, i32, i64, i32, i32, i64, i32, vp]),
    "sdlt_affine_grad_batch": (i32, [vp, i32, i32, i64, i32, i64, i64, i32, i32, i32, f32, i32, vp]),
    "sdlt_wgrad_im2col_t_batch": (i32, [vp, i32, i64, i32, i32, i32, i32, i32, i32, i64, i32, vp]),
    "sdlt_layernorm_affine_grad": (i32, [vp, i64, vp, i64, i32, i32, vp, vp, vp, i32, vp]),
    "sdlt_groupnorm_affine_grad": (i32, [vp, vp, vp, i32, vp]),
    "sdlt_lora_shadow_refresh": (i32, [vp, vp, vp, i32, vp, vp]),
    "sdlt_adamw_shadow_refresh": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp]),
    "sdlt_adamw8_shadow_refresh": (i32, [vp, vp, vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]),
    "sdlt_adamw8_flat": (i32, [vp, vp, vp, vp, vp, i64, vp, vp, vp]),
    "sdlt_dora_refresh": (i32, [vp, vp, vp, i32, i32, i32, vp]),
    "sdlt_dora_scale_wt": (i32, [vp, vp, vp, i32, vp]),
    "sdlt_dora_mag_grad": (i32, [vp, vp, vp, i32, vp, vp, i32, vp, vp]),
    "sdlt_strip_gemm": (i32, [C.POINTER(StripParams), vp]),
    "sdlt_strip_gemm_pair": (i32, [C.POINTER(StripParams), C.POINTER(StripParams), vp]),
    "sdlt_attn_pair_ok": (i32, [C.POINTER(AttnParams), C.POINTER(AttnParams)]),
    "sdlt_attn_fwd_pair": (i32, [C.POINTER(AttnParams), C.POINTER(AttnParams), vp]),
    "sdlt_attn_bwd_pair": (i32, [C.POINTER(AttnParams), C.POINTER(AttnParams), vp]),
    "sdlt_layernorm_bwd_slabs_pair": (i32, [C.POINTER(LnSlabsParams), C.POINTER(LnSlabsParams), vp]),
    "sdlt_wsk_gemm": (i32, [vp, i64, vp, i64, i32, i32, i32, vp, vp, i64, vp, i64, vp, i64, vp, i64, f32, vp, i64, i32, vp]),
    "sdlt_wsk_gemm_rowdot": (i32, [vp, i64, vp, i64, i32, i32, i32, vp, vp, i64, vp, i64, vp, i64, vp, i64, f32, vp, i64, i32, vp, i32, vp]),
    "sdlt_wsk_pack_weight": (i32, [vp, i64, i32, i32, vp, vp]),
    "sdlt_wsk_conv": (i32, [vp, i64, vp, i64, i32, i32, i32, i32, i32, i32, vp, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, f32, vp, i64, vp, vp]),
    "sdlt_wsk_gemm_parts": (i32, [vp, i64, vp, i64, i32, i32, i32, vp, vp, i64, vp, i64, vp, i64, vp, i64, f32, vp, i64, i32, vp, vp]),
    "sdlt_wsk_gemm_p": (i32, [C.POINTER(WskGemmParams), vp]),
    "sdlt_wsk_gemm_ln": (i32, [vp, i64, vp, i64, i32, i32, i32, vp, vp, i64, vp, i64, vp, i64, vp, i64, f32, vp, i64, vp, vp, f32, vp, vp]),
    "sdlt_token_attention_ws_floats": (i64, [C.POINTER(TaParams)]),
    "sdlt_token_attention_loss": (i32, [C.POINTER(TaParams), vp]),
    "sdlt_sum2x2": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "sdlt_colsum": (i32, [vp, i64, i32, i32, i32, vp, i64, vp, vp, vp]),
    "sdlt_embed_gather": (i32, [vp, i64, vp, vp, i64, i32, i32, i32, i32, vp, i64, vp]),
    "sdlt_embed_grad": (i32, [vp, i64, vp, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "sdlt_ti_std_reg": (i32, [vp, i32, i32, f32, f32, f32, vp, vp, vp]),
}

_lib = None


class KernelLibraryError(RuntimeError):
    pass


def load():
    """Load the HIP library (once).  Raises KernelLibraryError - never falls back to anything."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KernelLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for the training step.")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise KernelLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    for which, (name, size) in enumerate(struct_sizes()):
        if lib.sdlt_struct_size(which) != size:
            raise KernelLibraryError(f"struct layout mismatch for {name}: C {lib.sdlt_struct_size(which)} vs binding {size}")
    _lib = lib
    return lib


def struct_sizes():
    """(name, bytes) of every struct of include/sdlt_kernels.h as this binding lays it out, in sdlt_struct_size()'s order.  The two item tables that
    ops.py packs with `struct` (8 / 3 pointers) are listed by their packed size."""
    mirrored = (GemmParams, LoraGradDesc, AttnParams, GroupNormParams, ShadowDesc, GemmBatchItem, DoraDesc, DoraWtDesc, DoraGradDesc, SplitsumDesc, StripParams,
                TaParams, LnSlabsParams, TaGroup)
    return [(c.__name__, C.sizeof(c)) for c in mirrored] + [("sdlt_affine_grad_item", 8 * 8), ("sdlt_wgrad_tr_item", 3 * 8), ("LnFoldDesc", C.sizeof(LnFoldDesc)), ("ColsumFinishDesc", C.sizeof(ColsumFinishDesc)),
                                                               ("WskGemmParams", C.sizeof(WskGemmParams))]


def check(rc, what):
    if rc != 0:
        msg = load().sdlt_last_error()
        raise KernelLibraryError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
