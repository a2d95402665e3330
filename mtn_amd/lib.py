"""ctypes binding of libmtn_hip.so (C ABI in include/mtn_hip.h).

The HIP library is the product path: if it is missing this module raises — there is no
CPU or PyTorch fallback behind these calls.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmtn_hip.so")

MTN_F32, MTN_BF16 = 0, 1


class Dropout(C.Structure):
    _fields_ = [("p", C.c_float), ("salt", C.c_uint32), ("seed", C.c_void_p)]


class AdamFuse(C.Structure):
    _fields_ = [("p", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("p_lp", C.c_void_p), ("p_lpT", C.c_void_p),
                ("ldT", C.c_int), ("write_grad", C.c_int), ("state", C.c_void_p), ("grad_scale", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float)]


class LnEpilogue(C.Structure):
    _fields_ = [("mode", C.c_int), ("fold", C.c_void_p), ("gate_inv_scale", C.c_float), ("part", C.c_void_p), ("np", C.c_int),
                ("x", C.c_void_p), ("a2", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("dres", C.c_void_p),
                ("eps", C.c_float), ("dx", C.c_void_p), ("dx_lp", C.c_void_p), ("dx_lp_drop", Dropout), ("colpart", C.c_void_p)]


class LnFoldDesc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("bias", C.c_void_p), ("a2", C.c_void_p), ("b2", C.c_void_p), ("out", C.c_void_p),
                ("K", C.c_int), ("block_start", C.c_int)]


class GemmProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("lda", C.c_int), ("ldb", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("a_trans", C.c_int), ("b_trans", C.c_int),
                ("bias", C.c_void_p), ("relu", C.c_int), ("drop", Dropout),
                ("gate", C.c_void_p), ("gate_scale", C.c_float),
                ("residual", C.c_void_p), ("ldr", C.c_int),
                ("out_f32", C.c_void_p), ("out_lp", C.c_void_p), ("ldc", C.c_int), ("lp_drop_after_residual", C.c_int),
                ("rowsum_out", C.c_void_p),
                ("adam", C.POINTER(AdamFuse)), ("ln", C.POINTER(LnEpilogue))]


class AttnArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("h", C.c_int), ("a", C.c_int), ("m", C.c_int), ("dk", C.c_int),
                ("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("ldq", C.c_int), ("ldkv", C.c_int),
                ("mask", C.c_void_p), ("mask_sb", C.c_long), ("mask_sq", C.c_long), ("drop", Dropout),
                ("o", C.c_void_p), ("ldo", C.c_int), ("lse", C.c_void_p),
                ("d_o", C.c_void_p), ("dq", C.c_void_p), ("dk_out", C.c_void_p), ("dv_out", C.c_void_p),
                ("q0", C.c_int), ("qn", C.c_int), ("kv_accum", C.c_int), ("kv_acc", C.c_void_p), ("kv_last", C.c_int)]


class MhaArgs(C.Structure):
    _fields_ = [("B", C.c_int), ("a", C.c_int), ("m", C.c_int), ("d", C.c_int), ("h", C.c_int),
                ("self_attn", C.c_int), ("ln_eps", C.c_float), ("drop_attn", Dropout), ("drop_out", Dropout),
                ("x", C.c_void_p), ("mem", C.c_void_p), ("mask", C.c_void_p), ("mask_sb", C.c_long), ("mask_sq", C.c_long),
                ("ln_a", C.c_void_p), ("ln_b", C.c_void_p), ("w_qkv", C.c_void_p), ("b_qkv", C.c_void_p),
                ("w_o", C.c_void_p), ("b_o", C.c_void_p), ("w_qkv_t", C.c_void_p), ("w_o_t", C.c_void_p),
                ("y", C.c_void_p), ("xn", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("qkv", C.c_void_p), ("kv", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
                ("dy", C.c_void_p), ("dx", C.c_void_p), ("dmem", C.c_void_p), ("dmem_accumulate", C.c_int),
                ("d_ln_a", C.c_void_p), ("d_ln_b", C.c_void_p),
                ("d_w_qkv", C.c_void_p), ("d_b_qkv", C.c_void_p), ("d_w_o", C.c_void_p), ("d_b_o", C.c_void_p),
                ("ws_lp", C.c_void_p), ("ws_f32", C.c_void_p), ("defer_param_grads", C.c_int),
                ("dyl_ready", C.c_void_p), ("next_dyl", C.c_void_p), ("next_drop", Dropout), ("kv_ready", C.c_int),
                ("dmem_lp", C.c_void_p), ("dmem_lp_drop", Dropout), ("ln_fold", C.c_void_p)]


class FfnArgs(C.Structure):
    _fields_ = [("rows", C.c_int), ("d", C.c_int), ("d_ff", C.c_int), ("ln_eps", C.c_float),
                ("drop_hidden", Dropout), ("drop_out", Dropout),
                ("x", C.c_void_p), ("ln_a", C.c_void_p), ("ln_b", C.c_void_p),
                ("w1", C.c_void_p), ("b1", C.c_void_p), ("w2", C.c_void_p), ("b2", C.c_void_p),
                ("w1_t", C.c_void_p), ("w2_t", C.c_void_p),
                ("y", C.c_void_p), ("xn", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("hid", C.c_void_p),
                ("dy", C.c_void_p), ("dx", C.c_void_p),
                ("d_ln_a", C.c_void_p), ("d_ln_b", C.c_void_p), ("d_w1", C.c_void_p), ("d_b1", C.c_void_p),
                ("d_w2", C.c_void_p), ("d_b2", C.c_void_p),
                ("ws_lp", C.c_void_p), ("ws_f32", C.c_void_p), ("defer_param_grads", C.c_int),
                ("dyl_ready", C.c_void_p), ("next_dyl", C.c_void_p), ("next_drop", Dropout), ("y_lp", C.c_void_p),
                ("ln_fold", C.c_void_p)]


class LossHeadArgs(C.Structure):
    _fields_ = [("n_seg", C.c_int), ("rows", C.c_int * 4), ("target", C.c_void_p * 4), ("norm", C.c_void_p * 4), ("coef", C.c_float * 4),
                ("V", C.c_int), ("ldz", C.c_int), ("pad", C.c_int), ("smoothing", C.c_float),
                ("logits", C.c_void_p), ("lse", C.c_void_p), ("rowloss", C.c_void_p),
                ("gloss", C.c_void_p), ("dlogits", C.c_void_p), ("ldd", C.c_int)]


class TransposeDesc(C.Structure):
    _fields_ = [("off", C.c_long), ("rows", C.c_int), ("cols", C.c_int), ("tile_start", C.c_int), ("reserved", C.c_int), ("dst_off", C.c_long)]


class LnFwdDesc(C.Structure):
    _fields_ = [("rows", C.c_int), ("d", C.c_int), ("eps", C.c_float), ("x", C.c_void_p), ("a2", C.c_void_p), ("b2", C.c_void_p),
                ("y_f32", C.c_void_p), ("y_lp", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p),
                ("tokens", C.c_void_p), ("lut", C.c_void_p), ("emb_scale", C.c_float), ("pe", C.c_void_p), ("seq_len", C.c_int),
                ("drop", Dropout), ("x_out", C.c_void_p), ("no_ln", C.c_int)]


class EmbedBwdDesc(C.Structure):
    _fields_ = [("rows", C.c_int), ("d", C.c_int), ("tokens", C.c_void_p), ("dx", C.c_void_p), ("emb_scale", C.c_float),
                ("drop", Dropout), ("dlut", C.c_void_p), ("lut_rows", C.c_int)]


class LnBwdDesc(C.Structure):
    _fields_ = [("rows", C.c_int), ("d", C.c_int), ("eps", C.c_float), ("x", C.c_void_p), ("a2", C.c_void_p), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("g", C.c_void_p), ("dres", C.c_void_p), ("dx", C.c_void_p), ("partial", C.c_void_p),
                ("dx_lp", C.c_void_p), ("dx_lp_dtype", C.c_int), ("dx_lp_drop", Dropout)]


class CastDesc(C.Structure):
    _fields_ = [("n", C.c_long), ("src", C.c_void_p), ("dst", C.c_void_p), ("drop", Dropout), ("gate", C.c_void_p)]


class TtLnUnit(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("nparts", C.c_int), ("d", C.c_int), ("a_off", C.c_long), ("b_off", C.c_long)]


class TtAux(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("lp", C.c_void_p), ("n_flat", C.c_long),
                ("n_ln", C.c_int), ("ln", C.POINTER(TtLnUnit)), ("n_chunks", C.c_int), ("chunk_off", C.POINTER(C.c_long)),
                ("chunk_len", C.POINTER(C.c_int)), ("bias_adam", C.c_int), ("state", C.c_void_p), ("grad_scale", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float)]


class DecodeStage(C.Structure):
    _fields_ = [("kind", C.c_int), ("N", C.c_int), ("K", C.c_int), ("w", C.c_void_p), ("bias", C.c_void_p), ("ln_a", C.c_void_p),
                ("ln_b", C.c_void_p), ("ln_eps", C.c_float), ("kv", C.c_void_p), ("m", C.c_int), ("mask", C.c_void_p),
                ("mask_stride", C.c_long), ("cache", C.c_void_p)]


class DecodeArgs(C.Structure):
    _fields_ = [("W", C.c_int), ("d", C.c_int), ("h", C.c_int), ("L", C.c_int), ("n_stages", C.c_int), ("d_ff", C.c_int), ("xg", C.c_void_p), ("qg", C.c_void_p),
                ("og", C.c_void_p), ("hg", C.c_void_p), ("out_lp", C.c_void_p), ("tokens", C.c_void_p), ("lut", C.c_void_p),
                ("emb_scale", C.c_float), ("pe", C.c_void_p), ("pos", C.c_void_p), ("anc", C.c_void_p), ("sync", C.c_void_p), ("dbg", C.c_void_p), ("max_m", C.c_int)]


DEC_EMBED, DEC_SELF_QKV, DEC_SELF_ATT, DEC_OUT, DEC_CROSS, DEC_FFN1, DEC_FFN2, DEC_FINAL = range(8)


class BeamArgs(C.Structure):
    _fields_ = [("dialogues", C.c_int), ("width", C.c_int), ("L", C.c_int), ("k_top", C.c_int), ("k", C.c_int), ("beam", C.c_int), ("unk", C.c_int),
                ("eos", C.c_int), ("pad", C.c_int), ("min_len", C.c_int), ("penalty", C.c_double), ("top", C.c_void_p), ("tokens", C.c_void_p),
                ("pos", C.c_void_p), ("anc", C.c_void_p), ("lp", C.c_void_p), ("n_live", C.c_void_p), ("step", C.c_void_p), ("flags", C.c_void_p),
                ("log_parent", C.c_void_p), ("log_tok", C.c_void_p), ("log_score", C.c_void_p), ("log_done", C.c_void_p), ("log_n_old", C.c_void_p),
                ("log_n_new", C.c_void_p)]


class LnFinalizeDesc(C.Structure):
    _fields_ = [("partial", C.c_void_p), ("nparts", C.c_int), ("d", C.c_int), ("da2", C.c_void_p), ("db2", C.c_void_p)]


GEMM_MAX_GROUP = 16


# every symbol include/mtn_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
class AssembleTokensDesc(C.Structure):
    _fields_ = [("flat", C.c_void_p), ("start", C.c_void_p), ("len", C.c_void_p), ("ids", C.c_void_p), ("B", C.c_int), ("L", C.c_int),
                ("pad", C.c_int64), ("out", C.c_void_p), ("mask", C.c_void_p), ("std_mask", C.c_void_p), ("n_nonpad", C.c_void_p)]


class AssembleFeaturesDesc(C.Structure):
    _fields_ = [("flat", C.c_void_p), ("start", C.c_void_p), ("len", C.c_void_p), ("ids", C.c_void_p), ("B", C.c_int), ("V", C.c_int),
                ("F", C.c_int), ("skip", C.c_int), ("out", C.c_void_p), ("mask", C.c_void_p)]


class CensusLaunch(C.Structure):
    _fields_ = [("dtype", C.c_int), ("count", C.c_int), ("variant", C.c_int), ("workgroups", C.c_int),
                ("flops", C.c_double), ("bytes", C.c_double), ("M", C.c_int * 4), ("N", C.c_int * 4), ("K", C.c_int * 4)]


SYMBOLS = {
    "mtn_assemble_tokens": (C.c_int, [C.c_int, C.POINTER(AssembleTokensDesc), _P]),
    "mtn_assemble_features": (C.c_int, [C.c_int, C.POINTER(AssembleFeaturesDesc), _P]),
    "mtn_topk_rows": (C.c_int, [_P, C.c_int, C.c_int, C.c_long, C.c_int, C.c_int, _P, _P]),
    "mtn_log_softmax_rows": (C.c_int, [_P, C.c_int, C.c_int, C.c_long, _P, C.c_long, _P]),
    "mtn_census_begin": (C.c_int, []),
    "mtn_census_end": (C.c_int, []),
    "mtn_census_info": (C.c_int, [C.c_int, C.POINTER(CensusLaunch)]),
    "mtn_census_replay": (C.c_int, [C.c_int, C.c_int, _P]),
    "mtn_census_variant_name": (C.c_char_p, [C.c_int]),
    "mtn_measure_mfma_peak": (C.c_int, [C.c_int, _P, _P, C.POINTER(C.c_double)]),
    "mtn_measure_mfma_peak_shapes": (C.c_int, [C.c_int, _P, _P, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mtn_measure_hbm_peak": (C.c_int, [_P, _P, C.c_long, _P, C.POINTER(C.c_double)]),
    "mtn_reload_env": (C.c_int, []),
    "mtn_stream_create_cu_masked": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "mtn_stream_destroy": (C.c_int, [_P]),
    "mtn_last_error": (C.c_char_p, []),
    "mtn_version": (C.c_int, []),
    "mtn_gemm": (C.c_int, [C.c_int, C.c_int, C.POINTER(GemmProblem), _P]),
    "mtn_gemm_tt_table": (C.c_int, [C.c_int, C.c_int, C.POINTER(GemmProblem), _P]),
    "mtn_decode_step": (C.c_int, [C.POINTER(DecodeArgs), _P, C.c_int, _P]),
    "mtn_beam_advance": (C.c_int, [C.POINTER(BeamArgs), _P]),
    "mtn_debug_hold_cus": (C.c_int, [C.c_int, C.c_int, C.c_int, _P]),
    "mtn_gemm_tt_table_aux": (C.c_int, [C.c_int, C.c_int, C.POINTER(GemmProblem), C.POINTER(TtAux), _P]),
    "mtn_layernorm_fwd": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mtn_layernorm_fwd_group": (C.c_int, [C.c_int, C.c_int, C.POINTER(LnFwdDesc), _P]),
    "mtn_embed_bwd_group": (C.c_int, [C.c_int, C.POINTER(EmbedBwdDesc), _P]),
    "mtn_layernorm_bwd_group": (C.c_int, [C.c_int, C.POINTER(LnBwdDesc), _P]),
    "mtn_attention_fwd_group": (C.c_int, [C.c_int, C.c_int, C.POINTER(AttnArgs), _P]),
    "mtn_attention_bwd_group": (C.c_int, [C.c_int, C.c_int, C.POINTER(AttnArgs), _P]),
    "mtn_cast_group": (C.c_int, [C.c_int, C.c_int, C.POINTER(CastDesc), _P]),
    "mtn_fused_enable": (C.c_int, [C.c_int]),
    "mtn_fused_counters": (C.c_int, [C.POINTER(C.c_long)]),
    "mtn_ln_epilogue_groups": (C.c_long, []),
    "mtn_ln_fold": (C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, _P]),
    "mtn_sublayer_group_fwd": (C.c_int, [C.c_int, C.c_int, C.POINTER(MhaArgs), C.c_int, C.POINTER(FfnArgs), _P]),
    "mtn_sublayer_group_bwd": (C.c_int, [C.c_int, C.c_int, C.POINTER(MhaArgs), C.c_int, C.POINTER(FfnArgs), _P]),
    "mtn_layernorm_bwd_partial_floats": (C.c_long, [C.c_int, C.c_int]),
    "mtn_layernorm_bwd_nparts": (C.c_int, [C.c_int]),
    "mtn_layernorm_bwd_finalize": (C.c_int, [C.c_int, C.POINTER(LnFinalizeDesc), _P]),
    "mtn_mha_param_grad_work": (C.c_int, [C.c_int, C.POINTER(MhaArgs), C.POINTER(GemmProblem), C.POINTER(LnFinalizeDesc)]),
    "mtn_ffn_param_grad_work": (C.c_int, [C.c_int, C.POINTER(FfnArgs), C.POINTER(GemmProblem), C.POINTER(LnFinalizeDesc)]),
    "mtn_layernorm_bwd": (C.c_int, [C.c_int, C.c_int, C.c_float, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "mtn_attention_fwd": (C.c_int, [C.c_int, C.POINTER(AttnArgs), _P]),
    "mtn_attention_bwd": (C.c_int, [C.c_int, C.POINTER(AttnArgs), _P]),
    "mtn_mha_sublayer_fwd": (C.c_int, [C.c_int, C.POINTER(MhaArgs), _P]),
    "mtn_mha_sublayer_bwd": (C.c_int, [C.c_int, C.POINTER(MhaArgs), _P]),
    "mtn_mha_bwd_ws_lp_elems": (C.c_long, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "mtn_mha_bwd_ws_f32_floats": (C.c_long, [C.c_int, C.c_int, C.c_int, C.c_int]),
    "mtn_ffn_sublayer_fwd": (C.c_int, [C.c_int, C.POINTER(FfnArgs), _P]),
    "mtn_ffn_sublayer_bwd": (C.c_int, [C.c_int, C.POINTER(FfnArgs), _P]),
    "mtn_ffn_bwd_ws_f32_floats": (C.c_long, [C.c_int, C.c_int, C.c_int]),
    "mtn_cast_f32_to_lp": (C.c_int, [C.c_int, C.c_long, _P, _P, _P]),
    "mtn_dropout_bwd_to_lp": (C.c_int, [C.c_int, C.c_long, _P, Dropout, _P, _P]),
    "mtn_losshead_fwd": (C.c_int, [C.POINTER(LossHeadArgs), _P]),
    "mtn_losshead_bwd": (C.c_int, [C.c_int, C.POINTER(LossHeadArgs), _P]),
    "mtn_transpose_group": (C.c_int, [C.c_int, _P, _P, _P, C.c_int, C.c_int, _P]),
    "mtn_noam_tick": (C.c_int, [_P, C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, _P]),
    "mtn_adam_step": (C.c_int, [C.c_int, C.c_long, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, _P]),
    "mtn_adam_step_chunks": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, _P]),
}

_lib = None


class MtnHipError(RuntimeError):
    pass


def load():
    """Load the HIP library or raise.  No fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("MTN_HIP_LIB", LIB_PATH)      # A/B builds of the kernels (same ABI) — development only
    if not os.path.exists(path):
        raise MtnHipError(f"{path} is missing: build it with `python -m mtn_amd.build` "
                          "(hipcc --offload-arch=gfx950).  mtn_amd has no CPU/PyTorch fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)             # AttributeError here = ABI drift: fail loudly
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def fused_counters():
    """(forward fused, forward per-stage, backward fused, backward per-stage) group counts since the library was loaded."""
    out = (C.c_long * 4)()
    check(load().mtn_fused_counters(out))
    return tuple(int(v) for v in out)


def reload_env():
    """Make the library re-read its MTN_* environment switches (they are cached per call site): call after changing one."""
    if _lib is not None:
        _lib.mtn_reload_env()


def check(rc: int):
    if rc != 0:
        raise MtnHipError(f"libmtn_hip error {rc}: {load().mtn_last_error().decode()}")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    return 0 if t is None else t.data_ptr()


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return MTN_F32
    if dt == torch.bfloat16:
        return MTN_BF16
    raise ValueError(f"unsupported compute dtype {dt}")
