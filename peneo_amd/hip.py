"""ctypes binding of libpeneo_hip.so (the C ABI declared in include/peneo_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails this
module raises — the product path never silently degrades to PyTorch ops or to the oracle.
PyTorch is used only for device memory (tensors), streams and torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PENEO_HIP_LIB") or os.path.join(_HERE, "lib", "libpeneo_hip.so")   # override: tools/ experiments

F32, BF16 = 0, 1
ACT_NONE, ACT_GELU, ACT_SILU = 0, 1, 2
MAX_HEADS = 8

_vp, _i, _i64, _f, _u32, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32, C.c_size_t


class PeneoHipError(RuntimeError):
    pass


class GemmEpilogue(C.Structure):
    _fields_ = [
        ("bias", _vp), ("act", _i), ("preact", _vp), ("ld_preact", _i64),
        ("grad_src", _vp), ("ld_grad", _i64), ("grad_act", _i),
        ("residual", _vp), ("ld_res", _i64), ("alpha", _f), ("accumulate", _i),
        ("drop_p", _f), ("drop_seed", _u32), ("pair_dz", _vp), ("pair_dz_ws", _vp), ("a_colsum", _vp),
    ]


class EncoderLayer(C.Structure):
    _fields_ = ([(n, _vp) for n in ("Wqkv", "bqkv", "Wo", "bo", "g1", "b1", "Wi", "bi", "Wo2", "bo2", "g2", "b2", "bias")] +
                [("bias_ld", _i64)] +
                [(n, _vp) for n in ("key_bias", "drop_words", "x", "qkv", "att", "lse", "h1", "m1", "r1", "a", "zi", "inter", "h2",
                                    "m2", "r2")] +
                [(n, _i) for n in ("B", "T", "H", "nh", "I", "reserved")] +
                [(n, _f) for n in ("eps", "attn_scale", "p_hidden", "p_attn")] + [("seed_o", _u32), ("seed_o2", _u32)])


class EncoderLayerGrads(C.Structure):
    _fields_ = [(n, _vp) for n in ("d_out", "d_x", "d_h2", "d_dense2", "d_zi", "d_a", "d_h1", "d_dense1", "d_att", "dqkv", "delta",
                                   "ds_out", "dwqkv", "dbqkv", "dwo", "dbo", "dg1", "db1", "dwi", "dbi", "dwo2", "dbo2", "dg2", "db2")]


class GemmProblem(C.Structure):
    _fields_ = [("M", _i), ("N", _i), ("K", _i), ("A", _vp), ("lda", _i64), ("B", _vp), ("ldb", _i64), ("C", _vp), ("ldc", _i64),
                ("accumulate", _i), ("ep", _vp)]


class CastItem(C.Structure):
    _fields_ = [("src", _vp), ("dst", _vp), ("numel", _i64)]


class AdamwTensor(C.Structure):
    _fields_ = [("param", _vp), ("grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("numel", _i64), ("lr", _f),
                ("weight_decay", _f), ("step_offset", _i), ("reserved", _i)]


class EmbedTables(C.Structure):
    _fields_ = [("word", _vp), ("type0", _vp), ("pos", _vp), ("x", _vp), ("y", _vp), ("h", _vp), ("w", _vp),
                ("coord_size", _i), ("shape_size", _i), ("max_2d", _i), ("vocab", _i), ("max_pos", _i)]


class EmbedGrads(C.Structure):
    _fields_ = [("word", _vp), ("type0", _vp), ("pos", _vp), ("x", _vp), ("y", _vp), ("h", _vp), ("w", _vp)]


class PairHeadsDesc(C.Structure):
    _fields_ = [("num_heads", _i), ("D", _i), ("classes", _i * MAX_HEADS), ("w_packed", _vp), ("b1", _vp), ("b2", _vp),
                ("drop_p", _f), ("drop_seed", _u32)]


class PairLoss(C.Structure):
    _fields_ = [("tags", _vp * MAX_HEADS), ("class_weight", _vp * MAX_HEADS), ("partials", _vp),
                ("dlogits", _vp * MAX_HEADS)]


class PairDzArgs(C.Structure):
    _fields_ = [("num_heads", _i), ("D", _i), ("classes", _i * MAX_HEADS), ("dlogits", _vp * MAX_HEADS),
                ("w2", _vp * MAX_HEADS), ("scale", _vp), ("drop_p", _f), ("drop_seed", _u32), ("drop_doc", _i),
                ("drop_pair0", _i64)]


# name -> (restype, argtypes); every symbol declared in include/peneo_hip.h appears here
SIGNATURES = {
    "peneo_version": (_i, []),
    "peneo_last_error": (C.c_char_p, []),
    "peneo_gemm_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "peneo_gemm_group": (_i, [_i, _i, _i, _i, C.POINTER(GemmProblem), _i, _vp]),
    "peneo_gemm": (_i, [_i, _i, _i, _i, _i, _i, _vp, _i64, _vp, _i64, _vp, _i64, _i, C.POINTER(GemmEpilogue), _i, _vp,
                        _sz, _vp]),
    "peneo_cast": (_i, [_vp, _i, _vp, _i, _i64, _vp]),
    "peneo_cast_multi_chunk_elems": (_i, []),
    "peneo_cast_multi": (_i, [_vp, _vp, _vp, _i, _vp]),
    "peneo_copy2d": (_i, [_i, _vp, _i64, _vp, _i64, _i64, _i64, _f, _u32, _vp]),
    "peneo_copy_rows": (_i, [_i, _vp, _i64, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _i64, _f, _u32, _vp]),
    "peneo_head_concat": (_i, [_i, _vp, _i64, _i, _f, _vp, _i64, _i, _f, _vp, _i64, _i64, _i, _vp]),
    "peneo_head_split": (_i, [_i, _vp, _i64, _vp, _i64, _i, _f, _vp, _i64, _i, _f, _i64, _i, _vp]),
    "peneo_colsum": (_i, [_i, _vp, _i64, _i64, _i64, _vp, _i, _vp]),
    "peneo_layernorm_fwd": (_i, [_i, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _f, _vp, _vp, _i64, _i, _f, _u32, _vp]),
    "peneo_layernorm_bwd": (_i, [_i, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _i64,
                                 _i, _f, _u32, _vp, _f, _u32, _vp, _vp]),
    "peneo_layernorm_bwd_partial_rows": (_i64, [_i, _i64, _i]),
    "peneo_layernorm_bwd_partial": (_i, [_i, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _i64,
                                         _i, _f, _u32, _vp, _f, _u32, _vp]),
    "peneo_position_ids": (_i, [_vp, _i, _i, _i64, _vp, _vp]),
    "peneo_embed_text_fwd": (_i, [_i, _vp, _vp, _vp, C.POINTER(EmbedTables), _i, _i, _i, _i, _vp, _i64, _i64, _vp, _vp]),
    "peneo_embed_text_bwd": (_i, [_i, _vp, _i64, _i64, _vp, _vp, _vp, C.POINTER(EmbedGrads), _i, _i, _i, _i, _i, _i, _i,
                                  _i64, _vp]),
    "peneo_im2col_patch16": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _vp]),
    "peneo_visual_assemble_fwd": (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _vp, _vp]),
    "peneo_visual_assemble_bwd": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "peneo_relpos_inputs": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "peneo_relpos_buckets": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp]),
    "peneo_relpos_bias_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _i, _i, _i, _i, _vp, _vp, _vp]),
    "peneo_relpos_bias_bwd_layers": (_i, [_vp, _i, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _i, _i, _i, _i, _vp]),
    "peneo_relpos_bias_bwd": (_i, [_vp, _i64, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _f, _i, _i, _i, _vp]),
    "peneo_attn_padded_len": (_i, [_i]),
    "peneo_attn_padded_dim": (_i, [_i]),
    "peneo_head_transpose": (_i, [_i, _vp, _i64, _i, _i, _i, _i, _vp, _vp]),
    "peneo_attn_fwd": (_i, [_i, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _i, _f, _vp, _i64, _vp, _vp, _i64, _vp, _f, _vp, _vp]),
    "peneo_attn_drop_words_dims": (None, [_i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "peneo_attn_drop_words_count": (_i64, [_i, _i, _i]),
    "peneo_attn_drop_words": (_i, [_vp, _i, _i, _i, _f, _u32, _vp]),
    "peneo_attn_bwd": (_i, [_i, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i, _i, _i, _i, _f, _vp, _i64, _vp,
                            _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _f, _vp, _vp]),
    "peneo_pair_heads_packed_bytes": (_sz, [_i, _i, _i]),
    "peneo_pair_heads_pack": (_i, [_i, _vp, _vp, _vp, _i, _i, _vp, _vp]),
    "peneo_pair_heads_fwd": (_i, [_i, _vp, _i, _i, C.POINTER(PairHeadsDesc), _vp, C.POINTER(PairLoss), _vp]),
    "peneo_pair_x_fwd": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _vp, _vp]),
    "peneo_pair_x_bwd": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp]),
    "peneo_pair_dz_workspace_bytes": (_sz, [_i, _i]),
    "peneo_pair_dz": (_i, [_i, _vp, _i64, C.POINTER(PairDzArgs), _vp, _vp]),
    "peneo_pair_dz_fused": (_i, [_i, _vp, _i, _i, _i, _i, _vp, _vp, C.POINTER(PairDzArgs), _vp, _vp, _vp, _vp, _vp]),
    "peneo_pair_loss_partials": (_i64, [_i, _i]),
    "peneo_loss_finish": (_i, [_vp, _i64, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    "peneo_weighted_ce": (_i, [_vp, _vp, _vp, _i64, _i, _vp, _vp, _vp, _vp]),
    "peneo_pair_bwd_supported": (_i, [_i, _i, _i]),
    "peneo_pair_bwd_rows": (_i64, [_i]),
    "peneo_pair_bwd_packed_bytes": (_sz, [_i, _i]),
    "peneo_pair_bwd_pack": (_i, [_vp, _i, _i, _vp, _vp]),
    "peneo_pair_bwd_partial_bytes": (_sz, [_i, _i, _i]),
    "peneo_pair_bwd_fused": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "peneo_pair_save_supported": (_i, [_i, _i, _i]),
    "peneo_pair_save_bytes": (_sz, [_i, _i, _i, _i]),
    "peneo_pair_loss_partials_save": (_i64, [_i, _i]),
    "peneo_pair_heads_fwd_save": (_i, [_i, _vp, _i, _i, C.POINTER(PairHeadsDesc), _vp, C.POINTER(PairLoss), _vp, _vp, _vp]),
    "peneo_pair_bwd_saved": (_i, [_i, _vp, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "peneo_ohem_workspace_bytes": (_sz, [_i64]),
    "peneo_ohem_ce": (_i, [_vp, _vp, _vp, _i64, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp]),
    "peneo_ohem_finish": (_i, [_vp, _vp, _i, _vp, _vp, _vp, _vp]),
    "peneo_adamw_chunk_elems": (_i, []),
    "peneo_adamw_step": (_i, [_vp, _vp, _vp, _i, _f, _f, _f, _i, _vp]),
    "peneo_grad_sqnorm": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "peneo_grad_sqnorm_slots": (_i, []),
    "peneo_adamw_step_clip": (_i, [_vp, _vp, _vp, _i, _f, _f, _f, _i, _vp, _f, _vp]),
    "peneo_struct_bytes": (C.c_size_t, [_i]),
    "peneo_encoder_layer_workspace_bytes": (C.c_size_t, [_i, _i, _i, _i]),
    "peneo_encoder_layer_fwd": (_i, [_vp, _vp, _vp, C.c_size_t, _vp]),
    "peneo_encoder_layer_bwd": (_i, [_vp, _vp, _vp, C.c_size_t, _vp, C.c_size_t, _vp, _vp]),
    "peneo_spots_to_tags": (_i, [_vp, _i, _i, _i, _vp, _vp, _vp]),
    "peneo_spots_compact": (_i, [_vp, _i64, _i, _i, _vp, _vp, _vp, _i, _vp]),
    "peneo_gemm_set_big_mode": (None, [_i]),    # diagnostics block of the header: process-wide, not thread-safe
    "peneo_gemm_set_sk_mode": (None, [_i]),
    "peneo_gemm_sk_set_prof": (None, [_vp]),
    "peneo_gemm_sk_set_max_groups": (None, [_i]),
}

_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """Load libpeneo_hip.so and bind every entry point.  Raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PeneoHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C peneo_amd/csrc`).  peneo_amd has no CPU / PyTorch fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def lib() -> C.CDLL:
    return _lib if _lib is not None else load_library()


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().peneo_last_error().decode("utf-8", "replace")
        raise PeneoHipError(f"{what or 'peneo_hip'} failed ({rc}): {msg}")


# ----------------------------------------------------------------------------------------------
# tensor helpers
# ----------------------------------------------------------------------------------------------
def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return F32
    if t == torch.bfloat16:
        return BF16
    raise PeneoHipError(f"unsupported dtype {t}")


def torch_dtype(code: int) -> torch.dtype:
    return torch.float32 if code == F32 else torch.bfloat16


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise PeneoHipError("peneo_hip kernels need device tensors (no CPU path exists)")
    return t.data_ptr()


# the current stream's raw handle: torch.cuda.current_stream().cuda_stream builds a Python Stream object per call (~9 us; a LiLT
# train step asks ~300 times), the C accessor behind it answers in ~0.3 us
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def stream() -> int:
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def ptr_array(ts: Sequence[Optional[torch.Tensor]], n: int = MAX_HEADS):
    arr = (_vp * n)()
    for i, t in enumerate(ts):
        arr[i] = ptr(t)
    return arr
