"""ctypes binding of libtt.so (include/tt.h).  There is NO fallback: if the HIP
library is missing or a call fails, the error is raised to the caller."""
from __future__ import annotations

import ctypes as C
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libtt.so"

TT_OK, TT_ERR_BAD_SHAPE, TT_ERR_BAD_INDEX, TT_ERR_ZERO_LENGTH, TT_ERR_UNSUPPORTED, TT_ERR_WORKSPACE, TT_ERR_HIP = range(7)
TT_ENC_ONE_WORKGROUP = 0x100  # option bit of the encoder calls (include/tt.h)
TT_ENC_PHASE_BEGIN, TT_ENC_PHASE_FINISH = 0x200, 0x400  # tt_encoder_forward_f32 in two halves
TT_ENC_SEED_ON_DEVICE = 0x800  # dropout_seed is the address of a device uint64
TT_ENC_F32 = 0x1000  # every product of the call on the fp32-MFMA kernels (the reference's own arithmetic)
TT_ENC_PROJECTED = 0x2000  # tt_encoder_workspace_bytes: size for tt_encoder_forward_projected_f32
TT_STEP_GATE_WORDS = 4


class EncSync(C.Structure):
    """tt_enc_sync_t (include/tt.h): events that order the recurrences of encoder calls on different streams."""
    _fields_ = [("wait_before_recurrence", C.c_void_p), ("record_after_recurrence", C.c_void_p)]
TT_TOPK_INVALID_INDEX = 1 << 62


class TTError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"libtt error {code}: {msg}")
        self.code = code


_vp, _i, _i64, _sz, _f, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float, C.c_uint64

# name -> (restype, argtypes); mirrors include/tt.h one to one
SIGNATURES = {
    "tt_version": (C.c_char_p, []),
    "tt_last_error": (C.c_char_p, []),
    "tt_score_topk_workspace_bytes": (_sz, [_i, _i64, _i, _i]),
    "tt_score_topk_pace_timeouts_offset": (_sz, [_i, _i64, _i, _i]),
    "tt_score_topk_redo_flags_offset": (_sz, [_i, _i64, _i, _i]),
    "tt_score_topk_f32": (_i, [_vp, _i, _i, _vp, _i64, _i, _i64, _vp, _vp, _vp, _sz, _vp]),
    "tt_score_topk_partials_f32": (_i, [_vp, _i, _i, _vp, _i64, _i, _i64, _vp, _sz, _vp, _vp, _vp, _vp, _vp]),
    "tt_index_build_f16": (_i, [_vp, _i64, _i, _vp, _vp, _vp]),
    "tt_index_build_from_bf16": (_i, [_vp, _i64, _i, _vp, _vp, _vp, _i, _vp]),
    "tt_score_topk_screened_workspace_bytes": (_sz, [_i, _i64, _i, _i]),
    "tt_score_topk_screened_stats_offset": (_sz, [_i, _i64, _i, _i]),
    "tt_score_topk_screened_seed_f32": (_i, [_vp, _i, _i, _vp, _i64, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "tt_score_topk_screened_seed_list_f32": (_i, [_vp, _i, _i, _vp, _i64, _i, _i, _f, _vp, _vp, _vp, _sz, _vp]),
    "tt_seed_union_f32": (_i, [_vp, _i, _i, _i, _i, _vp, _vp]),
    "tt_score_topk_screened_seeded_f32": (_i, [_vp, _i, _i, _vp, _vp, _i64, _i, _f, _i64, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "tt_score_topk_screened_f32": (_i, [_vp, _i, _i, _vp, _vp, _i64, _i, _f, _i64, _vp, _vp, _vp, _vp, _sz, _vp, _vp]),
    "tt_event_create": (_i, [_vp]),
    "tt_event_destroy": (_i, [_vp]),
    "tt_event_elapsed_ms": (_i, [_vp, _vp, _vp]),
    "tt_topk_merge": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp, _vp]),
    "tt_topk_merge_shards": (_i, [_vp, _i, _sz, _sz, _i, _i, _i, _vp, _vp, _vp]),
    "tt_score_rank_f32": (_i, [_vp, _i, _i, _vp, _i64, _vp, _vp, _vp]),
    "tt_score_all_f32": (_i, [_vp, _i, _i, _vp, _i64, _vp, _vp]),
    "tt_tok_create": (_i, [_vp, _vp, _vp, _i64, _i64, _vp]),
    "tt_tok_destroy": (None, [_vp]),
    "tt_tok_encode": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _i]),
    "tt_tok_encode_sep": (_i, [_vp, _vp, _i64, C.c_char, _i64, _vp, _vp, _vp, _vp, _i]),
    "tt_tok_encode_ptrs": (_i, [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i]),
    "tt_tok_set_unicode": (_i, [_vp, _vp, _vp, _i64]),
    "tt_tok_encode_units": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _i]),
    "tt_tok_pad": (_i, [_vp, _vp, _vp, _i64, _i64, _vp, _i]),
    "tt_tok_pad_i32": (_i, [_vp, _vp, _vp, _i64, _i64, _vp, _i]),
    "tt_encoder_split_workgroups": (_i, [_i, _i, _i, _i]),
    "tt_encoder_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i]),
    "tt_encoder_forward_f32": (_i, [_vp, _i, _i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _i, _f, _u64, _vp, _vp,
                                    _sz, _vp, _vp, _vp]),
    "tt_concat_ids_i64": (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _i, _vp]),
    "tt_encoder_prepared_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "tt_encoder_prepare_f32": (_i, [_i, _i, _i, _i, _i, _vp, _vp, _sz, _vp]),
    "tt_encoder_forward_prepared_f32": (_i, [_vp, _i, _i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp,
                                             _sz, _vp, _vp]),
    "tt_encoder_projected_bytes": (_sz, [_i64, _i, _i, _i, _i]),
    "tt_encoder_project_table_f32": (_i, [_vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _sz, _vp]),
    "tt_encoder_forward_projected_f32": (_i, [_vp, _i, _i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _vp, _vp,
                                              _sz, _vp, _vp]),
    "tt_encoder_backward_f32": (_i, [_vp, _i, _i, _vp, _i64, _i, _i, _i, _i, _i, _vp, _vp, _vp, _i, _f, _u64, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _sz, _i, _vp, _vp, _vp]),
    "tt_triplet_loss_f32": (_i, [_vp, _vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    "tt_allgather_topk": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "tt_allreduce_grads": (_i, [_vp, _vp, _i64, _vp]),
    "tt_comm_library": (C.c_char_p, []),
    "tt_comm_unique_id": (_i, [_vp]),
    "tt_comm_init_rank": (_i, [_vp, _i, _vp, _i]),
    "tt_comm_info": (_i, [_vp, _vp, _vp]),
    "tt_comm_destroy": (_i, [_vp]),
    "tt_clip_adam_scratch_bytes": (_sz, []),
    "tt_clip_adam_step_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp]),
    "tt_clip_adam_step_gated_f32": (_i, [_vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f, _f, _f, _vp, _vp, _vp, _vp]),
    "tt_step_gate_f32": (_i, [_vp, _i, _vp, _vp]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m twotowermlretrieval_amd.build` "
                "(hipcc, gfx950).  There is no CPU or PyTorch fallback for this path.")
        l = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != TT_OK:
        msg = lib().tt_last_error().decode("utf-8", "replace")
        if rc == TT_ERR_BAD_INDEX:
            raise IndexError(msg)
        if rc == TT_ERR_BAD_SHAPE:
            raise ValueError(msg)
        raise TTError(rc, msg)
