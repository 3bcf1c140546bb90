"""ctypes binding of libctts_hip.so (the C ABI declared in include/ctts.h).

cffi is not available in the image; ctypes is the binding.  The library is
built in-tree by csrc/build.sh (hipcc --offload-arch=gfx950).  There is NO fallback: if the
shared object is missing or a kernel reports an error, the call raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CTTS_LIB") or os.path.join(_HERE, "csrc", "libctts_hip.so")   # CTTS_LIB: tuning A/B builds

_c_f32p = C.c_void_p
_i32, _i64, _f32, _u32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_uint32, C.c_void_p


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", _vp), ("B", _vp), ("C", _vp),
        ("M", _i32), ("N", _i32), ("K", _i32),
        ("lda", _i64), ("ldb", _i64), ("ldc", _i64),
        ("a_kc", _i32), ("b_kc", _i32),
        ("nb0", _i32), ("nb1", _i32),
        ("sA0", _i64), ("sA1", _i64), ("sB0", _i64), ("sB1", _i64), ("sC0", _i64), ("sC1", _i64),
        ("lens", _vp),
        ("lim_m", _i32), ("lim_n", _i32), ("lim_k", _i32),
        ("conv_T", _i32), ("conv_pad", _i32), ("conv_cin", _i32), ("conv_on_b", _i32),
        ("split_k", _i32),
        ("alpha", _f32),
        ("bias", _vp),
        ("Z", _vp), ("ldz", _i64),
        ("act", _i32),
        ("p_drop", _f32), ("seed", _vp), ("drop_offset", _u32),
        ("R", _vp), ("ldr", _i64),
        ("rowscale", _vp),
        ("row_lens", _vp), ("row_T", _i32), ("row_halo", _i32),
        ("tile_map", _vp),
        ("tile_group_n", _i32),
        ("E", _vp), ("rowsub", _vp),
        ("sk_ws", _vp), ("sk_ws_bytes", _i64),
        ("epi_bwd", _i32),
        ("split_out", _vp), ("split_out_floats", _i64),
        ("split_overwrite", _i32),
        ("bf16_split", _i32),
        ("A_planes", _vp),
        ("B_planes", _vp),
        ("C_planes", _vp),
    ]


class PsumTask(C.Structure):
    """ctts_psum_task of include/ctts.h"""
    _fields_ = [("src", _vp), ("dst", _vp), ("n", _i64), ("stride", _i64), ("count", _i32), ("alpha", _f32)]


class SplitTask(C.Structure):
    """ctts_split_task of include/ctts.h"""
    _fields_ = [("src", _vp), ("dst", _vp), ("rows", _i64), ("cols", _i64), ("ld", _i64)]


class RepackTask(C.Structure):
    """ctts_repack_task of include/ctts.h"""
    _fields_ = [("src", _vp), ("dst", _vp), ("cout", _i32), ("cin", _i32), ("k", _i32)]


# name -> argtypes (every function returns int status except the two listed below)
_SIGNATURES = {
    "ctts_conv_dgrad_weights": [C.POINTER(RepackTask), C.c_int, _vp],
    "ctts_gemm_split_plan": [C.POINTER(GemmDesc), C.POINTER(_i32), C.POINTER(_i64)],
    "ctts_partial_sums": [C.POINTER(PsumTask), C.c_int, _vp],
    "ctts_reduce_parts": [C.c_int, _i64, C.c_int],
    "ctts_xcd_probe": [_vp, C.c_int, _vp],
    "ctts_posembed_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, _f32, _vp, _u32, _vp],
    "ctts_posembed_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, _f32, _vp, _u32, C.c_int, _vp, _vp],
    "ctts_gemm": [C.POINTER(GemmDesc), _vp],
    "ctts_gemm_takes_persistent": [C.POINTER(GemmDesc)],
    "ctts_gemm_ws_enable": [C.c_int],
    "ctts_gemm_takes_weight_stationary": [C.POINTER(GemmDesc)],
    "ctts_gemm_takes_bf16_split": [C.POINTER(GemmDesc)],
    "ctts_gemm_takes_planes": [C.POINTER(GemmDesc)],
    "ctts_split_planes": [C.POINTER(SplitTask), C.c_int, _vp],
    "ctts_rowdot_heads": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_mel_prepare": [_vp, C.c_int, C.c_int, _vp, _vp],
    "ctts_mel_spectrogram": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _f32, C.c_int, _vp, _vp],
    "ctts_mha_fwd": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f32, _vp],
    "ctts_mha_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f32, C.c_int, _vp, _vp],
    "ctts_relmha_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f32, _f32, _vp, _u32, _vp],
    "ctts_relmha_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int,
                        _f32, _f32, _vp, _u32, _vp],
    "ctts_weighted_colsum": [_vp, _vp, _vp, _i64, C.c_int, _f32, C.c_int, _vp, _vp, _vp],
    "ctts_epilogue_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, C.c_int, _f32, _vp, _u32, _f32, C.c_int, _vp, _vp, _vp],
    "ctts_row_tile_map": [_vp, C.c_int, C.c_int, C.c_int, _vp, _vp],
    "ctts_conv_weight_repack": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_lr_index": [_vp, C.c_int, C.c_int, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp],
    "ctts_lr_gather_fwd": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_lr_gather_bwd": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_embedding_fwd": [_vp, _vp, _vp, _i64, C.c_int, C.c_int, _vp],
    "ctts_embedding_bwd": [_vp, _vp, _vp, _i64, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp],
    "ctts_positions": [_vp, C.c_int, _i64, C.c_int, C.c_int, _vp, _vp],
    "ctts_cwt_pitch": [_vp, _i64, C.c_int, _vp, _vp, _f32, _vp, C.c_int, _f32, _f32, _f32, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _f32, _f32, _vp, _u32, _vp, _vp, _vp],
    "ctts_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, _f32, _vp, _u32, _vp, C.c_int, _vp, _vp, _vp, _vp],
    "ctts_colstats": [_vp, _vp, C.c_int, C.c_int, _vp, _vp],
    "ctts_bn_finalize": [_vp, C.c_int, C.c_int, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp],
    "ctts_bn_apply": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _f32, _vp, _u32, _vp, _vp],
    "ctts_bn_bwd_reduce": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _f32, _vp, _u32, _vp, _vp],
    "ctts_bn_bwd_apply": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _f32, _vp,
                          _u32, C.c_int, _vp, _vp],
    "ctts_softmax_fwd": [_vp, _vp, C.c_int, C.c_int, C.c_int, _i64, _vp],
    "ctts_softmax_bwd": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _i64, _vp],
    "ctts_act_dropout_bwd": [_vp, _vp, _vp, _i64, C.c_int, C.c_int, _f32, _f32, _vp, _u32, _vp],
    "ctts_rowscale_dropout": [_vp, _vp, _i64, C.c_int, _vp, _f32, _vp, _u32, _vp],
    "ctts_colsum": [_vp, _vp, _i64, C.c_int, _i64, _f32, C.c_int, _vp, _vp, _vp],
    "ctts_reflect_pad": [_vp, _vp, C.c_int, C.c_int, C.c_int, _i64, _vp],
    "ctts_stft_magnitude": [_vp, _i64, _vp, _i64, _vp, _i64, C.c_int, _vp],
    "ctts_log_clamp_transpose": [_vp, _vp, C.c_int, C.c_int, C.c_int, _f32, _vp],
    "ctts_relattn_split_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, _vp],
    "ctts_relattn_split_bwd": [_vp, _vp, _vp, _vp, _i64, C.c_int, _vp],
    "ctts_glu_fwd": [_vp, _vp, _i64, C.c_int, _vp],
    "ctts_glu_bwd": [_vp, _vp, _vp, _i64, C.c_int, _vp],
    "ctts_dwconv_fwd": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_dwconv_wgrad": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_relpos_softmax_fwd": [_vp, _vp, _vp, C.c_int, C.c_int, _f32, _f32, _vp, _u32, _vp],
    "ctts_relpos_softmax_bwd": [_vp, _vp, C.c_int, C.c_int, _f32, _f32, _vp, _u32, _vp],
    "ctts_relshift_bwd": [_vp, _vp, C.c_int, C.c_int, _vp],
    "ctts_neg_sqdist": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _f32, _vp],
    "ctts_mas": [_vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_forward_sum_fwd": [_vp, _vp, _vp, _f32, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_forward_sum_bwd": [_vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_var_loss_fwd": [_vp] * 2 + [C.c_int] + [_vp] * 12 + [C.c_int] * 3 + [_vp, C.c_int, _vp] + [_vp] * 4 + [_vp],
    "ctts_var_loss_bwd": [_vp] * 2 + [C.c_int] + [_vp] * 12 + [C.c_int] * 3 + [_vp, C.c_int, _vp] + [_vp] * 4 + [_vp] * 5 + [_vp],
    "ctts_bin_loss_fwd": [_vp, _vp, _i64, _vp, _vp, _vp],
    "ctts_bin_loss_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _vp],
    "ctts_masked_loss_fwd": [_vp, _vp, _vp, _i64, _i32, _vp, _vp, _vp],
    "ctts_masked_loss_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp],
    "ctts_mel_l1_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, _vp, _vp],
    "ctts_mel_l1_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, C.c_int, _vp],
    "ctts_adam_clip_step": [_vp, _vp, _vp, _vp, C.c_int64, _vp, _f32, _f32, _f32, _f32, _f32, _vp, _vp],
    "ctts_im2col_3x3s2": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_col2im_3x3s2": [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_gru_fwd": [_vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_gru_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_softmax_rect_fwd": [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_softmax_rect_bwd": [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp],
    "ctts_comm_unique_id": [_vp],
    "ctts_comm_create": [C.POINTER(_vp), _i32, _i32, _vp],
    "ctts_comm_destroy": [_vp],
    "ctts_allreduce_mean": [_vp, _i64, _vp, _vp],
}
COMM_ID_BYTES = 128                   # CTTS_COMM_ID_BYTES of include/ctts.h
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ["ctts_last_error", "ctts_version", "ctts_mha_supported", "ctts_relmha_workspace_floats",
                                               "ctts_mel_spectrogram_workspace_bytes", "ctts_gemm_workspace_bytes", "ctts_workspace_bytes",
                                               "ctts_workspace_error_word"])
ADAM_STATE_FLOATS = 3 + 2048          # CTTS_ADAM_STATE_FLOATS of include/ctts.h

_lib = None


class CttsError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise CttsError(
            f"{LIB_PATH} not found - build it with comprehensive-transformer-tts_amd/csrc/build.sh "
            "(or __graft_entry__.build()).  There is no CPU / PyTorch fallback for the hot path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.ctts_last_error.restype = C.c_char_p
    lib.ctts_last_error.argtypes = []
    lib.ctts_version.restype = C.c_int
    lib.ctts_version.argtypes = []
    lib.ctts_mha_supported.restype = C.c_int
    lib.ctts_mha_supported.argtypes = [C.c_int, C.c_int]
    lib.ctts_mel_spectrogram_workspace_bytes.restype = C.c_size_t
    lib.ctts_mel_spectrogram_workspace_bytes.argtypes = [C.c_int, C.c_int]
    lib.ctts_gemm_workspace_bytes.restype = C.c_size_t
    lib.ctts_gemm_workspace_bytes.argtypes = []
    lib.ctts_workspace_bytes.restype = C.c_size_t
    lib.ctts_workspace_bytes.argtypes = []
    lib.ctts_workspace_error_word.restype = C.c_void_p
    lib.ctts_workspace_error_word.argtypes = [C.c_void_p]
    lib.ctts_relmha_workspace_floats.restype = C.c_size_t
    lib.ctts_relmha_workspace_floats.argtypes = [C.c_int, C.c_int, C.c_int]
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        msg = load().ctts_last_error().decode(errors="replace")
        raise CttsError(f"{what} failed with status {status}: {msg}")
