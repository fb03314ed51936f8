"""ctypes binding of libconmamba_hip.so (C ABI: include/conmamba_hip.h).

The library is built in-tree (``mamba_asr_amd/lib/libconmamba_hip.so``) by
``__graft_entry__.build()`` / ``make -C mamba_asr_amd/csrc``.  There is no CPU fallback:
if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CM_LIB_PATH") or os.path.join(_HERE, "lib", "libconmamba_hip.so")    # CM_LIB_PATH: A/B builds of the same ABI

CM_F32, CM_BF16, CM_F16 = 0, 1, 2
CM_SCAN_CHUNK = 64
ABI_VERSION = 10

i32, i64, vp, fp = C.c_int32, C.c_int64, C.c_void_p, C.c_void_p


class ScanFwdArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("dim", i32), ("seqlen", i32), ("dstate", i32),
        ("io_dtype", i32), ("bc_dtype", i32), ("delta_softplus", i32), ("reverse_time", i32),
        ("u", vp), ("delta", vp), ("A", fp), ("B", vp), ("C", vp), ("D", fp), ("z", vp), ("delta_bias", fp),
        ("out", vp), ("out_z", vp), ("x", fp),
        ("u_bs", i64), ("u_ds", i64), ("delta_bs", i64), ("delta_ds", i64), ("z_bs", i64), ("z_ds", i64),
        ("out_bs", i64), ("out_ds", i64), ("B_bs", i64), ("B_ns", i64), ("C_bs", i64), ("C_ns", i64),
        ("stream", vp), ("h0", fp), ("lanes_per_channel", i32), ("pad4_", i32),
    ]


class ScanBwdArgs(C.Structure):
    _fields_ = [
        ("fwd", ScanFwdArgs),
        ("dout", vp), ("dout_bs", i64), ("dout_ds", i64),
        ("du", vp), ("ddelta", vp), ("dz", vp),
        ("du_bs", i64), ("du_ds", i64), ("ddelta_bs", i64), ("ddelta_ds", i64), ("dz_bs", i64), ("dz_ds", i64),
        ("dA", fp), ("dB", fp), ("dC", fp), ("dD", fp), ("ddelta_bias", fp),
        ("workspace", vp), ("workspace_bytes", i64),
    ]


class ConvArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("dim", i32), ("seqlen", i32), ("width", i32),
        ("io_dtype", i32), ("silu", i32), ("reverse_time", i32),
        ("x", vp), ("weight", fp), ("bias", fp), ("y", vp),
        ("x_bs", i64), ("x_ds", i64), ("y_bs", i64), ("y_ds", i64),
        ("dy", vp), ("dx", vp), ("dweight", fp), ("dbias", fp),
        ("dy_bs", i64), ("dy_ds", i64), ("dx_bs", i64), ("dx_ds", i64),
        ("stream", vp), ("workspace", fp),
    ]


class ScanClDir(C.Structure):
    _fields_ = [
        ("u", vp), ("delta", vp), ("A", fp), ("B", fp), ("C", fp), ("dt_low", fp), ("dt_weight", fp), ("D", fp),
        ("delta_bias", fp), ("out", vp),
        ("u_bs", i64), ("u_ts", i64), ("delta_bs", i64), ("delta_ts", i64), ("out_bs", i64), ("out_ts", i64),
        ("bc_ns", i64), ("bc_bs", i64), ("reverse_time", i32), ("dt_rank", i32),
        ("xdbl", vp), ("xdbl_bs", i64), ("xdbl_ts", i64),
        ("h0", fp), ("h_last", fp), ("decay", fp),
        ("ckpt", fp), ("ypre", vp), ("ypre_bs", i64), ("ypre_ts", i64),
    ]


class ScanClArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("seqlen", i32), ("dim", i32), ("dstate", i32),
        ("io_dtype", i32), ("delta_softplus", i32), ("ndir", i32), ("time_chunks", i32),
        ("z", vp), ("z_bs", i64), ("z_ts", i64),
        ("dir", ScanClDir * 2),
        ("stream", vp),
        ("workspace", vp), ("workspace_bytes", i64), ("lanes_per_channel", i32), ("pad5_", i32),
    ]


class CtcArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("T", i32), ("V", i32), ("S", i32), ("blank", i32), ("Sx_max", i32),
        ("log_probs", fp), ("targets", vp), ("input_lengths", vp), ("target_lengths", vp), ("nll", fp), ("grad", fp),
        ("workspace", fp), ("workspace_floats", i64), ("alpha", fp), ("beta", fp), ("stream", vp),
    ]


class FfnElemArgs(C.Structure):
    _fields_ = [
        ("rows", i64), ("dim", i32), ("io_dtype", i32), ("act", i32), ("dy_f32", i32),
        ("a", vp), ("bias", fp), ("res", fp), ("y", vp), ("mask", vp), ("dy", vp), ("da", vp), ("dbias", fp), ("dbias_part", fp),
        ("p", C.c_float), ("alpha", C.c_float), ("seed", C.c_uint64), ("stream", vp), ("act_out", vp), ("overwrite", i32), ("reserved0", i32),
        ("seed_epoch", vp),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("rows", i32), ("m", i32), ("n", i32), ("variant", i32), ("a", vp), ("b", vp), ("lda", i64), ("ldb", i64),
        ("out", fp), ("workspace", fp), ("workspace_floats", i64), ("stream", vp),
    ]


class ConvClBwdArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("seqlen", i32), ("dim", i32), ("width", i32), ("io_dtype", i32), ("pad_", i32),
        ("x", vp), ("weight_f", fp), ("bias_f", fp), ("weight_b", fp), ("bias_b", fp),
        ("du_f", vp), ("du_b", vp), ("dz_f", vp), ("dz_b", vp), ("dx", vp), ("dz", vp),
        ("dweight_f", fp), ("dbias_f", fp), ("dweight_b", fp), ("dbias_b", fp),
        ("x_bs", i64), ("x_ts", i64), ("duf_bs", i64), ("duf_ts", i64), ("dub_bs", i64), ("dub_ts", i64),
        ("dzf_bs", i64), ("dzf_ts", i64), ("dzb_bs", i64), ("dzb_ts", i64), ("dx_bs", i64), ("dx_ts", i64), ("dz_bs", i64), ("dz_ts", i64),
        ("stream", vp), ("workspace", fp), ("workspace_floats", i64), ("overwrite", i32), ("reserved0", i32),
    ]


class ScanClBwdDir(C.Structure):
    _fields_ = [
        ("u", vp), ("xdbl", vp), ("A", fp), ("dt_weight", fp), ("D", fp), ("delta_bias", fp), ("ckpt", fp), ("ypre", vp),
        ("dout", vp), ("du", vp), ("dz", vp), ("dxdbl", vp), ("dA", fp), ("ddt_weight", fp), ("dD", fp), ("ddelta_bias", fp),
        ("u_bs", i64), ("u_ts", i64), ("xdbl_bs", i64), ("xdbl_ts", i64), ("ypre_bs", i64), ("ypre_ts", i64),
        ("dout_bs", i64), ("dout_ts", i64), ("du_bs", i64), ("du_ts", i64), ("dz_bs", i64), ("dz_ts", i64),
        ("dxdbl_bs", i64), ("dxdbl_ts", i64), ("reverse_time", i32), ("dt_rank", i32),
    ]


class ScanClBwdArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("seqlen", i32), ("dim", i32), ("dstate", i32), ("io_dtype", i32), ("ndir", i32),
        ("z", vp), ("z_bs", i64), ("z_ts", i64), ("time_chunks", i32), ("overwrite", i32),
        ("dir", ScanClBwdDir * 2),
        ("stream", vp), ("workspace", vp), ("workspace_bytes", i64), ("da_log", i32), ("reserved0", i32),
    ]


class ConvClArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("seqlen", i32), ("dim", i32), ("width", i32), ("io_dtype", i32), ("silu", i32),
        ("x", vp), ("weight_f", fp), ("bias_f", fp), ("weight_b", fp), ("bias_b", fp), ("y_fwd", vp), ("y_bwd", vp),
        ("x_bs", i64), ("x_ts", i64), ("yf_bs", i64), ("yf_ts", i64), ("yb_bs", i64), ("yb_ts", i64),
        ("stream", vp),
    ]


class ConvXprojArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("seqlen", i32), ("dim", i32), ("width", i32),
        ("x", vp), ("weight_f", fp), ("bias_f", fp), ("weight_b", fp), ("bias_b", fp), ("wx_f", vp), ("wx_b", vp),
        ("y_fwd", vp), ("y_bwd", vp), ("xdbl", vp),
        ("x_bs", i64), ("x_ts", i64), ("yf_bs", i64), ("yf_ts", i64), ("yb_bs", i64), ("yb_ts", i64),
        ("xdbl_bs", i64), ("xdbl_ts", i64), ("stream", vp), ("dt_pad", i32), ("variant", i32),
    ]


class ConvUpdateArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("dim", i32), ("width", i32), ("io_dtype", i32), ("silu", i32), ("pad_", i32),
        ("x", vp), ("conv_state", fp), ("weight", fp), ("bias", fp), ("out", vp), ("stream", vp),
    ]


class StateUpdateArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("dim", i32), ("dstate", i32), ("io_dtype", i32), ("dt_softplus", i32), ("pad_", i32),
        ("state", fp), ("x", vp), ("dt", vp), ("A", fp), ("B", vp), ("C", vp), ("D", fp), ("z", vp), ("dt_bias", fp),
        ("out", vp), ("stream", vp),
    ]


class Dwconv1dArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("dim", i32), ("seqlen", i32), ("ksize", i32), ("pad_left", i32), ("io_dtype", i32),
        ("x", vp), ("weight", fp), ("bias", fp), ("y", vp), ("dy", vp), ("dx", vp), ("dweight", fp), ("dbias", fp),
        ("x_bs", i64), ("x_ds", i64), ("y_bs", i64), ("y_ds", i64), ("dy_bs", i64), ("dy_ds", i64), ("dx_bs", i64), ("dx_ds", i64),
        ("stream", vp),
    ]


class DwconvClArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("dim", i32), ("seqlen", i32), ("ksize", i32), ("pad_left", i32), ("io_dtype", i32),
        ("x", vp), ("weight", fp), ("bias", fp), ("y", vp), ("dy", vp), ("dx", vp), ("dweight", fp), ("dbias", fp), ("partial", fp),
        ("x_bs", i64), ("x_ts", i64), ("y_bs", i64), ("y_ts", i64), ("dy_bs", i64), ("dy_ts", i64), ("dx_bs", i64), ("dx_ts", i64),
        ("stream", vp), ("overwrite", i32), ("reserved0", i32),
    ]


class LayerNormArgs(C.Structure):
    # (dres appended in ABI 7)
    _fields_ = [
        ("rows", i64), ("dim", i32), ("x_dtype", i32), ("y_dtype", i32), ("eps", C.c_float),
        ("x", vp), ("gamma", fp), ("beta", fp), ("y", vp), ("mean", fp), ("rstd", fp),
        ("dy", vp), ("dx", vp), ("dgamma", fp), ("dbeta", fp), ("workspace", fp), ("stream", vp), ("dres", fp),
        ("act", i32), ("act_slope", C.c_float), ("chan_mask", fp), ("mask_rows", i32), ("mask_c", i32),
    ]


class LnPwGluArgs(C.Structure):
    _fields_ = [
        ("rows", i32), ("dim", i32), ("x", fp), ("y", vp), ("ln_g", fp), ("ln_b", fp), ("w", vp), ("bias", fp),
        ("x_out", fp), ("out", vp), ("alpha", C.c_float), ("eps", C.c_float), ("stream", vp),
    ]


class CnnFrontArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("T", i32), ("F", i32), ("C1", i32), ("C2", i32), ("pad_", i32),
        ("feats", fp), ("w1", fp), ("b1", fp), ("ln1_g", fp), ("ln1_b", fp), ("w2", vp), ("b2", fp), ("ln2_g", fp), ("ln2_b", fp),
        ("eps1", C.c_float), ("eps2", C.c_float), ("slope", C.c_float), ("pad2_", i32), ("out", vp), ("stream", vp),
    ]


class AddLnArgs(C.Structure):
    _fields_ = [
        ("rows", i64), ("dim", i32), ("y_dtype", i32), ("out_dtype", i32), ("out_act", i32),
        ("x", fp), ("y", vp), ("alpha", C.c_float), ("eps1", C.c_float), ("eps2", C.c_float), ("pad2_", i32),
        ("g1", fp), ("b1", fp), ("g2", fp), ("b2", fp), ("x_out", fp), ("out", vp), ("stream", vp),
    ]


class GluDwconvArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("seqlen", i32), ("dim", i32), ("ksize", i32), ("io_dtype", i32), ("glu_done", i32),
        ("in_", vp), ("weight", fp), ("bias", fp), ("ln_g", fp), ("ln_b", fp), ("eps", C.c_float), ("variant", i32),
        ("out", vp), ("stream", vp), ("weight_t", fp), ("lin_w", vp), ("lin_b", fp),
    ]


class CnnBlock1Args(C.Structure):
    _fields_ = [
        ("batch", i32), ("T", i32), ("F", i32), ("C", i32), ("io_dtype", i32), ("pad_out", i32),
        ("feats", fp), ("weight", fp), ("bias", fp), ("ln_g", fp), ("ln_b", fp), ("eps", C.c_float), ("slope", C.c_float),
        ("out", vp), ("stream", vp),
    ]


class CnnBlock2Args(C.Structure):
    _fields_ = [
        ("batch", i32), ("T_in", i32), ("F_in", i32), ("C_in", i32), ("C_out", i32), ("pad_", i32),
        ("in_", vp), ("weight", vp), ("bias", fp), ("ln_g", fp), ("ln_b", fp), ("eps", C.c_float), ("slope", C.c_float),
        ("out", vp), ("stream", vp),
    ]


class GemmArgs(C.Structure):
    _fields_ = [
        ("M", i32), ("N", i32), ("K", i32), ("epilogue", i32),
        ("A", vp), ("lda", i64), ("W", vp), ("ldw", i64), ("bias", fp), ("out", vp), ("ldo", i64), ("x", fp),
        ("alpha", C.c_float), ("eps1", C.c_float), ("eps2", C.c_float), ("pad_", i32),
        ("g1", fp), ("b1", fp), ("g2", fp), ("b2", fp), ("stream", vp),
    ]


class FfnArgs(C.Structure):
    _fields_ = [
        ("rows", i32), ("dim", i32), ("hidden", i32), ("h_dtype", i32),
        ("x", fp), ("addend", vp), ("pre_g", fp), ("pre_b", fp), ("w1", vp), ("b1", fp), ("w2", vp), ("b2", fp),
        ("n1_g", fp), ("n1_b", fp), ("n2_g", fp), ("n2_b", fp), ("x_out", fp), ("h_out", vp),
        ("add_scale", C.c_float), ("alpha", C.c_float), ("pre_eps", C.c_float), ("n1_eps", C.c_float), ("n2_eps", C.c_float),
        ("proj_dim", i32), ("stream", vp), ("proj_w", vp), ("proj_b", fp), ("proj_out", vp),
        ("pre_out", vp), ("xn_out", vp), ("p1", C.c_float), ("p2", C.c_float), ("seed1", C.c_uint64), ("seed2", C.c_uint64),
        ("stats_out", fp), ("layout", i32), ("tokens", i32), ("seed_epoch", vp),
    ]


class FfnBwdArgs(C.Structure):
    _fields_ = [
        ("rows", i32), ("dim", i32), ("hidden", i32), ("reserved0", i32), ("dout", fp), ("w2t", vp), ("w1t", vp), ("pre", vp),
        ("da2", vp), ("da1", vp), ("act", vp), ("dh", vp), ("db1", fp), ("db2", fp),
        ("alpha", C.c_float), ("p1", C.c_float), ("p2", C.c_float), ("reserved1", C.c_float), ("seed1", C.c_uint64), ("seed2", C.c_uint64),
        ("workspace", fp), ("workspace_floats", i64), ("stream", vp), ("db1_part", fp), ("db2_part", fp), ("seed_epoch", vp),
    ]


class FbankArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("n_freq", i32), ("frames", i32), ("n_mels", i32),
        ("spec", fp), ("fbank", fp), ("db", fp), ("umax", fp), ("amin", C.c_float), ("top_db", C.c_float),
        ("mean", fp), ("std", fp), ("band_lo", vp), ("band_hi", vp), ("band_off", vp), ("band_w", fp), ("spec_bs", i64), ("spec_fs", i64), ("spec_ts", i64),
        ("stream", vp), ("umax_part", fp),
        ("wav", fp), ("window", fp), ("twiddle", fp), ("wav_bs", i64), ("samples", i32), ("hop", i32), ("n_fft", i32), ("pad3_", i32),
    ]


class SpecDropArgs(C.Structure):
    _fields_ = [
        ("batch", i32), ("frames", i32), ("n_mels", i32), ("n_masks", i32), ("dim", i32), ("pad_", i32),
        ("feats", fp), ("start", vp), ("length", vp), ("fill", fp), ("stream", vp),
    ]


# every symbol include/conmamba_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("cm_abi_version", C.c_int, []),
    ("cm_last_error", C.c_char_p, []),
    ("cm_scan_num_chunks", C.c_int, [C.c_int]),
    ("cm_selective_scan_fwd", C.c_int, [C.POINTER(ScanFwdArgs)]),
    ("cm_selective_scan_bwd", C.c_int, [C.POINTER(ScanBwdArgs)]),
    ("cm_selective_scan_bwd_workspace_bytes", C.c_int64, [C.POINTER(ScanBwdArgs)]),
    ("cm_causal_conv1d_fwd", C.c_int, [C.POINTER(ConvArgs)]),
    ("cm_causal_conv1d_bwd", C.c_int, [C.POINTER(ConvArgs)]),
    ("cm_scan_cl_fwd", C.c_int, [C.POINTER(ScanClArgs)]),
    ("cm_scan_cl_fwd_workspace_bytes", C.c_int64, [C.POINTER(ScanClArgs)]),
    ("cm_scan_cl_fwd_auto_chunks", i32, [i32, i32, i32, i32]),
    ("cm_scan_cl_bwd_workspace_bytes", C.c_int64, [C.POINTER(ScanClBwdArgs)]),
    ("cm_scan_cl_bwd_auto_chunks", C.c_int, [C.c_int] * 4),
    ("cm_reflect_pad_tf", C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    ("cm_ffn_pack_weights32", C.c_int, [vp, i32, i32, vp, vp]),
    ("cm_ffn_bwd_workspace_floats", C.c_int64, [i32, i32]),
    ("cm_ffn_bwd_fused", C.c_int, [C.POINTER(FfnBwdArgs)]),
    ("cm_wgrad_supported", C.c_int, [i32, i32, i32]),
    ("cm_wgrad_workspace_floats", C.c_int64, [i32, i32, i32]),
    ("cm_wgrad_bf16", C.c_int, [C.POINTER(WgradArgs)]),
    ("cm_scan_cl_bwd", C.c_int, [C.POINTER(ScanClBwdArgs)]),
    ("cm_conv_cl_fwd", C.c_int, [C.POINTER(ConvClArgs)]),
    ("cm_sum_leading", C.c_int, [vp, vp, i32, i64, i32, i32, vp]),
    ("cm_ctc_workspace_floats", C.c_int64, [i32, i32, i32]),
    ("cm_ctc_loss", C.c_int, [C.POINTER(CtcArgs)]),
    ("cm_bias_act_dropout_bwd_workspace_floats", C.c_int64, [i64, i32]),
    ("cm_bias_act_dropout_fwd", C.c_int, [C.POINTER(FfnElemArgs)]),
    ("cm_bias_act_dropout_bwd", C.c_int, [C.POINTER(FfnElemArgs)]),
    ("cm_conv_cl_bwd_workspace_floats", C.c_int64, [i32, i32, i32]),
    ("cm_conv_cl_bwd", C.c_int, [C.POINTER(ConvClBwdArgs)]),
    ("cm_conv_xproj", C.c_int, [C.POINTER(ConvXprojArgs)]),
    ("cm_add_layernorm", C.c_int, [C.POINTER(AddLnArgs)]),
    ("cm_glu_dwconv_ln_gelu", C.c_int, [C.POINTER(GluDwconvArgs)]),
    ("cm_ln_pw_glu", C.c_int, [C.POINTER(LnPwGluArgs)]),
    ("cm_causal_conv1d_update", C.c_int, [C.POINTER(ConvUpdateArgs)]),
    ("cm_selective_state_update", C.c_int, [C.POINTER(StateUpdateArgs)]),
    ("cm_layernorm_bwd_workspace_floats", C.c_int64, [i64, i32]),
    ("cm_layernorm_fwd", C.c_int, [C.POINTER(LayerNormArgs)]),
    ("cm_layernorm_bwd", C.c_int, [C.POINTER(LayerNormArgs)]),
    ("cm_dwconv_cl_workspace_floats", C.c_int64, [i32, i32, i32]),
    ("cm_dwconv_cl_fwd", C.c_int, [C.POINTER(DwconvClArgs)]),
    ("cm_dwconv_cl_bwd", C.c_int, [C.POINTER(DwconvClArgs)]),
    ("cm_dwconv1d_fwd", C.c_int, [C.POINTER(Dwconv1dArgs)]),
    ("cm_dwconv1d_bwd", C.c_int, [C.POINTER(Dwconv1dArgs)]),
    ("cm_cnn_block1", C.c_int, [C.POINTER(CnnBlock1Args)]),
    ("cm_cnn_block2", C.c_int, [C.POINTER(CnnBlock2Args)]),
    ("cm_cnn_front", C.c_int, [C.POINTER(CnnFrontArgs)]),
    ("cm_gemm_bf16", C.c_int, [C.POINTER(GemmArgs)]),
    ("cm_ffn_fused", C.c_int, [C.POINTER(FfnArgs)]),
    ("cm_ffn_pack_weights", C.c_int, [vp, i32, i32, vp, vp]),
    ("cm_fbank_mel_db", C.c_int, [C.POINTER(FbankArgs)]),
    ("cm_fbank_finish", C.c_int, [C.POINTER(FbankArgs)]),
    ("cm_fbank_wav", C.c_int, [C.POINTER(FbankArgs)]),
    ("cm_spec_drop", C.c_int, [C.POINTER(SpecDropArgs)]),
]

_lib = None


def lib():
    """dlopen the HIP library (once).  Raises RuntimeError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C mamba_asr_amd/csrc` (hipcc, --offload-arch=gfx950). There is no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(handle, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        got = handle.cm_abi_version()
        if got != ABI_VERSION:
            raise RuntimeError(f"libconmamba_hip ABI {got} != binding {ABI_VERSION}: rebuild the library")
        if hasattr(handle, "cm_debug_set"):  # the ablation build (make ablate; CM_LIB_PATH): timing-only kernel variants
            handle.cm_debug_set.restype, handle.cm_debug_set.argtypes = C.c_int, [C.c_int]
            handle.cm_debug_get.restype, handle.cm_debug_get.argtypes = C.c_int, []
            if os.environ.get("CM_DEBUG"):
                handle.cm_debug_set(int(os.environ["CM_DEBUG"]))
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().cm_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
