"""ctypes binding of the C ABI declared in ``include/diffmst_hip.h``.

``bind(lib)`` only attaches argument / return types to an already opened shared
library; which library gets opened is decided elsewhere (``mst._hip`` opens the
gfx950 build and nothing else).
"""
from __future__ import annotations

import ctypes as C

NUM_TRACK_PARAMS = 27
NUM_FX_PARAMS = 25
NUM_MASTER_PARAMS = 26

USE_TRACK_INPUT_FADER = 0x01
USE_TRACK_EQ = 0x02
USE_TRACK_COMPRESSOR = 0x04
USE_TRACK_PANNER = 0x08
USE_FX_BUS = 0x10
USE_MASTER_BUS = 0x20
USE_OUTPUT_FADER = 0x40
SAVE_FOR_BACKWARD = 0x100
DEV_MULTIPASS_EQ = 0x200
NO_RANGE_CHECK = 0x400
BWD_PREPARED = 0x800
SPLIT_BATCH = 0x1000

ABI_VERSION = 9


class ConsoleDesc(C.Structure):
    _fields_ = [
        ("bs", C.c_int32),
        ("n_tracks", C.c_int32),
        ("n_samples", C.c_int64),
        ("track_row_stride", C.c_int64),
        ("sample_rate", C.c_float),
        ("flags", C.c_uint32),
        ("track_lookahead", C.c_int32),
        ("master_lookahead", C.c_int32),
        ("track_lo", C.c_float * NUM_TRACK_PARAMS),
        ("track_hi", C.c_float * NUM_TRACK_PARAMS),
        ("master_lo", C.c_float * NUM_MASTER_PARAMS),
        ("master_hi", C.c_float * NUM_MASTER_PARAMS),
        ("fx_lo", C.c_float * NUM_FX_PARAMS),
        ("fx_hi", C.c_float * NUM_FX_PARAMS),
        ("fx_ir_samples", C.c_int32),
        ("fx_bandpass_taps", C.c_int32),
    ]


class ConsoleFx(C.Structure):  # mirrors mst_console_fx
    _fields_ = [("noise", C.c_void_p), ("filters", C.c_void_p), ("tables", C.c_void_p)]


class ConsoleOverlap(C.Structure):  # mirrors mst_console_overlap: the side stream and the two events a split call borrows
    _fields_ = [("side_stream", C.c_void_p), ("fork_event", C.c_void_p), ("join_event", C.c_void_p)]


MAX_RESOLUTIONS = 8


class MrstftDesc(C.Structure):
    _fields_ = [
        ("rows", C.c_int32),
        ("n_samples", C.c_int64),
        ("n_res", C.c_int32),
        ("fft_size", C.c_int32 * MAX_RESOLUTIONS),
        ("hop_size", C.c_int32 * MAX_RESOLUTIONS),
        ("win_length", C.c_int32 * MAX_RESOLUTIONS),
        ("w_sc", C.c_float),
        ("w_log_mag", C.c_float),
        ("w_lin_mag", C.c_float),
        ("sc_per_example", C.c_int32),
        ("eps", C.c_float),
    ]


class Cnn14Desc(C.Structure):  # mirrors mst_cnn14_desc
    _fields_ = [("n", C.c_int32), ("frames", C.c_int32), ("bins", C.c_int32), ("embed_dim", C.c_int32), ("precision", C.c_int32),
                ("training", C.c_int32), ("bn_eps", C.c_float), ("world", C.c_int32)]


SYNC_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p)  # mst_sync_fn(user, sums, n_doubles, stream)


CNN14_CONVS = 12


class Cnn14Params(C.Structure):  # mirrors mst_cnn14_params
    _fields_ = [("conv_w", C.c_void_p * CNN14_CONVS), ("bn_gamma", C.c_void_p * CNN14_CONVS), ("bn_beta", C.c_void_p * CNN14_CONVS),
                ("bn_mean", C.c_void_p * CNN14_CONVS), ("bn_var", C.c_void_p * CNN14_CONVS), ("fc_w", C.c_void_p), ("fc_b", C.c_void_p)]


class Cnn14Grads(C.Structure):  # mirrors mst_cnn14_grads
    _fields_ = [("conv_w", C.c_void_p * CNN14_CONVS), ("bn_gamma", C.c_void_p * CNN14_CONVS), ("bn_beta", C.c_void_p * CNN14_CONVS),
                ("fc_w", C.c_void_p), ("fc_b", C.c_void_p)]


CTRL_FIELDS = ("in_proj_weight", "in_proj_bias", "out_proj_weight", "out_proj_bias", "linear1_weight", "linear1_bias",
               "linear2_weight", "linear2_bias", "norm1_weight", "norm1_bias", "norm2_weight", "norm2_bias")


class CtrlDesc(C.Structure):  # mirrors mst_ctrl_desc
    _fields_ = [("bs", C.c_int32), ("seq", C.c_int32), ("d_model", C.c_int32), ("nhead", C.c_int32), ("d_ff", C.c_int32),
                ("n_layers", C.c_int32), ("ln_eps", C.c_float)]


CTRL_IO_FIELDS = ("track_embedding", "mix_embedding", "fx_bus_embedding", "master_bus_embedding", "track_w", "track_b", "fx_w", "fx_b",
                  "master_w", "master_b")


class CtrlIO(C.Structure):  # mirrors mst_ctrl_io and mst_ctrl_io_grads (same ten pointers)
    _fields_ = [(name, C.c_void_p) for name in CTRL_IO_FIELDS]


class CtrlLayer(C.Structure):  # mirrors mst_ctrl_layer and mst_ctrl_layer_grads (same twelve pointers)
    _fields_ = [(name, C.c_void_p) for name in CTRL_FIELDS]


_P = C.c_void_p

SIGNATURES = {
    "mst_abi_version": (C.c_int, []),
    "mst_console_workspace_bytes": (C.c_size_t, [C.POINTER(ConsoleDesc)]),
    "mst_console_fx_tables_bytes": (C.c_size_t, []),
    "mst_console_fx_init_tables": (C.c_int, [_P, _P]),
    "mst_console_forward": (C.c_int, [C.POINTER(ConsoleDesc), _P, _P, _P, _P, C.POINTER(ConsoleFx), _P, _P, _P, _P, C.c_size_t, _P]),
    "mst_console_forward_mirrored": (C.c_int, [C.POINTER(ConsoleDesc), _P, _P, _P, _P, C.POINTER(ConsoleFx), _P, _P, _P, _P, C.c_size_t, _P, _P, _P]),
    "mst_console_backward": (C.c_int, [C.POINTER(ConsoleDesc), _P, _P, _P, _P, C.POINTER(ConsoleFx), _P, _P, _P, _P, _P, _P, _P, _P,
                                       C.c_size_t, _P]),
    "mst_console_backward_prepare": (C.c_int, [C.POINTER(ConsoleDesc), _P, C.c_size_t, _P]),
    "mst_console_forward_overlapped": (C.c_int, [C.POINTER(ConsoleDesc), _P, _P, _P, _P, C.POINTER(ConsoleFx), _P, _P, _P, _P, C.c_size_t, _P,
                                                 C.POINTER(ConsoleOverlap)]),
    "mst_console_backward_overlapped": (C.c_int, [C.POINTER(ConsoleDesc), _P, _P, _P, _P, C.POINTER(ConsoleFx), _P, _P, _P, _P, _P, _P, _P, _P,
                                                  C.c_size_t, _P, C.POINTER(ConsoleOverlap)]),
    "mst_mrstft_tables_bytes": (C.c_size_t, [C.POINTER(MrstftDesc)]),
    "mst_mrstft_init_tables": (C.c_int, [C.POINTER(MrstftDesc), _P, _P]),
    "mst_mrstft_workspace_bytes": (C.c_size_t, [C.POINTER(MrstftDesc)]),
    "mst_mrstft_forward": (C.c_int, [C.POINTER(MrstftDesc), _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "mst_mrstft_forward_eval": (C.c_int, [C.POINTER(MrstftDesc), _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "mst_mrstft_forward_partial": (C.c_int, [C.POINTER(MrstftDesc), _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "mst_mrstft_forward_finish": (C.c_int, [C.POINTER(MrstftDesc), _P, C.c_int32, _P, _P, C.c_size_t, _P]),
    "mst_mrstft_backward": (C.c_int, [C.POINTER(MrstftDesc), _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "mst_peak_normalize_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int64]),
    "mst_peak_normalize_forward": (C.c_int, [_P, _P, C.c_int32, C.c_int64, _P, C.c_size_t, _P]),
    "mst_peak_normalize_backward": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int64, _P, C.c_size_t, _P]),
    "mst_afloss_tables_bytes": (C.c_size_t, []),
    "mst_afloss_init_tables": (C.c_int, [_P, _P]),
    "mst_afloss_workspace_bytes": (C.c_size_t, [C.c_int32, C.c_int64]),
    "mst_afloss_forward": (C.c_int, [_P, _P, C.c_int32, C.c_int64, C.POINTER(C.c_float), _P, _P, _P, _P, C.c_size_t, _P]),
    "mst_spectrogram_tables_bytes": (C.c_size_t, []),
    "mst_spectrogram_init_tables": (C.c_int, [_P, _P]),
    "mst_spectrogram_forward": (C.c_int, [_P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, _P, _P]),
    "mst_cnn14_workspace_bytes": (C.c_size_t, [C.POINTER(Cnn14Desc)]),
    "mst_cnn14_forward": (C.c_int, [C.POINTER(Cnn14Desc), _P, C.POINTER(Cnn14Params), _P, _P, _P, C.c_size_t, _P]),
    "mst_cnn14_backward": (C.c_int, [C.POINTER(Cnn14Desc), _P, C.POINTER(Cnn14Params), _P, C.POINTER(Cnn14Grads), _P, C.c_size_t, _P]),
    "mst_cnn14_forward_sync": (C.c_int, [C.POINTER(Cnn14Desc), _P, C.POINTER(Cnn14Params), _P, _P, _P, C.c_size_t, _P, SYNC_FN, _P]),
    "mst_cnn14_backward_sync": (C.c_int, [C.POINTER(Cnn14Desc), _P, C.POINTER(Cnn14Params), _P, C.POINTER(Cnn14Grads), _P, C.c_size_t, _P,
                                          SYNC_FN, _P]),
    "mst_afloss_backward": (C.c_int, [_P, _P, C.c_int32, C.c_int64, C.POINTER(C.c_float), _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "mst_ctrl_workspace_bytes": (C.c_size_t, [C.POINTER(CtrlDesc)]),
    "mst_ctrl_forward": (C.c_int, [C.POINTER(CtrlDesc), _P, _P, C.POINTER(CtrlLayer), _P, _P, C.c_size_t, _P]),
    "mst_ctrl_backward": (C.c_int, [C.POINTER(CtrlDesc), _P, C.POINTER(CtrlLayer), _P, C.POINTER(CtrlLayer), _P, _P, C.c_size_t, _P]),
    "mst_ctrl_tokens_forward": (C.c_int, [C.POINTER(CtrlDesc), C.c_int32, _P, _P, _P, C.POINTER(CtrlIO), _P, _P, _P]),
    "mst_ctrl_heads_forward": (C.c_int, [C.POINTER(CtrlDesc), C.c_int32, _P, C.POINTER(CtrlIO), C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "mst_ctrl_heads_scratch_bytes": (C.c_size_t, [C.POINTER(CtrlDesc), C.c_int32]),
    "mst_ctrl_heads_backward": (C.c_int, [C.POINTER(CtrlDesc), C.c_int32, _P, C.POINTER(CtrlIO), C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P,
                                          C.POINTER(CtrlIO), _P, _P, _P]),
    "mst_ctrl_tokens_backward": (C.c_int, [C.POINTER(CtrlDesc), C.c_int32, _P, C.POINTER(CtrlIO), _P]),
}


def bind(lib: C.CDLL) -> C.CDLL:
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.mst_abi_version() != ABI_VERSION:
        raise RuntimeError(f"diffmst ABI mismatch: library {lib.mst_abi_version()} != binding {ABI_VERSION}")
    return lib


def ptr(t):
    """Device (or host, in the simulator tests) address of a tensor, or NULL for None."""
    return None if t is None else C.c_void_p(t.data_ptr())
