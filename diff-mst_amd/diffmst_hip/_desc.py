"""Build the C-ABI console descriptor from the console's ``param_ranges`` (host logic, no device work)."""
from __future__ import annotations

from . import _cabi

EQ_BANDS = ("low_shelf", "band0", "band1", "band2", "band3", "high_shelf")
EQ_NAMES = tuple(f"{b}_{w}" for b in EQ_BANDS for w in ("gain_db", "cutoff_freq", "q_factor"))
COMP_NAMES = ("threshold_db", "ratio", "attack_ms", "release_ms", "knee_db", "makeup_gain_db")

# (effect, parameter) of every column of the three parameter tensors - reference mst/modules.py:353-460
TRACK_INDEX = (
    (("input_fader", "gain_db"),)
    + tuple(("parametric_eq", n) for n in EQ_NAMES)
    + tuple(("compressor", n) for n in COMP_NAMES)
    + (("stereo_panner", "pan"), ("fx_bus", "send_db"))
)
FX_INDEX = (
    tuple(("reverberation", f"band{i}_gain") for i in range(12))
    + tuple(("reverberation", f"band{i}_decay") for i in range(12))
    + (("reverberation", "mix"),)
)
MASTER_INDEX = (
    tuple(("parametric_eq", n) for n in EQ_NAMES)
    + tuple(("compressor", n) for n in COMP_NAMES)
    + (("output_fader", "gain_db"), ("input_fader", "gain_db"))
)
assert len(TRACK_INDEX) == 27 and len(FX_INDEX) == 25 and len(MASTER_INDEX) == 26

FLAG_BITS = {
    "use_track_input_fader": _cabi.USE_TRACK_INPUT_FADER,
    "use_track_eq": _cabi.USE_TRACK_EQ,
    "use_track_compressor": _cabi.USE_TRACK_COMPRESSOR,
    "use_track_panner": _cabi.USE_TRACK_PANNER,
    "use_fx_bus": _cabi.USE_FX_BUS,
    "use_master_bus": _cabi.USE_MASTER_BUS,
    "use_output_fader": _cabi.USE_OUTPUT_FADER,
}


EXCHANGE_TIMEOUT = 2000

_IDENTITY = ([0.0] * 27, [1.0] * 27, [0.0] * 26, [1.0] * 26, [0.0] * 25, [1.0] * 25)


def range_vectors(param_ranges: dict, index):
    lo = [float(param_ranges[e][p][0]) for e, p in index]
    hi = [float(param_ranges[e][p][1]) for e, p in index]
    return lo, hi


def flag_word(save_for_backward: bool = False, multipass_eq: bool = False, split_batch: bool = False, **flags) -> int:
    word = 0
    for name, bit in FLAG_BITS.items():
        if flags.get(name, True):
            word |= bit
    if save_for_backward:
        word |= _cabi.SAVE_FOR_BACKWARD
    if multipass_eq:  # test switch (include/diffmst_hip.h MST_DEV_MULTIPASS_EQ): console._multipass_eq = True
        word |= _cabi.DEV_MULTIPASS_EQ
    if split_batch:  # include/diffmst_hip.h MST_SPLIT_BATCH: two halves of the batch on two streams (mst_console_*_overlapped)
        word |= _cabi.SPLIT_BATCH
    return word


def make_desc(param_ranges, sample_rate, bs, n_tracks, n_samples, track_row_stride, flags_word,
              track_lookahead=2048, master_lookahead=1024, identity_ranges=False,
              fx_ir_samples=65536, fx_bandpass_taps=1023) -> _cabi.ConsoleDesc:
    """identity_ranges: lo = 0, hi = 1 for every parameter, i.e. v*(hi-lo)+lo == v exactly - the descriptor of a call
    whose parameter tensors already hold DENORMALISED values (forward_mix_console)."""
    d = _cabi.ConsoleDesc()
    d.bs, d.n_tracks, d.n_samples = int(bs), int(n_tracks), int(n_samples)
    d.track_row_stride = int(track_row_stride)
    d.sample_rate = float(sample_rate)
    d.flags = int(flags_word)
    d.track_lookahead, d.master_lookahead = int(track_lookahead), int(master_lookahead)
    if identity_ranges:
        tlo, thi, mlo, mhi, flo, fhi = _IDENTITY
    else:  # read on every call, like the reference does (a caller may edit console.param_ranges between calls)
        tlo, thi = range_vectors(param_ranges, TRACK_INDEX)
        mlo, mhi = range_vectors(param_ranges, MASTER_INDEX)
        flo, fhi = range_vectors(param_ranges, FX_INDEX)
    d.fx_lo[:], d.fx_hi[:] = flo, fhi  # slice assignment: one call per array instead of one per element
    d.fx_ir_samples, d.fx_bandpass_taps = int(fx_ir_samples), int(fx_bandpass_taps)  # reference mst/modules.py:281-282
    d.track_lo[:], d.track_hi[:] = tlo, thi
    d.master_lo[:], d.master_hi[:] = mlo, mhi
    return d


def status_to_error(status: int):
    """Decode the device status word into the reference's ValueError (mst/modules.py:86-89)."""
    if status == 0:
        return None
    if status >= EXCHANGE_TIMEOUT:  # include/diffmst_hip.h: MST_STATUS_EXCHANGE_TIMEOUT
        return RuntimeError(
            "diffmst_hip: an in-launch aggregate exchange timed out (status %d); the outputs / gradients of that call are poisoned "
            "and must not be used" % status)
    code = 1000 - status - 1
    if code < 27:
        effect, name = TRACK_INDEX[code]
    elif code < 52:
        effect, name = FX_INDEX[code - 27]
    else:
        effect, name = MASTER_INDEX[code - 52]
    return ValueError(f"Parameter {name} of effect {effect} is out of range.")
