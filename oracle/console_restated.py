"""Restatement of the reference's mix-console orchestration (mst-owned logic).

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Follows
``/root/reference/mst/modules.py``:

* parameter ranges                          modules.py:121-181
* ``denormalize`` / range check             modules.py:71-72, 79-97
* index maps of the three parameter tensors modules.py:353-460
* op order, flags, look-ahead constants     modules.py:186-314
* ``naive_random_mix``                      /root/reference/mst/mixing.py:35-94
* ``batch_stereo_peak_normalize``           /root/reference/mst/utils.py:14-29

Pinned against the real modules by ``tests/golden/make_golden.py``.
The table-driven layout here is deliberately different from the reference's
literal dictionaries; the arithmetic is the same.
"""
from __future__ import annotations

import torch

from . import dasp_restated as dasp

EQ_NAMES = [
    f"{band}_{what}"
    for band in ("low_shelf", "band0", "band1", "band2", "band3", "high_shelf")
    for what in ("gain_db", "cutoff_freq", "q_factor")
]
COMP_NAMES = ["threshold_db", "ratio", "attack_ms", "release_ms", "knee_db", "makeup_gain_db"]

NUM_TRACK_PARAMS = 27
NUM_FX_PARAMS = 25
NUM_MASTER_PARAMS = 26
TRACK_LOOKAHEAD = 2048  # modules.py:250
MASTER_LOOKAHEAD = 1024  # modules.py:304


def param_ranges(
    sample_rate: float,
    input_min_gain_db=-48.0,
    input_max_gain_db=48.0,
    output_min_gain_db=-48.0,
    output_max_gain_db=48.0,
    min_send_db=-80.0,
    max_send_db=12.0,
    eq_min_gain_db=-12.0,
    eq_max_gain_db=12.0,
    min_pan=0.0,
    max_pan=1.0,
    reverb_min_band_gain=0.0,
    reverb_max_band_gain=1.0,
    reverb_min_band_decay=0.0,
    reverb_max_band_decay=1.0,
):
    top = (sample_rate // 2) - 1000
    freq = {
        "low_shelf": (20, 2000),
        "band0": (80, 2000),
        "band1": (2000, 8000),
        "band2": (8000, 12000),
        "band3": (12000, top),
        "high_shelf": (6000, top),
    }
    eq = {}
    for band, fr in freq.items():
        eq[f"{band}_gain_db"] = (eq_min_gain_db, eq_max_gain_db)
        eq[f"{band}_cutoff_freq"] = fr
        eq[f"{band}_q_factor"] = (0.1, 5.0)
    rev = {f"band{i}_gain": (reverb_min_band_gain, reverb_max_band_gain) for i in range(12)}
    rev.update({f"band{i}_decay": (reverb_min_band_decay, reverb_max_band_decay) for i in range(12)})
    rev["mix"] = (0.0, 1.0)
    return {
        "input_fader": {"gain_db": (input_min_gain_db, input_max_gain_db)},
        "output_fader": {"gain_db": (output_min_gain_db, output_max_gain_db)},
        "parametric_eq": eq,
        "compressor": {
            "threshold_db": (-60.0, 0.0),
            "ratio": (1.0, 10.0),
            "attack_ms": (5.0, 250.0),
            "release_ms": (10.0, 250.0),
            "knee_db": (3.0, 12.0),
            "makeup_gain_db": (0.0, 6.0),
        },
        "reverberation": rev,
        "fx_bus": {"send_db": (min_send_db, max_send_db)},
        "stereo_panner": {"pan": (min_pan, max_pan)},
    }


def split_track_params(p: torch.Tensor) -> dict:
    d = {"input_fader": {"gain_db": p[..., 0]}}
    d["parametric_eq"] = {name: p[..., 1 + i] for i, name in enumerate(EQ_NAMES)}
    d["compressor"] = {name: p[..., 19 + i] for i, name in enumerate(COMP_NAMES)}
    d["stereo_panner"] = {"pan": p[..., 25]}
    d["fx_bus"] = {"send_db": p[..., 26]}
    return d


def split_fx_params(p: torch.Tensor) -> dict:
    rev = {f"band{i}_gain": p[..., i] for i in range(12)}
    rev.update({f"band{i}_decay": p[..., 12 + i] for i in range(12)})
    rev["mix"] = torch.ones_like(p[..., 24])  # modules.py:420 - forced wet
    return {"reverberation": rev}


def split_master_params(p: torch.Tensor) -> dict:
    d = {"parametric_eq": {name: p[..., i] for i, name in enumerate(EQ_NAMES)}}
    d["compressor"] = {name: p[..., 18 + i] for i, name in enumerate(COMP_NAMES)}
    d["output_fader"] = {"gain_db": p[..., 24]}
    d["input_fader"] = {"gain_db": p[..., 25]}
    return d


def denormalize_parameters(param_dict: dict, ranges: dict) -> dict:
    out = {}
    for effect, params in param_dict.items():
        out[effect] = {}
        for name, t in params.items():
            if t.min() < 0 or t.max() > 1:
                raise ValueError(f"Parameter {name} of effect {effect} is out of range.")
            lo, hi = ranges[effect][name]
            out[effect][name] = t * (hi - lo) + lo
    return out


def console_chain(
    tracks: torch.Tensor,
    tp: dict,
    mp: dict,
    sample_rate: float,
    use_track_input_fader=True,
    use_track_eq=True,
    use_track_compressor=True,
    use_track_panner=True,
    use_fx_bus=False,
    use_master_bus=True,
    use_output_fader=True,
    time_domain: bool = False,
    fp: dict = None,
    fx_noise=None,
    fx_ir_samples: int = 65536,
    fx_bandpass_taps: int = 1023,
):
    """Denormalised dicts in, (mixed_tracks (bs,2,T,n), mix (bs,2,n)) out."""
    bs, n_tracks, n = tracks.shape
    rows = tracks.reshape(bs * n_tracks, 1, n)
    if use_track_input_fader:
        rows = dasp.gain(rows, sample_rate, **tp["input_fader"])
    if use_track_eq:
        rows = dasp.parametric_eq(rows, sample_rate, time_domain=time_domain, **tp["parametric_eq"])
    if use_track_compressor:
        rows = dasp.compressor(
            rows, sample_rate, **tp["compressor"], lookahead_samples=TRACK_LOOKAHEAD, time_domain=time_domain
        )
    rows = rows.view(bs, n_tracks, n)
    if not use_track_panner:
        raise RuntimeError("reference non-panner branch is shape-inconsistent (modules.py:269)")
    mixed = dasp.stereo_panner(rows, sample_rate, **tp["stereo_panner"])
    bus = mixed.sum(dim=2)
    if use_fx_bus:  # modules.py:275-284
        fx_bus = dasp.stereo_bus(mixed, sample_rate, **tp["fx_bus"])
        fx_bus = dasp.noise_shaped_reverberation(fx_bus, sample_rate, **fp["reverberation"], num_samples=fx_ir_samples,
                                                 num_bandpass_taps=fx_bandpass_taps, noise=fx_noise)
        bus = bus + fx_bus
    if use_master_bus:
        bus = dasp.gain(bus, sample_rate, **mp["input_fader"])
        bus = dasp.parametric_eq(bus, sample_rate, time_domain=time_domain, **mp["parametric_eq"])
        bus = dasp.compressor(
            bus, sample_rate, **mp["compressor"], lookahead_samples=MASTER_LOOKAHEAD, time_domain=time_domain
        )
    if use_output_fader:
        bus = dasp.gain(bus, sample_rate, **mp["output_fader"])
    return mixed, bus


def console_forward(
    tracks, track_params, fx_bus_params, master_bus_params, sample_rate=44100, ranges=None, **flags
):
    """Normalised (0,1) tensors in; same 5-tuple as AdvancedMixConsole.forward (modules.py:481-487)."""
    ranges = ranges or param_ranges(sample_rate)
    tp = denormalize_parameters(split_track_params(track_params), ranges)
    fp = denormalize_parameters(split_fx_params(fx_bus_params), ranges)
    mp = denormalize_parameters(split_master_params(master_bus_params), ranges)
    mixed, mix = console_chain(tracks, tp, mp, sample_rate, fp=fp, **flags)
    return mixed, mix, tp, fp, mp


def batch_stereo_peak_normalize(x: torch.Tensor) -> torch.Tensor:
    peak = x.abs().amax(dim=(-1, -2), keepdim=True)
    return x / peak.clamp(1e-8)
