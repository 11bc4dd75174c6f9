"""Host-side logic of the mst package (no device work)."""
import os
import re

import pytest
import torch

from mst import _cabi, _desc
from mst.modules import AdvancedMixConsole, BasicMixConsole, _LazyParamDict, denormalize, denormalize_parameters, normalize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_console_attributes_match_the_reference_surface():
    from oracle import console_restated as oc

    c = AdvancedMixConsole(sample_rate=44100, input_min_gain_db=-48.0, input_max_gain_db=48.0, output_min_gain_db=-48.0,
                           output_max_gain_db=48.0, eq_min_gain_db=-12.0, eq_max_gain_db=12.0, min_pan=0.0, max_pan=1.0)
    assert c.param_ranges == oc.param_ranges(44100)  # itself pinned to the real class by make_golden.py
    assert (c.num_track_control_params, c.num_fx_bus_control_params, c.num_master_bus_control_params) == (27, 25, 26)
    assert c.sample_rate == 44100
    assert len(list(c.parameters())) == 0 and len(list(c.buffers())) == 0  # state-free like the reference
    c48 = AdvancedMixConsole(48000)
    assert c48.param_ranges["parametric_eq"]["band3_cutoff_freq"] == (12000, 23000)


def test_index_maps():
    assert _desc.TRACK_INDEX[0] == ("input_fader", "gain_db")
    assert _desc.TRACK_INDEX[1] == ("parametric_eq", "low_shelf_gain_db")
    assert _desc.TRACK_INDEX[18] == ("parametric_eq", "high_shelf_q_factor")
    assert _desc.TRACK_INDEX[19] == ("compressor", "threshold_db")
    assert _desc.TRACK_INDEX[25] == ("stereo_panner", "pan") and _desc.TRACK_INDEX[26] == ("fx_bus", "send_db")
    assert _desc.MASTER_INDEX[17] == ("parametric_eq", "high_shelf_q_factor")
    assert _desc.MASTER_INDEX[24] == ("output_fader", "gain_db") and _desc.MASTER_INDEX[25] == ("input_fader", "gain_db")
    assert _desc.FX_INDEX[12] == ("reverberation", "band0_decay") and _desc.FX_INDEX[24] == ("reverberation", "mix")


def test_flag_word_and_status_decoding():
    w = _desc.flag_word(use_fx_bus=False)
    assert w & _cabi.USE_TRACK_EQ and w & _cabi.USE_MASTER_BUS and not (w & _cabi.USE_FX_BUS) and not (w & _cabi.SAVE_FOR_BACKWARD)
    assert _desc.flag_word(save_for_backward=True) & _cabi.SAVE_FOR_BACKWARD
    assert _desc.status_to_error(0) is None
    assert str(_desc.status_to_error(1000 - (1 + 25))) == "Parameter pan of effect stereo_panner is out of range."
    assert str(_desc.status_to_error(1000 - (1 + 27 + 3))) == "Parameter band3_gain of effect reverberation is out of range."
    assert str(_desc.status_to_error(1000 - (1 + 52 + 24))) == "Parameter gain_db of effect output_fader is out of range."


def test_no_cpu_fallback_and_unsupported_paths():
    c = AdvancedMixConsole(44100)
    t = (torch.zeros(1, 2, 4096), torch.rand(1, 2, 27), torch.rand(1, 25), torch.rand(1, 26))
    with pytest.raises(RuntimeError, match="no CPU path"):
        c(*t, use_fx_bus=False)
    with pytest.raises(NotImplementedError, match="use_fx_bus"):
        c(*t)  # the reference's default flag; not built yet
    with pytest.raises(RuntimeError, match="shape-inconsistent"):
        c(*t, use_fx_bus=False, use_track_panner=False)
    with pytest.raises(ValueError):
        AdvancedMixConsole(44100, validate="never")
    from mst.loss import AudioFeatureLoss, MultiResolutionSTFTLoss
    from mst.utils import batch_stereo_peak_normalize

    with pytest.raises(RuntimeError, match="no CPU path"):
        MultiResolutionSTFTLoss()(torch.zeros(1, 2, 4096), torch.zeros(1, 2, 4096))
    with pytest.raises(RuntimeError, match="no CPU path"):
        AudioFeatureLoss([1.0] * 5, 44100)(torch.zeros(1, 2, 20000), torch.zeros(1, 2, 20000))
    with pytest.raises(RuntimeError, match="no CPU path"):
        batch_stereo_peak_normalize(torch.zeros(1, 2, 100))


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under diff-mst_amd/ may import or mention it."""
    pkg = os.path.join(ROOT, "diff-mst_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text, flags=re.M), f
                assert "hostsim" not in text, f


def test_denormalize_helpers_and_lazy_dicts():
    assert denormalize(0.25, 10.0, 2.0) == 4.0 and normalize(4.0, 2.0, 10.0) == 0.25
    c = AdvancedMixConsole(44100)
    tp, fp, mp = torch.rand(2, 3, 27), torch.rand(2, 25), torch.rand(2, 26)
    built = []
    tpd, fpd, mpd = c._denormalized_dicts(tp, fp, mp)
    assert isinstance(tpd, dict) and isinstance(tpd, _LazyParamDict)
    assert tpd._build is not None  # nothing computed until somebody looks
    assert set(tpd.keys()) == {"input_fader", "parametric_eq", "compressor", "stereo_panner", "fx_bus"}
    assert set(mpd) == {"parametric_eq", "compressor", "output_fader", "input_fader"}
    lo, hi = c.param_ranges["compressor"]["ratio"]
    assert torch.allclose(tpd["compressor"]["ratio"], tp[..., 20] * (hi - lo) + lo)
    assert torch.equal(fpd["reverberation"]["mix"], torch.ones(2))  # reference mst/modules.py:420
    assert tpd["parametric_eq"]["band1_cutoff_freq"].shape == (2, 3) and mpd["input_fader"]["gain_db"].shape == (2,)
    # the reference helper (with its 156 host checks) is kept for API parity
    with pytest.raises(ValueError, match="Parameter gain_db of effect input_fader is out of range."):
        denormalize_parameters({"input_fader": {"gain_db": torch.tensor([1.2])}}, c.param_ranges)
    del built


def test_normalize_dict_round_trip():
    c = AdvancedMixConsole(44100)
    tp = torch.rand(2, 3, 27)
    tpd, _, _ = c._denormalized_dicts(tp, torch.rand(2, 25), torch.rand(2, 26))
    back = c._normalize_dict(dict(tpd), _desc.TRACK_INDEX)
    assert torch.allclose(back, tp, atol=1e-5)


def test_basic_console_is_gain_and_pan_only():
    b = BasicMixConsole(44100, min_gain_db=-12.0, max_gain_db=12.0)
    assert b.param_ranges["input_fader"]["gain_db"] == (-12.0, 12.0)
    assert b.param_ranges["stereo_panner"]["pan"] == (0.0, 1.0)


def test_shard_batch():
    import bench

    assert [bench.shard_batch(64, r, 8) for r in (0, 3, 7)] == [(0, 8), (24, 32), (56, 64)]
    with pytest.raises(AssertionError):
        bench.shard_batch(10, 0, 4)
