"""The oracle against the golden vectors produced by the REAL reference (tests/golden/make_golden.py).

CPU only.  These pin the mst-owned logic of the oracle (parameter maps, op order, flags, losses,
filterbank) to the reference's own code, independently of the container the fixtures were made in."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import console_restated as oc
from oracle import loss_restated as ol

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def parse_flags(arr):
    return {k: v == "True" for k, v in arr}


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "console_*.npz"))))
def test_console_fixture(path):
    g = np.load(path, allow_pickle=True)
    if g["tracks"].shape[-1] > 40000 and os.environ.get("MST_FAST_TESTS"):
        pytest.skip("long fixture")
    t = lambda k: torch.from_numpy(g[k]).float()
    flags = parse_flags(g["flags"])
    tp = t("track_params").requires_grad_(True)
    mp = t("master_bus_params").requires_grad_(True)
    mixed, mix, tpd, _, mpd = oc.console_forward(t("tracks"), tp, t("fx_bus_params"), mp, **flags)
    stride = int(g["mix_stride"])
    assert torch.allclose(mix[..., ::stride], t("mix"), rtol=1e-4, atol=1e-6 * float(t("mix").abs().max()))
    assert torch.allclose(mixed[..., ::64], t("mixed_tracks_sub"), rtol=1e-4, atol=1e-6 * float(mixed.abs().max()))
    (mix * t("grad_mix")).sum().backward()
    if np.abs(g["grad_track_params"]).max() > 0:
        ref = t("grad_track_params")
        assert (tp.grad - ref).norm() / ref.norm() < 1e-3
    for k in g.files:  # denormalised dictionaries are an exact affine map
        if k.startswith("tp."):
            _, eff, name = k.split(".")
            assert torch.equal(tpd[eff][name].detach(), t(k)), k
        if k.startswith("mp."):
            _, eff, name = k.split(".")
            assert torch.equal(mpd[eff][name].detach(), t(k)), k


def test_bark_filterbank_fixture():
    g = np.load(os.path.join(GOLD, "bark_fb.npz"))
    fb = ol.bark_filterbank(16385, 20.0, 20000.0, 24, 44100)
    assert fb.shape == (16385, 24)
    assert np.array_equal(fb.sum(0).numpy(), g["col_sums"])
    assert np.array_equal(fb[::257].numpy(), g["row_sub"])
    assert np.array_equal(fb.argmax(0).numpy(), g["argmax"])
    assert int((fb > 0).sum()) == int(g["nnz"])


def test_product_filterbank_is_bit_identical():
    """mst.filter.barkscale_fbanks (the constant table the HIP loss uploads) == the reference's."""
    from mst.filter import barkscale_fbanks

    g = np.load(os.path.join(GOLD, "bark_fb.npz"))
    fb = barkscale_fbanks(16385, 20.0, 20000.0, 24, 44100)
    assert np.array_equal(fb.sum(0).numpy(), g["col_sums"])
    assert np.array_equal(fb[::257].numpy(), g["row_sub"])
    assert torch.equal(fb, ol.bark_filterbank(16385, 20.0, 20000.0, 24, 44100))


def test_af_loss_fixture():
    g = np.load(os.path.join(GOLD, "af_loss.npz"))
    x = torch.from_numpy(g["input"]).requires_grad_(True)
    y = torch.from_numpy(g["target"])
    ld = ol.audio_feature_loss(x, y, list(g["weights"]))
    for k in ol.AF_KEYS:
        assert abs(ld[k].item() - float(g["loss." + k])) <= 1e-5 * abs(float(g["loss." + k])), k
    assert torch.allclose(ol.feat_rms(x.detach()), torch.from_numpy(g["feat.rms"]), rtol=1e-6)
    assert torch.allclose(ol.feat_crest_factor(x.detach()), torch.from_numpy(g["feat.crest"]), rtol=1e-5)
    assert torch.allclose(ol.feat_stereo_width(x.detach()), torch.from_numpy(g["feat.width"]), rtol=1e-5)
    assert torch.allclose(ol.feat_stereo_imbalance(x.detach()), torch.from_numpy(g["feat.imbalance"]), rtol=1e-5, atol=1e-8)
    assert torch.allclose(ol.feat_barkspectrum(x.detach()), torch.from_numpy(g["feat.bark"]), rtol=1e-5, atol=1e-6)
    sum(v.mean() for v in ld.values()).backward()
    ref = torch.from_numpy(g["grad_input_sub"])
    assert (x.grad[..., ::16] - ref).norm() / ref.norm() < 1e-4


def test_naive_random_mix_rng_order():
    """mst.mixing.naive_random_mix draws its three parameter tensors in the reference's order
    (mixing.py:61-69) and forwards the reference's (misspelt) keyword - checked with a recording console."""
    from mst.mixing import naive_random_mix

    g = np.load(os.path.join(GOLD, "naive_random_mix.npz"))

    class Recorder(torch.nn.Module):
        num_track_control_params, num_fx_bus_control_params, num_master_bus_control_params = 27, 25, 26

        def forward(self, tracks, tp, fp, mp, **kw):
            self.kw = kw
            return "mixed", "mix", "a", "b", "c"

    rec = Recorder()
    tracks = torch.cat([torch.from_numpy(g["tracks"])] * 8, dim=-1)
    torch.manual_seed(int(g["seed"]))
    out = naive_random_mix(tracks, rec, use_fx_bus=False, use_output_fader=False)  # System's spelling lands in **kwargs
    assert len(out) == 8 and out[:5] == ("mixed", "mix", "a", "b", "c")
    assert torch.equal(out[5], torch.from_numpy(g["mix_params"]))
    assert torch.equal(out[6], torch.from_numpy(g["fx_bus_params"]))
    assert torch.equal(out[7], torch.from_numpy(g["master_bus_params"]))
    assert rec.kw["use_output_fader"] is True and rec.kw["use_fx_bus"] is False
