"""the training-step caller.  ``diffmst_hip.system.CommonStep`` (the call order of the reference's
``System.common_step``, mst/system.py:102-407, without Lightning) on the HIP console / losses against
  * the fixture produced by the REAL ``System.common_step`` in the build container (tests/golden/make_golden.py system),
  * the oracle driven in the same order on the GPU box's host."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import StubModel, rel

RES = dict(fft_sizes=[512, 2048, 8192], hop_sizes=[256, 1024, 4096], win_lengths=[512, 2048, 8192])


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from mst import _hip

    _hip.lib()
    return torch.device("cuda:0")


def make_step(dev, seed_model, **kw):
    from mst.loss import MultiResolutionSTFTLoss
    from mst.mixing import naive_random_mix
    from mst.modules import AdvancedMixConsole
    from mst.system import CommonStep

    model = StubModel(seed=seed_model).to(dev)
    step = CommonStep(model, AdvancedMixConsole(44100), naive_random_mix, MultiResolutionSTFTLoss(**RES), generate_mix=True,
                      active_eq_epoch=0, active_compressor_epoch=0, active_fx_bus_epoch=1000, active_master_bus_epoch=0, **kw)
    return step, model


def test_common_step_against_the_real_system(dev, golden_dir, record):
    g = np.load(os.path.join(golden_dir, "system_step.npz"))
    bs, T, n = (int(v) for v in g["shape"])
    torch.manual_seed(int(g["seed_tracks"]))
    tracks = (0.1 * torch.randn(bs, T, n)).half().float()
    assert np.array_equal(tracks.numpy()[..., ::1024], g["tracks_sub"])  # same seeded input as the generator
    step, model = make_step(dev, int(g["seed_model"]))
    batch = (tracks.to(dev), None, None, torch.zeros(bs, T, dtype=torch.bool, device=dev), None, ["a", "b"])
    torch.manual_seed(int(g["seed_mix"]))
    loss, data = step(batch, train=True, collect=True)
    loss.backward()
    t = lambda k: torch.from_numpy(g[k])
    rep = dict(
        loss=abs(loss.item() - float(g["loss"])) / float(g["loss"]),
        ref_mix_a=rel(data["ref_mix_a"][..., ::16], t("ref_mix_a_sub")), ref_mix_b=rel(data["ref_mix_b_norm"][..., ::16], t("ref_mix_b_sub")),
        pred_mix_b=rel(data["pred_mix_b_norm"][..., ::16], t("pred_mix_b_sub")), sum_mix_b=rel(data["sum_mix_b"][..., ::16], t("sum_mix_b_sub")),
        g_w_track=rel(model.w_track.grad, t("g_w_track")), g_w_master=rel(model.w_master.grad, t("g_w_master")),
    )
    print("\n[common_step vs real System]", rep)
    record(**rep)
    assert rep["ref_mix_a"] < 1e-4 and rep["ref_mix_b"] < 1e-4 and rep["pred_mix_b"] < 1e-4 and rep["sum_mix_b"] < 1e-6
    assert rep["loss"] < 1e-4
    assert rep["g_w_track"] < 1e-2 and rep["g_w_master"] < 1e-2
    assert model.w_fx.grad is None or float(model.w_fx.grad.abs().max()) == 0.0  # fx bus off: no gradient, as in the fixture
    # the dictionaries System's callbacks read (mst/callbacks/audio.py:92-125)
    assert torch.allclose(data["pred_track_param_dict"]["compressor"]["ratio"].detach().cpu(), t("pred_track_ratio"), rtol=1e-5)
    assert torch.allclose(data["ref_master_bus_param_dict"]["compressor"]["threshold_db"].detach().cpu(), t("ref_master_thr"), rtol=1e-6)


def test_common_step_against_the_oracle_in_the_same_order(dev, record):
    """Every stage of the step restated with the oracle (float64), same RNG stream, strided tracks[..., mid:] view."""
    from oracle import console_restated as oc
    from oracle import loss_restated as ol

    bs, T, n = 2, 3, 131072
    torch.manual_seed(61)
    tracks = 0.1 * torch.randn(bs, T, n)
    step, model = make_step(dev, 9, repeat_reference_mix=True)
    batch = (tracks.to(dev), None, None, None, None, None)
    torch.manual_seed(62)
    loss, data = step(batch, train=True)
    loss.backward()

    torch.manual_seed(62)
    flags = dict(use_track_input_fader=False, use_track_eq=True, use_track_compressor=True, use_track_panner=True,
                 use_fx_bus=False, use_master_bus=True, use_output_fader=True)  # naive_random_mix swallows use_output_fader=False
    for _ in range(2):  # the reference draws (and mixes) twice, keeping the second (mst/system.py:149-173, :222-246)
        p = [torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)]
    with torch.no_grad():
        _, ref, *_ = oc.console_forward(tracks.double(), p[0].double(), p[1].double(), p[2].double(), **flags)
        ref = oc.batch_stereo_peak_normalize(ref)
    mid = n // 2
    omodel = StubModel(seed=9).double()
    tp, fp, mp = omodel(tracks.double()[..., mid:], ref[..., :mid])
    _, pred, *_ = oc.console_forward(tracks.double()[..., mid:], tp, fp, mp, **dict(flags, use_track_input_fader=True))
    res = tuple(zip(RES["fft_sizes"], RES["hop_sizes"], RES["win_lengths"]))
    oloss = ol.mrstft_loss(pred, ref[..., mid:], res)
    oloss.backward()
    rep = dict(loss=abs(loss.item() - oloss.item()) / oloss.item(), g_w_track=rel(model.w_track.grad, omodel.w_track.grad),
               g_w_master=rel(model.w_master.grad, omodel.w_master.grad))
    print("\n[common_step vs oracle]", rep)
    record(**rep)
    assert rep["loss"] < 1e-4 and rep["g_w_track"] < 1e-2 and rep["g_w_master"] < 1e-2
