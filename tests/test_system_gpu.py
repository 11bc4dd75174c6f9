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


def test_cfg5_step_with_the_encoder_model(dev, record):
    """BASELINE cfg #5 on one GPU at batch 1: the full training step - naive_random_mix reference, the REAL model structure
    (SpectrogramEncoder + Cnn14 on MFMA for 32 tracks + 2 mix channels, TransformerController), AdvancedMixConsole with 32 tracks,
    AudioFeatureLoss (weights of configs/models/unpaired+feat.yaml:55-60) - forward and backward through every part.
    Parity of the new link (audio -> estimated parameters) against the oracle's encoder + the same controller on the host; the
    console and the loss are pinned by their own tests."""
    from mst.loss import AudioFeatureLoss
    from mst.mixing import naive_random_mix
    from mst.modules import AdvancedMixConsole, MixStyleTransferModel, SpectrogramEncoder, TransformerController
    from mst.system import CommonStep
    from oracle import encoder_restated as oe

    bs, T, n = 1, 32, 262144
    torch.manual_seed(100)
    model = MixStyleTransferModel(SpectrogramEncoder(embed_dim=64, precision="fp32"), SpectrogramEncoder(embed_dim=64, precision="fp32"),
                                  TransformerController(64, 27, 25, 26, num_layers=2, nhead=4))
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    step = CommonStep(model, AdvancedMixConsole(44100), naive_random_mix, AudioFeatureLoss([0.1, 0.001, 1.0, 1.0, 0.1], 44100), generate_mix=True,
                      active_eq_epoch=0, active_compressor_epoch=0, active_fx_bus_epoch=1000, active_master_bus_epoch=0)
    torch.manual_seed(101)
    tracks = 0.05 * torch.randn(bs, T, n)
    batch = (tracks.to(dev), None, None, torch.zeros(bs, T, dtype=torch.bool, device=dev), None, ["a"])
    torch.manual_seed(102)
    loss, data = step(batch, train=True, collect=True)
    loss.backward()
    assert torch.isfinite(loss) and set(data["loss_terms"]) == {"mix-rms", "mix-crest_factor", "mix-stereo_width", "mix-stereo_imbalance", "mix-barkspectrum"}
    missing = [k for k, p in model.named_parameters() if p.grad is None and "fx_bus_projection" not in k]
    assert not missing, missing  # fx bus off: its projection gets no gradient, everything else does
    assert all(torch.isfinite(p.grad).all() for p in model.parameters() if p.grad is not None)
    assert float(model.track_encoder.model.conv_block1.conv1.weight.grad.abs().max()) > 0
    assert data["pred_track_param_dict"]["stereo_panner"]["pan"].shape == (bs, T)

    # audio -> parameters: the oracle's encoders (fp32, training mode: batch statistics over the 32 / 2 signals) + the same controller
    mid = n // 2
    ref_mix_a = data["ref_mix_a"]
    host = MixStyleTransferModel(torch.nn.Identity(), torch.nn.Identity(), TransformerController(64, 27, 25, 26, num_layers=2, nhead=4)).train()
    host.controller.load_state_dict({k[len("controller."):]: v for k, v in sd0.items() if k.startswith("controller.")})
    with torch.no_grad():
        te = oe.spectrogram_encoder(tracks[..., mid:].reshape(bs * T, 1, -1), {k[len("track_encoder."):]: v.clone() for k, v in sd0.items()
                                                                               if k.startswith("track_encoder.")}, training=True)
        me = oe.spectrogram_encoder(ref_mix_a.reshape(bs * 2, 1, -1), {k[len("mix_encoder."):]: v.clone() for k, v in sd0.items()
                                                                       if k.startswith("mix_encoder.")}, training=True)
        o_tp, o_fp, o_mp = host.controller(te.view(bs, T, -1), me.view(bs, 2, -1), torch.zeros(bs, T, dtype=torch.bool))
    model.load_state_dict(sd0)  # the step above moved the running statistics
    model.train()
    with torch.no_grad():
        h_tp, h_fp, h_mp = model(tracks.to(dev)[..., mid:], ref_mix_a.to(dev), track_padding_mask=torch.zeros(bs, T, dtype=torch.bool, device=dev))
    rep = dict(track_params=rel(h_tp, o_tp), fx_params=rel(h_fp, o_fp), master_params=rel(h_mp, o_mp), loss=float(loss.detach()))
    print("\n[cfg #5 step, batch 1] estimated parameters vs oracle encoder + controller:", rep)
    record(**rep)
    assert rep["track_params"] < 1e-3 and rep["fx_params"] < 1e-3 and rep["master_params"] < 1e-3


def _oracle_cfg5_step(tracks, sd0, ref_params, dtype, device, emulate_bf16=False):
    """The cfg #5 step restated with the oracle in `dtype` on `device`: naive_random_mix reference (given draws) -> peak normalise ->
    A/B split -> oracle encoders (training-mode BatchNorm) -> a torch TransformerController with the same weights -> console_restated ->
    audio_feature_loss.  Returns (loss, {parameter name: gradient}) on the CPU."""
    from mst.modules import TransformerController
    from oracle import console_restated as oc
    from oracle import encoder_restated as oe
    from oracle import loss_restated as ol

    bs, T, n = tracks.shape
    mid = n // 2
    tr = tracks.to(device=device, dtype=dtype)
    flags = dict(use_track_input_fader=False, use_track_eq=True, use_track_compressor=True, use_track_panner=True,
                 use_fx_bus=False, use_master_bus=True, use_output_fader=True)  # naive_random_mix swallows use_output_fader=False
    with torch.no_grad():
        _, ref, *_ = oc.console_forward(tr, *(p.to(device=device, dtype=dtype) for p in ref_params), **flags)
        ref = oc.batch_stereo_peak_normalize(ref)
    sd = {k: (v.to(device=device, dtype=dtype).requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "window"
              else v.to(device)) for k, v in sd0.items() if not k.startswith("controller.")}
    sd = {k: (v.to(dtype) if v.is_floating_point() and not v.requires_grad else v) for k, v in sd.items()}
    ctrl = TransformerController(512, 27, 25, 26, num_layers=12, nhead=8).to(device=device, dtype=dtype).train()
    ctrl.load_state_dict({k[len("controller."):]: v for k, v in sd0.items() if k.startswith("controller.")})
    sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
    te = oe.spectrogram_encoder(tr[..., mid:].reshape(bs * T, 1, -1), sub("track_encoder."), training=True, emulate_bf16=emulate_bf16)
    me = oe.spectrogram_encoder(ref[..., :mid].reshape(bs * 2, 1, -1), sub("mix_encoder."), training=True, emulate_bf16=emulate_bf16)
    tp, fp, mp = ctrl(te.view(bs, T, -1), me.view(bs, 2, -1), torch.zeros(bs, T, dtype=torch.bool, device=device))
    _, pred, *_ = oc.console_forward(tr[..., mid:], tp, fp, mp, **dict(flags, use_track_input_fader=True))
    loss = sum(v.mean() for v in ol.audio_feature_loss(pred, ref[..., mid:], [0.1, 0.001, 1.0, 1.0, 0.1]).values())
    loss.backward()
    grads = {k: v.grad.detach().double().cpu() for k, v in sd.items() if v.requires_grad and v.grad is not None}
    grads.update({"controller." + k: p.grad.detach().double().cpu() for k, p in ctrl.named_parameters() if p.grad is not None})
    return loss.detach().double().cpu(), grads


@pytest.mark.parametrize("precision", ["fp32", "bf16x6", "bf16x3", "bf16"])
def test_cfg5_step_as_benchmarked(dev, record, precision):
    """ONE step of exactly what ``bench.py`` times as cfg #5 - embed 512, 12-layer controller on csrc/mst_ctrl.hip (native=True),
    32 tracks x 262144, lean console, deferred range / NaN checks, AudioFeatureLoss - against the oracle step.
    fp32: loss <= 1e-4 and EVERY weight gradient three-way (HIP no further from float64 than twice the fp32 oracle is, + 1e-4).  The
    float64 leg runs the oracle's torch code on the device (rocBLAS / native kernels; 6 TFLOP of float64 convolutions would take
    the host minutes), the fp32 leg is the reference path on the host cores.
    bf16: the same step with bf16 operand storage, against the float64 oracle WITH a bf16 rounding at every store
    (oracle/encoder_restated.py emulate_bf16): the distance that is left is the kernels', not the format's."""
    from mst.loss import AudioFeatureLoss
    from mst.mixing import naive_random_mix
    from mst.modules import AdvancedMixConsole, MixStyleTransferModel, SpectrogramEncoder, TransformerController
    from mst.system import CommonStep

    bs, T, n = 1, 32, 262144
    torch.manual_seed(3001)
    model = MixStyleTransferModel(SpectrogramEncoder(embed_dim=512, precision=precision), SpectrogramEncoder(embed_dim=512, precision=precision),
                                  TransformerController(512, 27, 25, 26, num_layers=12, nhead=8, native=True))
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    step = CommonStep(model, AdvancedMixConsole(44100, materialize_mixed_tracks=False, validate="deferred", param_dicts="lazy"), naive_random_mix,
                      AudioFeatureLoss([0.1, 0.001, 1.0, 1.0, 0.1], 44100), generate_mix=True, active_eq_epoch=0, active_compressor_epoch=0,
                      active_fx_bus_epoch=1000, active_master_bus_epoch=0, nan_check="deferred")
    torch.manual_seed(3002)
    tracks = 0.05 * torch.randn(bs, T, n)
    batch = (tracks.to(dev), None, None, torch.zeros(bs, T, dtype=torch.bool, device=dev), None, ["a"])
    torch.manual_seed(3003)
    loss, _ = step(batch, train=True)
    loss.backward()
    step.check_finite()
    step.mix_console.check_parameters()
    hip = {k: p.grad.detach().double().cpu() for k, p in model.named_parameters() if p.grad is not None}

    torch.manual_seed(3003)
    for _ in range(2):  # the reference draws (and mixes) twice, keeping the second (mst/system.py:149-173, :222-246)
        ref_params = [torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)]
    l64, g64 = _oracle_cfg5_step(tracks, sd0, ref_params, torch.float64, dev, emulate_bf16=(precision == "bf16"))
    assert set(hip) == set(g64), set(hip) ^ set(g64)
    h64 = {k: rel(hip[k], g64[k]) for k in g64}
    e_loss = abs(loss.item() - l64.item()) / abs(l64.item())
    if precision in ("fp32", "bf16x6", "bf16x3"):  # fp32 tensors: the split-operand forms are held to the fp32 bound
        l32, g32 = _oracle_cfg5_step(tracks, sd0, ref_params, torch.float32, torch.device("cpu"))
        r64 = {k: rel(g32[k], g64[k]) for k in g64}
        worst = max(g64, key=lambda k: h64[k] / (2 * r64[k] + 1e-4))
        rep = dict(loss=e_loss, loss_ref32=abs(l32.item() - l64.item()) / abs(l64.item()), worst_ratio=h64[worst] / (2 * r64[worst] + 1e-4),
                   h64_max=max(h64.values()), r64_max=max(r64.values()), n_grads=float(len(g64)))
        print(f"\n[cfg #5 as benchmarked, {precision}] loss vs f64 {e_loss:.2e} (fp32 oracle {rep['loss_ref32']:.2e}); {len(g64)} weight gradients: HIP vs f64 <= "
              f"{rep['h64_max']:.2e}, fp32 oracle vs f64 <= {rep['r64_max']:.2e}; worst {worst}: HIP {h64[worst]:.2e} vs oracle {r64[worst]:.2e}")
        record(**rep)
        assert e_loss < 1e-4
        bad = {k: (h64[k], r64[k]) for k in g64 if h64[k] > 2 * r64[k] + 1e-4}
        assert not bad, bad
    else:
        by = lambda s: max(v for k, v in h64.items() if s in k)
        import statistics

        rep = dict(loss=e_loss, h_emul_max=max(h64.values()), h_emul_median=statistics.median(h64.values()), conv=by("conv"), bn=by(".bn"),
                   controller=by("controller."))
        print(f"\n[cfg #5 as benchmarked, bf16] loss vs bf16-storage emulation in f64 {e_loss:.2e}; gradients vs emulation: median "
              f"{rep['h_emul_median']:.2e}, conv <= {rep['conv']:.2e}, BatchNorm <= {rep['bn']:.2e}, controller <= {rep['controller']:.2e}")
        record(**rep)
        # batch 1: the mix encoder normalises over 2 signals - at fp32 the reference's own autograd sits up to 0.6 from float64 on the same
        # gradients (the fp32 case of this test), so the bound on a bf16-storage run is the loss, the typical gradient and the controller
        assert e_loss < 5e-3 and rep["h_emul_median"] < 0.1 and rep["controller"] < 0.1 and rep["h_emul_max"] < 0.75
