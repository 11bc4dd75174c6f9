#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ FROM THE REAL REFERENCE.

Runs ONLY in the build container (needs /root/reference, which never travels to
the GPU box).  Usage:  python -B tests/golden/make_golden.py [system] [fx]

What it does
------------
1. Installs ``sys.modules`` stubs for the third-party packages the reference imports
   but this image lacks (torchaudio, librosa, dasp_pytorch) - the stub for
   ``dasp_pytorch.functional`` exposes the oracle's restated ops, i.e. they are
   patched in exactly at the reference's import seam (mst/modules.py:7-14).
2. Imports the reference's own ``mst.modules`` / ``mst.mixing`` / ``mst.loss`` /
   ``mst.filter`` from /root/reference (no bytecode written) and runs
   ``AdvancedMixConsole.forward`` (+ autograd backward), ``naive_random_mix``,
   ``AudioFeatureLoss`` and ``barkscale_fbanks`` on seeded inputs.
3. Asserts the oracle's restatement of the mst-owned logic reproduces those
   outputs, then writes inputs + expected outputs as small ``.npz`` fixtures.

The fixtures are data (inputs and expected outputs); no reference source is stored.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import numpy as np
import torch

from oracle import console_restated as oc
from oracle import dasp_restated as od
from oracle import loss_restated as ol

REF = "/root/reference"


def install_stubs():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import refstubs

    # auraloss stand-in = the oracle's restatement (auraloss 0.4.0 is absent: parity unpinned, oracle/__init__.py)
    def mr(x, y, fft_sizes, hop_sizes, win_lengths, **kw):
        return ol.mrstft_loss(x, y, tuple(zip(fft_sizes, hop_sizes, win_lengths)), **kw)

    refstubs.install_stubs(dasp_ops=od, auraloss_mrstft=mr)


def flat(d):
    return {f"{e}.{p}": v for e, pd in d.items() for p, v in pd.items()}


def short_ir(tp, mp):
    """Keep every impulse response far shorter than the clip.

    The reference filters by frequency sampling with n_fft = 2^ceil(log2(2n-1)): a CIRCULAR
    convolution.  At production lengths (n >= 131072) the wrapped tail is < 1e-11, but in a
    16384-sample fixture a 250 ms attack (tau ~ 5000 samples) or a 20 Hz / Q 5 shelf would alias
    by percents - an artefact of the short clip, not of the console.  Small fixtures therefore
    draw attack <= 29.5 ms and low-shelf / band0 corner >= ~300 Hz; the full parameter box is
    exercised by the n = 131072 fixture.
    """
    tp = tp.clone()
    mp = mp.clone()
    tp[..., 21] *= 0.1
    mp[..., 20] *= 0.1
    tp[..., 2] = 0.15 + 0.85 * tp[..., 2]
    tp[..., 5] = 0.15 + 0.85 * tp[..., 5]
    mp[..., 1] = 0.15 + 0.85 * mp[..., 1]
    mp[..., 4] = 0.15 + 0.85 * mp[..., 4]
    return tp, mp


def console_case(name, bs, T, n, seed, flags, ref_console, full_box=False):
    torch.manual_seed(seed)
    tracks = 0.1 * torch.randn(bs, T, n)
    tp = torch.rand(bs, T, 27)
    fp = torch.rand(bs, 25)
    mp = torch.rand(bs, 26)
    if not full_box:
        tp, mp = short_ir(tp, mp)
    tp.requires_grad_(True)
    mp.requires_grad_(True)
    gmix = torch.randn(bs, 2, n)

    if full_box:  # stored as float16 to keep the fixture small: make the input exactly representable
        tracks = tracks.half().float()
        gmix = torch.sign(gmix)
    mixed, mix, tpd, fpd, mpd = ref_console(tracks, tp, fp, mp, **flags)
    (mix * gmix).sum().backward()
    zg = lambda t: torch.zeros_like(t) if t.grad is None else t.grad.clone()
    g_tp, g_mp = zg(tp), zg(mp)

    # the oracle's own orchestration must reproduce the reference's
    tp2 = tp.detach().clone().requires_grad_(True)
    mp2 = mp.detach().clone().requires_grad_(True)
    o_mixed, o_mix, o_tpd, _, o_mpd = oc.console_forward(tracks, tp2, fp, mp2, sample_rate=44100, **flags)
    (o_mix * gmix).sum().backward()
    assert torch.equal(o_mix, mix), f"{name}: oracle mix != reference orchestration"
    assert torch.equal(o_mixed, mixed)
    assert torch.equal(zg(tp2), g_tp) and torch.equal(zg(mp2), g_mp)
    for k, v in flat(tpd).items():
        assert torch.equal(v, flat(o_tpd)[k]), k
    for k, v in flat(mpd).items():
        assert torch.equal(v, flat(o_mpd)[k]), k

    # float64 evaluation of the same algorithm: the value both fp32 paths (the reference's autograd above, the HIP console) approximate -
    # lets the GPU test bound the parameter gradients three-way instead of by a measured tolerance
    tp64 = tp.detach().double().requires_grad_(True)
    mp64 = mp.detach().double().requires_grad_(True)
    _, mix64, *_ = oc.console_forward(tracks.double(), tp64, fp.double(), mp64, sample_rate=44100, **flags)
    (mix64 * gmix.double()).sum().backward()

    out = dict(
        grad_track_params_f64=zg(tp64).numpy(),
        grad_master_bus_params_f64=zg(mp64).numpy(),
        tracks=tracks.numpy().astype(np.float16) if full_box else tracks.numpy(),
        track_params=tp.detach().numpy(),
        fx_bus_params=fp.numpy(),
        master_bus_params=mp.detach().numpy(),
        grad_mix=gmix.numpy().astype(np.int8) if full_box else gmix.numpy(),
        mix=mix.detach().numpy()[..., ::4] if full_box else mix.detach().numpy(),
        mix_stride=np.array(4 if full_box else 1),
        mixed_tracks_sub=mixed.detach().numpy()[..., ::64],
        grad_track_params=g_tp.numpy(),
        grad_master_bus_params=g_mp.numpy(),
        flags=np.array(sorted(flags.items()), dtype=object).astype(str),
    )
    for k, v in flat(tpd).items():
        out["tp." + k] = v.detach().numpy()
    for k, v in flat(mpd).items():
        out["mp." + k] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, f"console_{name}.npz"), **out)
    print(f"console_{name}: mix rms {mix.pow(2).mean().sqrt():.4e}  |g_tp| {g_tp.abs().max():.3e}")


def fx_case(ref_console):
    """The REAL console with the reference's DEFAULT flags - fx bus on (mst/modules.py:275-284): stereo_bus +
    noise_shaped_reverberation (65536-tap impulse response, 1023-tap band-passes) with the restated dasp ops at the seam.
    The op draws its noise with torch.randn inside: the generator is seeded right before the call, the test re-draws it."""
    bs, T, n, noise_seed = 1, 3, 65536, 900
    torch.manual_seed(41)
    tracks = (0.1 * torch.randn(bs, T, n)).half().float()
    tp = torch.rand(bs, T, 27).requires_grad_(True)
    fp = torch.rand(bs, 25).requires_grad_(True)
    mp = torch.rand(bs, 26).requires_grad_(True)
    gmix = torch.sign(torch.randn(bs, 2, n))
    torch.manual_seed(noise_seed)
    mixed, mix, tpd, fpd, mpd = ref_console(tracks, tp, fp, mp)  # no flags: every default, fx bus included
    (mix * gmix).sum().backward()
    torch.manual_seed(noise_seed)
    noise = torch.randn(bs * 2, 12, 65536 + 1022)
    tp2, fp2, mp2 = (t.detach().clone().requires_grad_(True) for t in (tp, fp, mp))
    _, o_mix, *_ = oc.console_forward(tracks, tp2, fp2, mp2, fx_noise=noise, use_fx_bus=True)
    (o_mix * gmix).sum().backward()
    assert torch.equal(o_mix, mix), "oracle fx-bus orchestration != reference"
    assert torch.equal(tp2.grad, tp.grad) and torch.equal(fp2.grad, fp.grad) and torch.equal(mp2.grad, mp.grad)
    assert torch.equal(fpd["reverberation"]["mix"], torch.ones(bs))
    np.savez_compressed(
        os.path.join(HERE, "fxbus_1x3x65536.npz"), shape=np.array([bs, T, n]), tracks=tracks.numpy().astype(np.float16),
        track_params=tp.detach().numpy(), fx_bus_params=fp.detach().numpy(), master_bus_params=mp.detach().numpy(),
        grad_mix=gmix.numpy().astype(np.int8), noise_seed=np.array(noise_seed), noise_sub=noise.numpy()[:, :, ::4096],
        mix=mix.detach().numpy()[..., ::4], grad_track_params=tp.grad.numpy(), grad_fx_bus_params=fp.grad.numpy(),
        grad_master_bus_params=mp.grad.numpy(), band3_decay=fpd["reverberation"]["band3_decay"].detach().numpy(),
    )
    print(f"console_fxbus: mix rms {mix.pow(2).mean().sqrt():.4e}  |g_fp| {fp.grad.abs().max():.3e}")


def system_case(rsystem, rmodules, rmixing):
    """The REAL ``System.common_step`` (mst/system.py:102-407; Lightning stood in by tests/refstubs.py) with the real
    console / naive_random_mix, the restated dasp ops at the seam and the restated MR-STFT loss: generate_mix, both
    random-mix blocks, peak normalise, A/B split (strided tracks_b), stub model, console with grad, loss sum."""
    import auraloss
    from util import StubModel

    bs, T, n = 2, 4, 131072
    torch.manual_seed(31)
    tracks = (0.1 * torch.randn(bs, T, n)).half().float()
    model = StubModel(seed=7)
    res = dict(fft_sizes=[512, 2048, 8192], hop_sizes=[256, 1024, 4096], win_lengths=[512, 2048, 8192])
    system = rsystem.System(model=model, mix_console=rmodules.AdvancedMixConsole(sample_rate=44100), mix_fn=rmixing.naive_random_mix,
                            loss=auraloss.freq.MultiResolutionSTFTLoss(**res), generate_mix=True, active_eq_epoch=0,
                            active_compressor_epoch=0, active_fx_bus_epoch=1000, active_master_bus_epoch=0)
    batch = (tracks, None, None, torch.zeros(bs, T, dtype=torch.bool), None, ["a", "b"])
    torch.manual_seed(32)  # the RNG stream naive_random_mix draws from (twice: system.py:149-173 and :222-246)
    loss, data = system.common_step(batch, 0, train=True)
    loss.backward()
    np.savez_compressed(
        os.path.join(HERE, "system_step.npz"), shape=np.array([bs, T, n]), tracks_sub=tracks.numpy()[..., ::1024], seed_tracks=31, seed_model=7, seed_mix=32,
        loss=np.array(loss.item()), ref_mix_b_sub=data["ref_mix_b_norm"].numpy()[..., ::16],
        pred_mix_b_sub=data["pred_mix_b_norm"].numpy()[..., ::16], ref_mix_a_sub=data["ref_mix_a"].numpy()[..., ::16],
        sum_mix_b_sub=data["sum_mix_b"].numpy()[..., ::16],
        g_w_track=model.w_track.grad.numpy(), g_w_fx=np.zeros((25, 4), np.float32) if model.w_fx.grad is None else model.w_fx.grad.numpy(),
        g_w_master=model.w_master.grad.numpy(),
        pred_track_ratio=data["pred_track_param_dict"]["compressor"]["ratio"].detach().numpy(),
        ref_master_thr=data["ref_master_bus_param_dict"]["compressor"]["threshold_db"].detach().numpy(),
    )
    print(f"system_step: loss {loss.item():.6f}  |g_w_track| {model.w_track.grad.abs().max():.3e}")


def run_case(rmodules):
    """The REAL ``mst.utils.run_diffmst`` (mst/utils.py:32-173): LUFS normalisation (stubbed meter = tests/util.py
    simple_lufs), one parameter estimate, Hann-faded overlap-add of console forwards over 262144-sample windows.  Five tracks,
    one of them below -80 LUFS (dropped), a last window of 100000 samples."""
    import pyloudnorm
    from util import StubModel, simple_lufs

    import mst.utils as rutils

    pyloudnorm.Meter.integrated_loudness = lambda self, x: simple_lufs(x)
    T, n = 5, 3 * 131072 + 100000
    torch.manual_seed(51)
    tracks = (0.05 * torch.randn(1, T, n) * torch.tensor([1.0, 0.3, 2.0, 1e-6, 0.7]).view(1, T, 1)).half().float()
    ref = 0.2 * torch.randn(1, 2, 300000)
    model = StubModel(seed=8)
    with torch.no_grad():
        pred_mix, tpd, fpd, mpd = rutils.run_diffmst(tracks.clone(), ref, model, rmodules.AdvancedMixConsole(sample_rate=44100),
                                                     track_start_idx=1000, ref_start_idx=2000)
    assert pred_mix.shape == (1, 2, n) and tpd["compressor"]["ratio"].shape == (1, 4)
    np.savez_compressed(
        os.path.join(HERE, "run_diffmst.npz"), shape=np.array([T, n]), seed_tracks=51, seed_model=8, ref_len=300000,
        track_start_idx=1000, ref_start_idx=2000, tracks_sub=tracks.numpy()[..., ::4096], ref_sub=ref.numpy()[..., ::4096],
        pred_mix_sub=pred_mix.numpy()[..., ::8], pred_mix_l2=np.array(pred_mix.double().pow(2).sum().sqrt().item()),
        seam=pred_mix.numpy()[..., 262144 - 64:262144 + 64],
        track_ratio=tpd["compressor"]["ratio"].numpy(), master_thr=mpd["compressor"]["threshold_db"].numpy(),
    )
    print(f"run_diffmst: mix rms {pred_mix.pow(2).mean().sqrt():.4e}")


def encoder_setup(enc_cls, seed_init=71, seed_bn=72, embed_dim=64):
    """Seeded SpectrogramEncoder: default (xavier) initialisation under ``seed_init``, then BatchNorm affine parameters moved
    off their 1 / 0 defaults under ``seed_bn`` so that their gradients and the ReLU masks are generic.  The GPU test builds
    the HIP encoder with the same two calls (same torch build on the GPU box: same CPU generator stream)."""
    torch.manual_seed(seed_init)
    enc = enc_cls(embed_dim=embed_dim)
    torch.manual_seed(seed_bn)
    with torch.no_grad():
        for name, p in sorted(enc.named_parameters()):
            if ".bn" in name and name.endswith("weight"):
                p.copy_(0.5 + torch.rand_like(p))
            elif ".bn" in name and name.endswith("bias"):
                p.copy_(0.2 * torch.randn_like(p))
    return enc


def encoder_case(rmodules):
    """The REAL ``mst.modules.SpectrogramEncoder`` + ``mst.panns.Cnn14`` (pure torch, imported unchanged): one training-mode
    forward + backward and one eval-mode forward on seeded audio.  Also pins oracle/encoder_restated.py."""
    from oracle import encoder_restated as oe

    enc = encoder_setup(rmodules.SpectrogramEncoder)
    bs, n = 2, 65536
    torch.manual_seed(73)
    wave = (0.1 * torch.randn(bs, 1, n)).half().float()
    G = torch.randn(bs, 64)
    sd0 = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    enc.train()
    embed = enc(wave)
    (embed * G).sum().backward()
    grads = {k: p.grad.detach().clone() for k, p in enc.named_parameters()}
    sd1 = {k: v.detach().clone() for k, v in enc.state_dict().items()}
    enc.eval()
    with torch.no_grad():
        embed_eval = enc(wave)
    # the oracle's restatement on the same state_dict: identical torch ops, so identical numbers
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k and k != "window" else v.clone()) for k, v in sd0.items()}
    o_embed = oe.spectrogram_encoder(wave, osd, training=True)
    (o_embed * G).sum().backward()
    assert torch.equal(o_embed, embed), "oracle encoder != reference"
    for k, g in grads.items():
        assert torch.equal(osd[k].grad, g), k
    for k in sd1:
        if "running" in k:
            assert torch.equal(osd[k], sd1[k]), k
    with torch.no_grad():
        assert torch.equal(oe.spectrogram_encoder(wave, {k: v for k, v in sd1.items()}, training=False), embed_eval)
    spec = oe.spectrogram(wave.view(bs, n))
    out = dict(shape=np.array([bs, n]), seed_init=71, seed_bn=72, seed_wave=73, wave_sub=wave.numpy()[..., ::1024], G=G.numpy(),
               embed=embed.detach().numpy(), embed_eval=embed_eval.numpy(), spec_sub=spec.numpy()[:, ::41, ::7],
               spec_l2=np.array(spec.double().pow(2).sum().sqrt().item()))
    for k, g in grads.items():
        if g.numel() <= 4096 or k.startswith("model.fc"):
            out["g." + k] = g.numpy()
        else:
            out["gsub." + k] = g.flatten()[::997].numpy()
            out["gl2." + k] = np.array(g.double().pow(2).sum().sqrt().item())
    for k, v in sd0.items():
        if k.endswith("conv1.weight") or k.endswith("conv2.weight"):
            out["wsum." + k] = np.array([v.double().sum().item(), v.double().abs().sum().item()])
    for k in ("model.conv_block1.bn1.running_mean", "model.conv_block1.bn2.running_var", "model.conv_block3.bn1.running_var",
              "model.conv_block6.bn2.running_mean", "model.conv_block6.bn2.running_var"):
        out["run." + k] = sd1[k].numpy()
    np.savez_compressed(os.path.join(HERE, "encoder_2x65536.npz"), **out)
    print(f"encoder: |embed| {embed.abs().max():.4e}  |g conv1| {grads['model.conv_block1.conv1.weight'].abs().max():.3e}")


def controller_setup(ctrl_cls, seed_init=81, seed_pert=82, embed_dim=512, num_layers=3, nhead=8):
    """Seeded TransformerController: default initialisation under ``seed_init``, then LayerNorm parameters and every bias moved off
    their 1 / 0 defaults under ``seed_pert``.  The GPU test builds the HIP-backed controller with the same two calls (same torch
    build on the GPU box: same CPU generator stream, same construction order = same weights; the generator asserts that)."""
    torch.manual_seed(seed_init)
    ctrl = ctrl_cls(embed_dim, 27, 25, 26, num_layers=num_layers, nhead=nhead)
    torch.manual_seed(seed_pert)
    with torch.no_grad():
        for name, p in sorted(ctrl.named_parameters()):
            if "norm" in name or name.endswith("bias"):
                p.add_(0.1 * torch.randn_like(p))
    return ctrl


def controller_case(rmodules):
    """The REAL ``mst.modules.TransformerController`` (mst/modules.py:809-914, pure torch, imported unchanged): training-mode forward
    + backward on seeded embeddings with a padding mask; every output, the input gradients and every parameter gradient
    (small ones whole, large ones as a strided sample + L2 norm)."""
    sys.path[:0] = [os.path.join(ROOT, "diff-mst_amd")]
    from diffmst_hip.modules import TransformerController as Ours

    ctrl = controller_setup(rmodules.TransformerController).train()
    ours = controller_setup(Ours)
    sd, sd_ours = ctrl.state_dict(), ours.state_dict()
    assert list(sd) == list(sd_ours) and all(torch.equal(sd[k], sd_ours[k]) for k in sd), "seeded construction differs from the reference's"
    bs, T = 2, 10
    torch.manual_seed(83)
    te = torch.randn(bs, T, 512)
    me = torch.randn(bs, 2, 512)
    mask = torch.zeros(bs, T, dtype=torch.bool)
    mask[0, 6:] = True
    mask[1, 2] = True
    w_t, w_f, w_m = torch.randn(bs, T, 27), torch.randn(bs, 25), torch.randn(bs, 26)
    a, b = te.clone().requires_grad_(True), me.clone().requires_grad_(True)
    tp, fp, mp = ctrl(a.clone(), b.clone(), mask)  # the reference adds the type embeddings IN PLACE (:871-872): hand it copies
    ((tp * w_t).sum() + (fp * w_f).sum() + (mp * w_m).sum()).backward()
    out = dict(seed_init=81, seed_pert=82, track_embeds=te.numpy(), mix_embeds=me.numpy(), mask=mask.numpy(), w_t=w_t.numpy(), w_f=w_f.numpy(),
               w_m=w_m.numpy(), track_params=tp.detach().numpy(), fx_params=fp.detach().numpy(), master_params=mp.detach().numpy(),
               g_track_embeds=a.grad.numpy(), g_mix_embeds=b.grad.numpy())
    for k, p in ctrl.named_parameters():
        g = p.grad
        if g.numel() <= 4096:
            out["g." + k] = g.numpy()
        else:
            out["gsub." + k] = g.flatten()[::499].numpy()
            out["gl2." + k] = np.array(g.double().pow(2).sum().sqrt().item())
    np.savez_compressed(os.path.join(HERE, "controller_2x10.npz"), **out)
    print("controller fixture:", sum(v.nbytes for v in out.values() if hasattr(v, "nbytes")) // 1024, "KiB before compression")


def console_cases(ref_console):
    basic = dict(
        use_track_input_fader=True, use_track_eq=False, use_track_compressor=False, use_track_panner=True,
        use_fx_bus=False, use_master_bus=False, use_output_fader=False,
    )
    full = dict(
        use_track_input_fader=True, use_track_eq=True, use_track_compressor=True, use_track_panner=True,
        use_fx_bus=False, use_master_bus=True, use_output_fader=True,
    )
    refmix = dict(full, use_track_input_fader=False)  # system.py:162 (reference-mix flags)
    # cfg #1 shape cut to fixture size (gain+pan only = "BasicMixConsole")
    console_case("basic_2x4x16384", 2, 4, 16384, 1, basic, ref_console)
    console_case("full_2x4x16384", 2, 4, 16384, 2, full, ref_console)
    console_case("full_1x8x32768", 1, 8, 32768, 3, full, ref_console)
    console_case("refmix_2x3x8192", 2, 3, 8192, 4, refmix, ref_console)
    # System's training length (mst/system.py:255-258), full parameter box
    console_case("fullbox_1x2x131072", 1, 2, 131072, 5, full, ref_console, full_box=True)


def main():
    assert os.path.isdir(REF), "golden generation needs /root/reference (build container only)"
    install_stubs()
    sys.path.insert(0, REF)
    import mst.filter as rfilter
    import mst.loss as rloss
    import mst.mixing as rmixing
    import mst.modules as rmodules
    import mst.system as rsystem

    ref_console = rmodules.AdvancedMixConsole(sample_rate=44100)
    assert ref_console.param_ranges == oc.param_ranges(44100)
    only = set(sys.argv[1:])  # e.g. `make_golden.py system` regenerates one fixture family (default: all)
    if only and only <= {"system", "fx", "run", "encoder", "controller", "console"}:
        if "console" in only:
            console_cases(ref_console)
        if "encoder" in only:
            encoder_case(rmodules)
        if "controller" in only:
            controller_case(rmodules)
        if "run" in only:
            run_case(rmodules)
        if "system" in only:
            system_case(rsystem, rmodules, rmixing)
        if "fx" in only:
            fx_case(ref_console)
        return

    console_cases(ref_console)

    # naive_random_mix (mixing.py:35-94): RNG draw order and the 8-tuple
    torch.manual_seed(11)
    tracks = 0.1 * torch.randn(2, 4, 8192)
    tracks = torch.cat([tracks] * 8, dim=-1)  # 65536 samples: long enough for the full parameter box
    torch.manual_seed(12)
    r = rmixing.naive_random_mix(tracks, ref_console, use_fx_bus=False)
    assert len(r) == 8
    np.savez_compressed(
        os.path.join(HERE, "naive_random_mix.npz"),
        tracks=tracks.numpy()[..., :8192], seed=12, mix=r[1].numpy()[..., ::8], mix_params=r[5].numpy(),
        fx_bus_params=r[6].numpy(), master_bus_params=r[7].numpy(),
    )

    # out-of-range parameter => ValueError text (modules.py:86-89)
    bad = torch.rand(1, 2, 27)
    bad[0, 1, 25] = 1.5
    try:
        ref_console(torch.zeros(1, 2, 4096), bad, torch.rand(1, 25), torch.rand(1, 26), use_fx_bus=False)
        raise SystemExit("expected ValueError")
    except ValueError as e:
        err_text = str(e)
    print("ValueError text:", err_text)

    # Bark filterbank (filter.py:107-161) - constant table
    fb = rfilter.barkscale_fbanks(16385, 20.0, 20000.0, 24, 44100)
    ofb = ol.bark_filterbank(16385, 20.0, 20000.0, 24, 44100)
    assert torch.equal(fb, ofb), "bark filterbank restatement differs"
    np.savez_compressed(
        os.path.join(HERE, "bark_fb.npz"), col_sums=fb.sum(0).numpy(), row_sub=fb[::257].numpy(),
        argmax=fb.argmax(0).numpy(), nnz=np.array((fb > 0).sum().item()), err_text=np.array(err_text),
    )

    # AudioFeatureLoss (loss.py:198-260), weights of unpaired+feat.yaml:55-60
    weights = [0.1, 0.001, 1.0, 1.0, 0.1]
    torch.manual_seed(21)
    a = (0.2 * torch.randn(2, 2, 65536)).requires_grad_(True)
    b = 0.3 * torch.randn(2, 2, 65536) * torch.tensor([1.0, 0.6]).view(1, 2, 1)
    afl = rloss.AudioFeatureLoss(weights=weights, sample_rate=44100)
    ld = afl(a, b)
    sum(v.mean() for v in ld.values()).backward()  # system.py:334-336
    old = ol.audio_feature_loss(a.detach(), b, weights)
    assert list(ld.keys()) == list(ol.AF_KEYS)
    for k in ld:
        assert torch.allclose(ld[k].detach(), old[k], rtol=1e-6, atol=0), k
    feats = {
        "rms": rloss.compute_rms(a.detach()), "crest": rloss.compute_crest_factor(a.detach()),
        "width": rloss.compute_stereo_width(a.detach()), "imbalance": rloss.compute_stereo_imbalance(a.detach()),
        "bark": rloss.compute_barkspectrum(a.detach(), sample_rate=44100),
    }
    np.savez_compressed(
        os.path.join(HERE, "af_loss.npz"), input=a.detach().numpy(), target=b.numpy(), weights=np.array(weights),
        grad_input_sub=a.grad.numpy()[..., ::16], grad_input_l2=np.array(a.grad.pow(2).sum().sqrt().item()),
        **{"loss." + k: v.detach().numpy() for k, v in ld.items()}, **{"feat." + k: v.numpy() for k, v in feats.items()},
    )

    # batch_stereo_peak_normalize (utils.py:14-29) cannot be imported (pyloudnorm); its 3 lines of
    # arithmetic are evaluated literally here to pin the oracle.
    x = torch.randn(3, 2, 1000)
    g = x.abs().max(dim=-1, keepdim=True)[0].max(dim=-2, keepdim=True)[0]
    assert torch.equal(x / g.clamp(1e-8), oc.batch_stereo_peak_normalize(x))
    system_case(rsystem, rmodules, rmixing)
    fx_case(ref_console)
    run_case(rmodules)
    encoder_case(rmodules)
    controller_case(rmodules)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
