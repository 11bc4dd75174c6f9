"""The UNCHANGED kernel sources, run on the CPU by the host-side HIP simulator (tests/hostsim), against
the oracle.  Small sizes: this checks indexing / scan / carry / adjoint logic without a GPU; the parity
tests proper are the `-m gpu` ones."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "hostsim"))
import harness  # noqa: E402
from oracle import console_restated as oc  # noqa: E402
from oracle import loss_restated as ol  # noqa: E402
from util import FULL, rel, short_ir  # noqa: E402


@pytest.fixture(scope="module")
def ranges():
    harness.lib()  # builds tests/hostsim/build/libdiffmst_hostsim.so
    return oc.param_ranges(44100)


def test_console_forward_backward_small(ranges):
    torch.manual_seed(1)
    bs, T, n = 1, 2, 5003  # ragged: not a multiple of 4, 8 or 64; longer than both look-aheads
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    tp, mp = short_ir(tp, mp)
    tp[..., 21] *= 0.2
    mp[..., 20] *= 0.2
    gmix, gmixed = torch.randn(bs, 2, n), torch.randn(bs, 2, T, n)
    out = harness.console(ranges, tracks, tp, fp, mp, FULL, grad_mix=gmix, grad_mixed=gmixed, want_grad_tracks=True)
    assert out["status"] == 0
    tr = tracks.double().requires_grad_(True)
    a, b = tp.double().requires_grad_(True), mp.double().requires_grad_(True)
    mixed, mix, *_ = oc.console_forward(tr, a, fp.double(), b, **FULL)
    ((mix * gmix.double()).sum() + (mixed * gmixed.double()).sum()).backward()
    _, truth, *_ = oc.console_forward(tracks.double(), tp.double(), fp.double(), mp.double(), time_domain=True, **FULL)
    assert rel(out["mix"], truth) < 1e-4
    assert rel(out["mixed"], mixed) < 1e-3
    assert rel(out["grad_tracks"], tr.grad) < 1e-2
    assert rel(out["grad_tp"], a.grad) < 2e-2
    assert rel(out["grad_mp"], b.grad) < 2e-2


def test_console_inwave_scan_eq_matches_three_kernel_eq(ranges):
    """EQ carries scanned inside the zs / run kernels (tile aggregates, in-wave scans) vs zs / scan / run on the same
    inputs: 5 tiles per row, ragged tail, forward and both adjoint cascades (master and grad_tracks)."""
    torch.manual_seed(3)
    bs, T, n = 1, 2, 4 * 4096 + 777
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    gmix = torch.randn(bs, 2, n)
    a = harness.console(ranges, tracks, tp, fp, mp, FULL, grad_mix=gmix, want_mixed=False, want_grad_tracks=True)
    b = harness.console(ranges, tracks, tp, fp, mp, FULL, grad_mix=gmix, want_mixed=False, want_grad_tracks=True,
                        multipass_eq=True)
    assert a["status"] == 0 and b["status"] == 0
    assert rel(a["mix"], b["mix"]) < 5e-6  # round 3: the in-wave path takes its zero-state ends from the MFMA map (a 64-term dot product, not the recursion)
    # two fp32 evaluation orders (round 3: MFMA zero-state map vs recursion, in both adjoint cascades); on the GPU both sit
    # 1.3e-5 .. 3.5e-5 from float64 at 131072 samples and 6e-6 from each other (tools/dbg_gtracks_paths.py); this short clip: 1.0e-4
    assert rel(a["grad_tracks"], b["grad_tracks"]) < 2e-4
    assert rel(a["grad_tp"], b["grad_tp"]) < 1e-4
    assert rel(a["grad_mp"], b["grad_mp"]) < 1e-4


def test_console_generic_scan_width(ranges):
    """A row length for which a carry-scan lane owns K = 3 chunks (not one of the unrolled widths 1/2/4/8): the
    sub-span path of k_scan, for the 12-state cascades (three-kernel EQ, forward and adjoint) and the all-pole bank."""
    torch.manual_seed(4)
    bs, T, n = 1, 1, 64 * 1025 + 11  # 1026 chunks of 64 samples -> K = ceil(1026 / 512) = 3
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    tp, mp = short_ir(tp, mp)  # the frequency-sampling oracle is circular: keep impulse responses short at this length
    gmix = torch.randn(bs, 2, n)
    b = harness.console(ranges, tracks, tp, fp, mp, FULL, grad_mix=gmix, want_mixed=False, multipass_eq=True)
    x, y = tp.double().requires_grad_(True), mp.double().requires_grad_(True)
    _, mix, *_ = oc.console_forward(tracks.double(), x, fp.double(), y, **FULL)
    (mix * gmix.double()).sum().backward()
    assert rel(b["mix"], mix) < 1e-4
    assert rel(b["grad_tp"], x.grad) < 2e-2 and rel(b["grad_mp"], y.grad) < 2e-2


def test_console_denormalised_parameter_path(ranges):
    """MST_NO_RANGE_CHECK + identity ranges (forward_mix_console): same signal as the normalised call on the same
    parameters, gradients scaled by 1/(hi - lo), and values outside [0,1] raise no status."""
    from mst import _desc

    torch.manual_seed(6)
    bs, T, n = 1, 2, 3000
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    tp, mp = short_ir(tp, mp)
    gmix = torch.randn(bs, 2, n)
    a = harness.console(ranges, tracks, tp, fp, mp, FULL, grad_mix=gmix, want_mixed=False)
    tlo, thi = (torch.tensor(v) for v in _desc.range_vectors(ranges, _desc.TRACK_INDEX))
    mlo, mhi = (torch.tensor(v) for v in _desc.range_vectors(ranges, _desc.MASTER_INDEX))
    b = harness.console(ranges, tracks, tp * (thi - tlo) + tlo, fp, mp * (mhi - mlo) + mlo, FULL, grad_mix=gmix,
                        want_mixed=False, denormalized=True)
    assert a["status"] == 0 and b["status"] == 0  # e.g. -60 dB thresholds are far outside [0,1] and must not be flagged
    assert rel(b["mix"], a["mix"]) < 1e-6
    assert rel(b["grad_tp"] * (thi - tlo), a["grad_tp"]) < 1e-4
    assert rel(b["grad_mp"] * (mhi - mlo), a["grad_mp"]) < 1e-4


@pytest.mark.parametrize("S,n", [(8192, 2 * 4096 + 1237),
                                 pytest.param(65536, 16 * 4096 + 465, marks=pytest.mark.slow)])  # 17 blocks: two frame chunks of the dH walk; the large case takes 3 min on 8 cores
def test_console_fx_bus(ranges, S, n):
    """use_fx_bus = True (the reference's default): send bus + noise-shaped reverberation (partitioned FFT convolution on the
    8192-point engine) forward and backward, with a short impulse response (2 partitions) and short band-passes so that the
    simulator finishes; ragged length (not a multiple of the 4096-sample hop).  S = 65536 (16 partitions, the reference's
    size) takes the register-ring multiply-accumulate kernels and rows longer than one 16-frame chunk."""
    torch.manual_seed(12)
    bs, T = 1, 2
    taps = 63
    flags = dict(FULL, use_fx_bus=True)
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    tp, mp = short_ir(tp, mp)
    tp[..., 21] *= 0.2
    mp[..., 20] *= 0.2
    tp[..., 26] = 0.8 + 0.2 * tp[..., 26]  # send -6 .. +12 dB: the wet path carries weight
    if S == 65536:
        # 63-tap band-passes pass far more noise energy than the reference's 1023-tap ones: with full band gains the 65536-tap
        # response drives the mix to +-60 and the master compressor's gradient becomes chaotic (the fp32 oracle then sits 270 %
        # from its own float64 evaluation); keep the wet level comparable to the dry one
        fp[:, :12] *= 0.03
    noise = torch.randn(bs * 2, 12, S + taps - 1)
    gmix = torch.randn(bs, 2, n)
    out = harness.console(ranges, tracks, tp, fp, mp, flags, grad_mix=gmix, want_mixed=False, want_grad_tracks=True,
                          fx_noise=noise, fx_ir_samples=S, fx_bandpass_taps=taps)
    assert out["status"] == 0
    got = dict(mix=out["mix"], g_fp=out["grad_fp"][:, :24], g_send=out["grad_tp"][..., 26], g_tp=out["grad_tp"], g_mp=out["grad_mp"],
               g_tracks=out["grad_tracks"])
    ref = {}
    for dt in (torch.float32, torch.float64):  # three-way: the compressors make the fp32 reference algorithm itself noisy here
        tr = tracks.detach().clone().to(dt).requires_grad_(True)
        a, f, b = (t.detach().clone().to(dt).requires_grad_(True) for t in (tp, fp, mp))
        _, mix, *_ = oc.console_forward(tr, a, f, b, fx_noise=noise.to(dt), fx_ir_samples=S, fx_bandpass_taps=taps, **flags)
        (mix * gmix.to(dt)).sum().backward()
        ref[dt] = dict(mix=mix.detach(), g_fp=f.grad[:, :24], g_send=a.grad[..., 26], g_tp=a.grad, g_mp=b.grad, g_tracks=tr.grad)
    dry = oc.console_forward(tracks.double(), tp.double(), fp.double(), mp.double(), **FULL)[1]
    assert rel(dry, ref[torch.float64]["mix"]) > 0.05  # the reverberated send really is part of the mix
    for k, tol in (("mix", 1e-4), ("g_fp", 2e-3), ("g_send", 2e-3), ("g_tp", 2e-2), ("g_mp", 2e-2), ("g_tracks", 1e-2)):
        h64, r64 = rel(got[k], ref[torch.float64][k]), rel(ref[torch.float32][k], ref[torch.float64][k])
        assert h64 <= 2 * r64 + tol, (k, h64, r64)
    assert float(out["grad_fp"][:, 24].abs().max()) == 0.0  # the forced-wet "mix" parameter gets no gradient


def test_console_status_flag(ranges):
    tp, fp, mp = torch.rand(1, 1, 27), torch.rand(1, 25), torch.rand(1, 26)
    mp[0, 24] = 1.5
    tp[0, 0, 26] = -0.5
    out = harness.console(ranges, torch.zeros(1, 1, 300), tp, fp, mp, FULL, want_mixed=False)
    from mst import _desc

    assert str(_desc.status_to_error(out["status"])) == "Parameter send_db of effect fx_bus is out of range."


def test_mrstft_small():
    torch.manual_seed(0)
    x = 0.3 * torch.randn(1, 2, 3000)
    y = 0.5 * x + 0.2 * torch.randn(1, 2, 3000)
    res = ((512, 256, 512), (256, 50, 200), (2048, 1024, 2048))
    out = harness.mrstft(x, y, res, w_sc=1.0, w_log_mag=0.0)
    xo = x.double().requires_grad_(True)
    lo = ol.mrstft_loss(xo, y.double(), res, w_sc=1.0, w_log_mag=0.0)
    lo.backward()
    assert abs(out["loss"].item() - lo.item()) / lo.item() < 1e-6
    assert rel(out["grad_pred"], xo.grad) < 1e-5
    full = harness.mrstft(x, y, res, grad=False)
    assert abs(full["loss"].item() - ol.mrstft_loss(x.double(), y.double(), res).item()) < 1e-5


def test_mrstft_8192_inplace_transform():
    """n_fft = 8192 runs on the in-place kernels (fft_dif / fft_dit, digit-reversed slots, bank swizzle)."""
    torch.manual_seed(5)
    x = 0.3 * torch.randn(1, 1, 20000)
    y = 0.6 * x + 0.2 * torch.randn(1, 1, 20000)
    res = ((8192, 4096, 8192),)
    full = harness.mrstft(x, y, res, grad=False)
    assert abs(full["loss"].item() - ol.mrstft_loss(x.double(), y.double(), res).item()) < 1e-5
    # gradient of the well-conditioned term only (d log|X| ~ 1/|X| is fp32-noise-limited, see the GPU tests)
    out = harness.mrstft(x, y, res, w_sc=1.0, w_log_mag=0.0)
    xo = x.double().requires_grad_(True)
    lo = ol.mrstft_loss(xo, y.double(), res, w_sc=1.0, w_log_mag=0.0)
    lo.backward()
    assert abs(out["loss"].item() - lo.item()) / lo.item() < 1e-6
    assert rel(out["grad_pred"], xo.grad) < 1e-5


def test_mrstft_register_radix_engine():
    """The round-2 kernels (mst_stft2.hip / mst_fft2.h): reference-shaped resolutions (hop = n_fft / 2, full window) on rows
    that are whole hops long; 3 x 8192 samples = 7 frames of 8192, 25 of 2048, 97 of 512 per row, two rows."""
    torch.manual_seed(8)
    n = 3 * 8192
    x = 0.3 * torch.randn(1, 2, n)
    y = 0.6 * x + 0.2 * torch.randn(1, 2, n)
    for res in (((512, 256, 512),), ((2048, 1024, 2048),), ((8192, 4096, 8192),)):
        full = harness.mrstft(x, y, res, grad=False)
        want = ol.mrstft_loss(x.double(), y.double(), res).item()
        assert abs(full["loss"].item() - want) / want < 2e-6, (res, full["loss"].item(), want)
    # backward: owner-computes overlap-add (halo strips for 512 / 2048, seam strips for 8192), row-end mirrors, each
    # resolution alone (it owns the gradient buffer) and all three (8192 first onto zeros, the others add); the smooth
    # spectral-convergence term pins the transforms and the overlap-add to fp32 round-off
    for res in (((512, 256, 512),), ((2048, 1024, 2048),), ((8192, 4096, 8192),),
                ((512, 256, 512), (2048, 1024, 2048), (8192, 4096, 8192)), ((2048, 1024, 2048), (512, 256, 512))):
        out = harness.mrstft(x, y, res, w_sc=1.0, w_log_mag=0.0)
        xo = x.double().requires_grad_(True)
        lo = ol.mrstft_loss(xo, y.double(), res, w_sc=1.0, w_log_mag=0.0)
        lo.backward()
        assert abs(out["loss"].item() - lo.item()) / lo.item() < 1e-6
        assert rel(out["grad_pred"], xo.grad) < 1e-5, res
        if len(res) == 3:  # the seam hand-over between launches: reproducible bit for bit (every combination is on the GPU)
            again = harness.mrstft(x, y, res, w_sc=1.0, w_log_mag=0.0)
            assert torch.equal(out["grad_pred"], again["grad_pred"])
    # the shortest row the round-2 kernels take (two 8192-frames deep) and the full loss (log-magnitude term included)
    xs, ys = x[..., :16384].contiguous(), y[..., :16384].contiguous()
    res = ((512, 256, 512), (2048, 1024, 2048), (8192, 4096, 8192))
    out = harness.mrstft(xs, ys, res)
    xo = xs.double().requires_grad_(True)
    lo = ol.mrstft_loss(xo, ys.double(), res)
    lo.backward()
    assert abs(out["loss"].item() - lo.item()) / lo.item() < 2e-6
    assert rel(out["grad_pred"], xo.grad) < 2e-3  # d log|X| / dX ~ 1/|X|: fp32-noise-limited (see the GPU tests)


@pytest.mark.parametrize("n", [17000, pytest.param(40962, marks=pytest.mark.slow)])  # just above the 16384-sample reflect pad (3 frames); ragged rows (n % 4 = 2), 6 frames
def test_afloss_small(n):
    torch.manual_seed(0)
    w = [0.1, 0.001, 1.0, 1.0, 0.1]
    a = 0.2 * torch.randn(1, 2, n)
    b = 0.3 * torch.randn(1, 2, n) * torch.tensor([1.0, 0.6]).view(1, 2, 1)
    gl = torch.tensor([1.0, 2.0, 0.5, 1.5, 1.0])
    out = harness.afloss(a, b, w, grad_losses=gl)
    ao = a.double().requires_grad_(True)
    ld = ol.audio_feature_loss(ao, b.double(), w)
    vals = torch.stack([ld[k] for k in ol.AF_KEYS])
    (vals * gl.double()).sum().backward()
    assert ((out["losses"].double() - vals.detach()).abs() / vals.detach().abs() < 5e-5).all()
    assert rel(out["grad_pred"], ao.grad) < 1e-5
