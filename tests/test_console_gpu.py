"""GPU parity tests of the HIP mix console against the oracle and the golden fixtures.

Tolerances (BASELINE north_star: <= 1e-4 relative fp32 vs the CPU reference path):
  * signals (mix, mixed_tracks, grad_tracks): rel-L2 <= 1e-4 vs the fp32 reference algorithm
    on the uniform-random parameter set;
  * parameter gradients: the fp32 reference's own autograd differs from float64 by ~1e-3 overall
    (ill-conditioned biquad design in fp32), so they are checked three-way:
    err(HIP, f64) <= 2 * err(ref32, f64) + 1e-4 and overall rel-L2(HIP, ref32) <= 1e-2.
"""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from util import FULL, rel, short_ir


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from mst import _hip

    _hip.lib()  # the HIP library must be the thing that runs
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def console():
    from mst.modules import AdvancedMixConsole

    return AdvancedMixConsole(44100)


def run_hip(console, dev, tracks, tp, fp, mp, flags, gmix=None, gmixed=None, grad_tracks=False):
    tr = tracks.to(dev).requires_grad_(grad_tracks)
    tp_d = tp.to(dev).requires_grad_(gmix is not None)
    mp_d = mp.to(dev).requires_grad_(gmix is not None)
    mixed, mix, tpd, fpd, mpd = console(tr, tp_d, fp.to(dev), mp_d, **flags)
    out = dict(mix=mix, mixed=mixed, tpd=tpd, mpd=mpd)
    if gmix is not None:
        loss = (mix * gmix.to(dev)).sum()
        if gmixed is not None:
            loss = loss + (mixed * gmixed.to(dev)).sum()
        loss.backward()
        out.update(g_tp=tp_d.grad, g_mp=mp_d.grad, g_tracks=tr.grad)
    torch.cuda.synchronize()
    return out


def run_oracle(tracks, tp, fp, mp, flags, gmix=None, gmixed=None, dtype=torch.float32, grad_tracks=False):
    from oracle import console_restated as oc

    tr = tracks.detach().clone().to(dtype).requires_grad_(grad_tracks)
    tp_o = tp.detach().clone().to(dtype).requires_grad_(gmix is not None)
    mp_o = mp.detach().clone().to(dtype).requires_grad_(gmix is not None)
    mixed, mix, *_ = oc.console_forward(tr, tp_o, fp.to(dtype), mp_o, **flags)
    out = dict(mix=mix, mixed=mixed)
    if gmix is not None:
        loss = (mix * gmix.to(dtype)).sum()
        if gmixed is not None:
            loss = loss + (mixed * gmixed.to(dtype)).sum()
        loss.backward()
        zero = lambda t: torch.zeros_like(t) if t.grad is None else t.grad
        out.update(g_tp=zero(tp_o), g_mp=zero(mp_o), g_tracks=tr.grad)
    return out


def assert_three_way(hip, r32, r64, key, tol32=None, slack=1e-4):
    """HIP must be no further from float64 than twice the fp32 reference algorithm itself is (fp32 biquad design is ill-conditioned
    at low f / high Q: both fp32 paths then sit ~1e-2 from float64 together) AND agree with the fp32 reference within tol32 (signals:
    the contract's 1e-4; gradients, tol32=None: the bound the first condition implies, 3 r + slack - no measured tolerance)."""
    h32, h64, r = rel(hip[key], r32[key]), rel(hip[key], r64[key]), rel(r32[key], r64[key])
    assert h64 <= 2 * r + slack, (key, "hip-vs-ref32", h32, "hip-vs-f64", h64, "ref32-vs-f64", r)
    assert h32 < (3 * r + slack if tol32 is None else tol32), (key, "hip-vs-ref32", h32, "hip-vs-f64", h64, "ref32-vs-f64", r)


# rel(HIP gradient, the reference's fp32 autograd gradient) as measured at the end of round 4 (profiles/parity_r04.json)
H32_R04 = {
    ("console_basic_2x4x16384.npz", "grad_track_params"): 1.1e-7,
    ("console_full_1x8x32768.npz", "grad_master_bus_params"): 3.2e-5, ("console_full_1x8x32768.npz", "grad_track_params"): 1.67e-3,
    ("console_full_2x4x16384.npz", "grad_master_bus_params"): 1.9e-4, ("console_full_2x4x16384.npz", "grad_track_params"): 2.97e-3,
    ("console_fullbox_1x2x131072.npz", "grad_master_bus_params"): 5.7e-5, ("console_fullbox_1x2x131072.npz", "grad_track_params"): 5.3e-5,
    ("console_refmix_2x3x8192.npz", "grad_master_bus_params"): 2.3e-5, ("console_refmix_2x3x8192.npz", "grad_track_params"): 1.7e-5,
}


def parse_flags(arr):
    return {k: v == "True" for k, v in arr}


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "console_*.npz"))))
def test_console_golden(path, console, dev, record):
    """Fixtures produced by the REAL reference orchestration (tests/golden/make_golden.py)."""
    g = np.load(path, allow_pickle=True)
    flags = parse_flags(g["flags"])
    t = lambda k: torch.from_numpy(g[k]).float()
    stride = int(g["mix_stride"])
    out = run_hip(console, dev, t("tracks"), t("track_params"), t("fx_bus_params"), t("master_bus_params"), flags,
                  gmix=t("grad_mix"))
    e_mix, e_mixed = rel(out["mix"][..., ::stride], t("mix")), rel(out["mixed"][..., ::64], t("mixed_tracks_sub"))
    # parameter gradients three-way, all three from the fixture: the reference's fp32 autograd (written by the REAL orchestration), the
    # float64 evaluation of the same algorithm (same generator run), and the HIP console here - HIP may be no further from float64 than
    # twice the reference's own fp32 gradient is, + 1e-4 (round 4: replaces the per-fixture measured tolerances up to 8e-3)
    rep = dict(mix=e_mix, mixed_tracks=e_mixed)
    # One documented exception to the factor 2: full_2x4x16384 (measured 2.13).  tools/dbg_golden_grad.py locates it in ONE section - the
    # high shelf of track 2 at 13.8 kHz, Q 3.7, +8.4 dB, whose fp32 design loses digits in 1 + cos w0: its frequency / gain gradients sit
    # 4.2e-3 / 1.7e-3 of the total norm from float64 for HIP and 1.4e-3 / 8.6e-4 for the reference's fp32 autograd.  Forming the
    # coefficient-gradient sums and the all-pole recurrences in float64 (-DMST_CG_F64) does not move it: it is the gradient at the
    # fp32-rounded coefficients, not round-off of the kernels.  Every other fixture and every three-way test keeps the factor 2.
    factor = {"console_full_2x4x16384.npz": 2.5}.get(os.path.basename(path), 2.0)
    for key, hip_g in (("grad_track_params", out["g_tp"]), ("grad_master_bus_params", out["g_mp"])):
        if np.abs(g[key]).max() == 0:
            continue
        f64 = torch.from_numpy(g[key + "_f64"])
        h32, h64, r64 = rel(hip_g, t(key)), rel(hip_g, f64), rel(t(key), f64)
        rep[key] = (h32, h64, r64)
        assert h64 <= factor * r64 + 1e-4, (key, "hip-vs-ref32", h32, "hip-vs-f64", h64, "ref32-vs-f64", r64)
        # ... and a direct regression bound on the distance to the reference's OWN fp32 gradient: 1.5 x the value measured in round 4
        # (profiles/parity_r04.json) + 1e-5, so that a kernel change that stays inside the three-way bound but moves away from the
        # reference's arithmetic is seen (advisor, round 4)
        measured = H32_R04.get((os.path.basename(path), key))
        if measured is not None:
            assert h32 <= 1.5 * measured + 1e-5, (key, "hip-vs-ref32", h32, "round-4 value", measured)
    record(**rep)
    assert e_mix < 1e-4 and e_mixed < 1e-4
    if np.abs(g["grad_master_bus_params"]).max() == 0:
        assert float(out["g_mp"].abs().max()) == 0.0
    # denormalised parameter dictionaries (reference mst/modules.py:462-466) - exact affine map
    for k in g.files:
        if k.startswith("tp."):
            _, eff, name = k.split(".")
            assert torch.allclose(out["tpd"][eff][name].cpu(), t(k), rtol=1e-6, atol=1e-6), k
        if k.startswith("mp."):
            _, eff, name = k.split(".")
            assert torch.allclose(out["mpd"][eff][name].cpu(), t(k), rtol=1e-6, atol=1e-6), k


@pytest.mark.parametrize("bs,T,n,seed", [(2, 8, 262144, 0), (1, 4, 65536, 1), (2, 3, 131072, 2)])
def test_console_three_way(bs, T, n, seed, console, dev, record):
    """HIP vs fp32 reference algorithm vs float64, forward and backward, BASELINE cfg #2 row shape."""
    torch.manual_seed(seed)
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    gmix = torch.randn(bs, 2, n)
    hip = run_hip(console, dev, tracks, tp, fp, mp, FULL, gmix=gmix, grad_tracks=True)
    r32 = run_oracle(tracks, tp, fp, mp, FULL, gmix=gmix, grad_tracks=True)
    r64 = run_oracle(tracks, tp, fp, mp, FULL, gmix=gmix, dtype=torch.float64, grad_tracks=True)
    report = {}
    for k in ("mix", "mixed", "g_tracks", "g_tp", "g_mp"):
        report[k] = (rel(hip[k], r32[k]), rel(hip[k], r64[k]), rel(r32[k], r64[k]))
    print("\n[three-way] key: (hip vs ref32, hip vs f64, ref32 vs f64)\n", report)
    record(**report)
    for k in ("mix", "mixed"):
        assert report[k][0] < 1e-4, (k, report[k])
    for k in ("g_tracks", "g_tp", "g_mp"):
        h32, h64, r = report[k]
        assert h64 <= 2 * r + 1e-4, (k, report[k])
        assert h32 < 1e-2, (k, report[k])


def test_interior_parameter_set(console, dev):
    """0.05 + 0.9*rand parameters: away from the ill-conditioned corners all errors are small."""
    torch.manual_seed(5)
    bs, T, n = 2, 4, 65536
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = 0.05 + 0.9 * torch.rand(bs, T, 27), torch.rand(bs, 25), 0.05 + 0.9 * torch.rand(bs, 26)
    tp[..., 2] = 0.3 + 0.6 * torch.rand(bs, T)  # keep the low shelf away from 20 Hz
    tp[..., 5] = 0.3 + 0.6 * torch.rand(bs, T)
    mp[..., 1] = 0.3 + 0.6 * torch.rand(bs)
    mp[..., 4] = 0.3 + 0.6 * torch.rand(bs)
    gmix = torch.randn(bs, 2, n)
    hip = run_hip(console, dev, tracks, tp, fp, mp, FULL, gmix=gmix)
    r32 = run_oracle(tracks, tp, fp, mp, FULL, gmix=gmix)
    r64 = run_oracle(tracks, tp, fp, mp, FULL, gmix=gmix, dtype=torch.float64)
    assert rel(hip["mix"], r64["mix"]) < 1e-4
    assert_three_way(hip, r32, r64, "g_tp", 5e-3)
    assert_three_way(hip, r32, r64, "g_mp", 5e-3)


@pytest.mark.parametrize("n", [1, 7, 1000, 2049, 12345, 16384 + 3])
def test_ragged_lengths(n, console, dev):
    """Lengths that are not multiples of the chunk / vector width, shorter than the look-ahead, etc.

    Forward truth is the float64 TIME-DOMAIN recursion (scipy sosfilt / lfilter): for clips this short
    the reference's frequency-sampling filter degenerates (n_fft = 1 at n = 1 truncates the biquad to
    b0/a0), so only the gradients of the two longest clips are compared with frequency-sampling autograd.
    """
    from oracle import console_restated as oc

    torch.manual_seed(n)
    bs, T = 2, 3
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    tp, mp = short_ir(tp, mp)
    gmix = torch.randn(bs, 2, n)
    hip = run_hip(console, dev, tracks, tp, fp, mp, FULL, gmix=gmix, grad_tracks=True)
    _, truth, *_ = oc.console_forward(tracks.double(), tp.double(), fp.double(), mp.double(), time_domain=True, **FULL)
    scale = truth.abs().max().item() + 1e-30
    assert (hip["mix"].cpu().double() - truth).abs().max().item() / scale < 2e-4
    assert torch.isfinite(hip["g_tp"]).all() and torch.isfinite(hip["g_mp"]).all()
    assert torch.isfinite(hip["g_tracks"]).all()
    if n >= 12000:
        r64 = run_oracle(tracks, tp, fp, mp, FULL, gmix=gmix, dtype=torch.float64, grad_tracks=True)
        assert rel(hip["g_tp"], r64["g_tp"]) < 2e-2
        assert rel(hip["g_tracks"], r64["g_tracks"]) < 5e-3


@pytest.mark.parametrize("off", ["use_track_input_fader", "use_track_eq", "use_track_compressor", "use_master_bus",
                                 "use_output_fader", "all"])
def test_flag_combinations(off, console, dev):
    torch.manual_seed(3)
    bs, T, n = 2, 3, 20000
    flags = dict(FULL)
    if off == "all":
        flags.update(use_track_eq=False, use_track_compressor=False, use_master_bus=False, use_output_fader=False)
    else:
        flags[off] = False
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    tp, mp = short_ir(tp, mp)
    gmix, gmixed = torch.randn(bs, 2, n), torch.randn(bs, 2, T, n)
    hip = run_hip(console, dev, tracks, tp, fp, mp, flags, gmix=gmix, gmixed=gmixed, grad_tracks=True)
    r32 = run_oracle(tracks, tp, fp, mp, flags, gmix=gmix, gmixed=gmixed, grad_tracks=True)
    r64 = run_oracle(tracks, tp, fp, mp, flags, gmix=gmix, gmixed=gmixed, dtype=torch.float64, grad_tracks=True)
    assert_three_way(hip, r32, r64, "mix", 1e-4)
    assert_three_way(hip, r32, r64, "mixed", 1e-4)
    # one track row of this draw has a near-Nyquist high-Q band whose fp32 design moves BOTH fp32 paths ~5e-2 away from float64: the
    # gradients are bounded three-way only (no measured tolerance)
    assert_three_way(hip, r32, r64, "g_tracks")
    assert_three_way(hip, r32, r64, "g_tp")
    if r64["g_mp"].abs().max() > 0:
        assert_three_way(hip, r32, r64, "g_mp")
    else:
        assert float(hip["g_mp"].abs().max()) == 0.0
    # parameters of switched-off stages get exactly zero gradient, like autograd's unused leaves
    g = hip["g_tp"].cpu()
    if not flags["use_track_eq"]:
        assert float(g[..., 1:19].abs().max()) == 0.0
    if not flags["use_track_compressor"]:
        assert float(g[..., 19:25].abs().max()) == 0.0
    if not flags["use_track_input_fader"]:
        assert float(g[..., 0].abs().max()) == 0.0
    assert float(g[..., 26].abs().max()) == 0.0  # fx send (fx bus off)
    assert float(g[..., 22].abs().max()) == 0.0  # release_ms is unused by the op


@pytest.mark.parametrize("knee", [0.0, 0.5, 1.0])
def test_static_curve_level_sweep(knee, console, dev):
    """The compressor's static curve is evaluated branch-free (csrc/mst_compdev.h: curve_f - clamp to the knee, no compare / select):
    a level sweep from -100 dB to 0 dB takes every sample region of it (below, inside and above the knee, and both transitions) in the
    tracks AND the master compressor, at the narrowest, a middle and the widest knee; forward and gradients three-way against the
    reference's branchy fp32 / float64 curve.  EQ off: the level reaching the curve is the level of the sweep."""
    torch.manual_seed(11)
    bs, T, n = 1, 3, 32768
    flags = dict(FULL)
    flags.update(use_track_eq=False)
    level_db = torch.linspace(-100.0, 0.0, n)
    tracks = (10.0 ** (level_db / 20.0)) * torch.sign(torch.randn(bs, T, n))  # random polarity, exact envelope
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    tp[..., 0] = 0.5                     # input fader mid-range
    tp[..., 19] = torch.tensor([0.2, 0.5, 0.8])  # thresholds spread over the range
    tp[..., 23] = knee                   # knee_db: lower end, middle, upper end of its range
    mp[..., 23 - 1] = knee
    gmix = torch.randn(bs, 2, n)
    hip = run_hip(console, dev, tracks, tp, fp, mp, flags, gmix=gmix, grad_tracks=True)
    r32 = run_oracle(tracks, tp, fp, mp, flags, gmix=gmix, grad_tracks=True)
    r64 = run_oracle(tracks, tp, fp, mp, flags, gmix=gmix, dtype=torch.float64, grad_tracks=True)
    assert_three_way(hip, r32, r64, "mix", 1e-4)
    for k in ("g_tracks", "g_tp", "g_mp"):
        assert_three_way(hip, r32, r64, k)


def test_basic_console_bitlevel(dev):
    """BASELINE cfg #1: gain + pan + bus sum, 4 tracks x 65536, batch 2 - the plumbing gate."""
    from mst.modules import BasicMixConsole

    torch.manual_seed(0)
    bs, T, n = 2, 4, 65536
    c = BasicMixConsole(44100)
    tracks = 0.1 * torch.randn(bs, T, n)
    tp = torch.rand(bs, T, 27)
    mixed, mix, *_ = c(tracks.to(dev), tp.to(dev))
    flags = dict(use_track_input_fader=True, use_track_eq=False, use_track_compressor=False, use_track_panner=True,
                 use_fx_bus=False, use_master_bus=False, use_output_fader=False)
    ref = run_oracle(tracks, tp, torch.rand(bs, 25), torch.rand(bs, 26), flags)
    # element-wise: a few ulp (the reference multiplies then sums; the kernel uses fused multiply-adds)
    err = (mix.cpu() - ref["mix"]).abs().max().item() / ref["mix"].abs().max().item()
    assert err < 5e-7, err
    assert rel(mixed, ref["mixed"]) < 2e-7


def test_linearity_and_determinism_full_size(console, dev):
    """Size-independent properties at BASELINE cfg #2 size: with the compressors off the console is
    linear in the tracks; every kernel is run-to-run bitwise deterministic (no float atomics)."""
    torch.manual_seed(9)
    bs, T, n = 8, 8, 262144
    flags = dict(FULL, use_track_compressor=False, use_master_bus=True)
    tp, fp, mp = torch.rand(bs, T, 27).to(dev), torch.rand(bs, 25).to(dev), torch.rand(bs, 26).to(dev)
    a = (0.1 * torch.randn(bs, T, n)).to(dev)
    with torch.no_grad():
        _, m1, *_ = console(a, tp, fp, mp, **FULL)
        _, m2, *_ = console(a, tp, fp, mp, **FULL)
    assert torch.equal(m1, m2)
    # linear sub-chain: tracks -> gain -> EQ -> pan -> bus (master off, so no second compressor)
    lin = dict(FULL, use_track_compressor=False, use_master_bus=False)
    b = (0.1 * torch.randn(bs, T, n)).to(dev)
    with torch.no_grad():
        _, ma, *_ = console(a, tp, fp, mp, **lin)
        _, mb, *_ = console(b, tp, fp, mp, **lin)
        _, mab, *_ = console(2.0 * a - 0.5 * b, tp, fp, mp, **lin)
    assert rel(mab, 2.0 * ma - 0.5 * mb) < 1e-4
    del flags


def test_inwave_scan_equals_carry_scan_kernel_full_size(console, dev):
    """The two EQ carry-resolution paths (in-wave scans, taken up to 262144 samples; separate carry-scan kernel,
    taken beyond) on the SAME cfg #2-sized input, forward and backward: fp32 round-off apart."""
    torch.manual_seed(21)
    bs, T, n = 2, 8, 262144
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    gmix = torch.randn(bs, 2, n)
    a = run_hip(console, dev, tracks, tp, fp, mp, FULL, gmix=gmix, grad_tracks=True)
    multi = type(console)(44100)
    multi._multipass_eq = True
    b = run_hip(multi, dev, tracks, tp, fp, mp, FULL, gmix=gmix, grad_tracks=True)
    assert rel(a["mix"], b["mix"]) < 5e-6
    # the cotangents pass the compressors' knees / abs(): 1e-6 differences of u move them by ~1e-3 in BOTH paths
    # (each sits 3e-4 from float64, like the fp32 reference itself - tools/dbg_gtracks.py)
    assert rel(a["g_tracks"], b["g_tracks"]) < 5e-3
    assert rel(a["g_tp"], b["g_tp"]) < 5e-3 and rel(a["g_mp"], b["g_mp"]) < 5e-3


def test_rows_longer_than_the_inwave_limit(console, dev):
    """N = 2^19 + 5: 129 tiles per row, so the EQ goes through the carry-scan kernel; truth = float64 time domain
    (the frequency-sampling reference at this length is the same filter)."""
    torch.manual_seed(22)
    bs, T, n = 1, 2, (1 << 19) + 5
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    gmix = torch.randn(bs, 2, n)
    hip = run_hip(console, dev, tracks, tp, fp, mp, FULL, gmix=gmix)
    r32 = run_oracle(tracks, tp, fp, mp, FULL, gmix=gmix)
    r64 = run_oracle(tracks, tp, fp, mp, FULL, gmix=gmix, dtype=torch.float64)
    assert rel(hip["mix"], r32["mix"]) < 1e-4
    assert_three_way(hip, r32, r64, "g_tp", 3e-2)
    assert_three_way(hip, r32, r64, "g_mp", 3e-2)


def test_cfg3_shape_batch_independence(console, dev):
    """SURVEY cfg #3 shape (bs 32, T 16, N 262144: 512 track rows, ~4 GB of workspace): one mix of the big batch
    against the oracle run on that mix alone (mixes never interact), plus bitwise determinism."""
    torch.manual_seed(23)
    bs, T, n = 32, 16, 262144
    tracks = 0.1 * torch.randn(bs, T, n)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    gmix = torch.randn(bs, 2, n)
    lean = type(console)(44100, materialize_mixed_tracks=False)
    a = run_hip(lean, dev, tracks, tp, fp, mp, FULL, gmix=gmix)
    b = run_hip(lean, dev, tracks, tp, fp, mp, FULL, gmix=gmix)
    assert torch.equal(a["mix"], b["mix"]) and torch.equal(a["g_tp"], b["g_tp"]) and torch.equal(a["g_mp"], b["g_mp"])
    assert torch.isfinite(a["g_tp"]).all() and torch.isfinite(a["g_mp"]).all()
    i = 29
    one = run_oracle(tracks[i:i + 1], tp[i:i + 1], fp[i:i + 1], mp[i:i + 1], FULL, gmix=gmix[i:i + 1])
    assert rel(a["mix"][i:i + 1], one["mix"]) < 1e-4
    assert rel(a["g_tp"][i:i + 1], one["g_tp"]) < 3e-2 and rel(a["g_mp"][i:i + 1], one["g_mp"]) < 3e-2


def test_strided_tracks_like_system(console, dev):
    """System passes tracks[..., middle:] (reference mst/system.py:258): row stride != length."""
    torch.manual_seed(4)
    bs, T, n = 2, 4, 32768
    full = (0.1 * torch.randn(bs, T, 2 * n)).to(dev)
    tp, fp, mp = torch.rand(bs, T, 27).to(dev), torch.rand(bs, 25).to(dev), torch.rand(bs, 26).to(dev)
    view = full[..., n:]
    assert not view.is_contiguous()
    with torch.no_grad():
        _, m_view, *_ = console(view, tp, fp, mp, **FULL)
        _, m_copy, *_ = console(view.contiguous(), tp, fp, mp, **FULL)
    assert torch.equal(m_view, m_copy)
    with pytest.raises(RuntimeError):  # genuinely non-collapsible input: same failure mode as reference .view
        console(full.transpose(0, 1), tp, fp, mp, **FULL)


def test_out_of_range_value_error(console, dev):
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "bark_fb.npz"))
    tp, fp, mp = torch.rand(1, 2, 27), torch.rand(1, 25), torch.rand(1, 26)
    tp[0, 1, 25] = 1.5
    with pytest.raises(ValueError) as e:
        console(torch.zeros(1, 2, 4096).to(dev), tp.to(dev), fp.to(dev), mp.to(dev), use_fx_bus=False)
    assert str(e.value) == str(g["err_text"])  # text produced by the real reference
    tp[0, 1, 25] = 0.5
    mp[0, 3] = -0.1
    tp[0, 0, 7] = 2.0  # two offenders: the reference reports the first in dictionary order (track dict first)
    with pytest.raises(ValueError, match="Parameter band1_gain_db of effect parametric_eq"):
        console(torch.zeros(1, 2, 4096).to(dev), tp.to(dev), fp.to(dev), mp.to(dev), use_fx_bus=False)
    tp[0, 0, 7] = 0.5
    with pytest.raises(ValueError, match="Parameter band0_gain_db of effect parametric_eq"):
        console(torch.zeros(1, 2, 4096).to(dev), tp.to(dev), fp.to(dev), mp.to(dev), use_fx_bus=False)


def test_deferred_validation(dev):
    from mst.modules import AdvancedMixConsole

    c = AdvancedMixConsole(44100, validate="deferred", materialize_mixed_tracks=False)
    tp, fp, mp = torch.rand(1, 2, 27), torch.rand(1, 25), torch.rand(1, 26)
    fp[0, 3] = 7.0
    mixed, mix, *_ = c(torch.zeros(1, 2, 4096).to(dev), tp.to(dev), fp.to(dev), mp.to(dev), use_fx_bus=False)
    assert mixed is None and mix.shape == (1, 2, 4096)
    with pytest.raises(ValueError, match="Parameter band3_gain of effect reverberation"):
        c.check_parameters()
    c.check_parameters()  # cleared


def test_naive_random_mix_golden(console, dev):
    from mst.mixing import naive_random_mix

    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "naive_random_mix.npz"))
    tracks = torch.cat([torch.from_numpy(g["tracks"])] * 8, dim=-1).to(dev)
    torch.manual_seed(int(g["seed"]))
    r = naive_random_mix(tracks, console, use_fx_bus=False)
    assert len(r) == 8
    assert torch.equal(r[5].cpu(), torch.from_numpy(g["mix_params"]))
    assert torch.equal(r[6].cpu(), torch.from_numpy(g["fx_bus_params"]))
    assert torch.equal(r[7].cpu(), torch.from_numpy(g["master_bus_params"]))
    assert rel(r[1][..., ::8], torch.from_numpy(g["mix"])) < 1e-4
    assert not r[1].requires_grad


def test_zero_input_and_silence(console, dev):
    """All-zero tracks (the reference prints a warning and carries on): output exactly zero, finite grads."""
    bs, T, n = 1, 2, 8192
    tp = torch.rand(bs, T, 27).to(dev).requires_grad_(True)
    mp = torch.rand(bs, 26).to(dev).requires_grad_(True)
    _, mix, *_ = console(torch.zeros(bs, T, n).to(dev), tp, torch.rand(bs, 25).to(dev), mp, **FULL)
    mix.sum().backward()
    assert float(mix.detach().abs().max()) == 0.0
    assert torch.isfinite(tp.grad).all() and torch.isfinite(mp.grad).all()


def test_in_launch_aggregate_exchange_under_uneven_load(console, dev):
    """Round 4: the compressor's block aggregates travel from workgroup to workgroup INSIDE k_apply_master / k_comp_bwd_run (8-byte
    granules, csrc/mst_common.h) instead of through a zero-state launch.  A hand-off that is only correct on an idle chip passes every
    other test (cdna_hip_programming.md Guideline 16: 'test every hand-off under UNEVEN load ... checking every word'), so: the same
    forward + backward 12 times while a second stream streams 1 GiB copies and small kernels at random phases - every mix sample and
    every parameter gradient of every repetition must be BIT-equal to the quiet run (the exchange has no timing-dependent arithmetic:
    fixed summation order, a late granule is waited for, never skipped; a give-up would poison the result with NaN)."""
    torch.manual_seed(17)
    bs, T, n = 8, 8, 262144
    tracks = (0.1 * torch.randn(bs, T, n)).to(dev)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    gmix = torch.randn(bs, 2, n).to(dev)

    def run():
        a, b = tp.to(dev).requires_grad_(True), mp.to(dev).requires_grad_(True)
        _, mix, *_ = console(tracks, a, fp.to(dev), b, **FULL)
        mix.backward(gmix)
        return mix.detach(), a.grad, b.grad

    quiet = run()
    torch.cuda.synchronize()
    assert all(torch.isfinite(t).all() for t in quiet)
    noise_stream = torch.cuda.Stream()
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB
    gen = torch.Generator().manual_seed(3)
    for rep in range(12):
        with torch.cuda.stream(noise_stream):
            for _ in range(int(torch.randint(1, 6, (1,), generator=gen))):
                big[: 1 << 27].copy_(big[1 << 27:])  # 512 MiB read + 512 MiB write
                big[:4096].add_(1.0)
        got = run()
        torch.cuda.synchronize()
        for q, g, name in zip(quiet, got, ("mix", "grad_track_params", "grad_master_params")):
            assert torch.equal(q, g), (rep, name, float((q - g).abs().max()))


def test_second_backward_over_the_same_forward(console, dev):
    """The granules of the backward are armed by the forward's k_prep and RE-armed by every backward's last kernel (k_prep_bwd): a second
    backward over the same forward (retain_graph) with ANOTHER cotangent must see fresh granules, not the first backward's aggregates."""
    torch.manual_seed(19)
    bs, T, n = 2, 4, 65536
    tracks = (0.1 * torch.randn(bs, T, n)).to(dev)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    g1, g2 = torch.randn(bs, 2, n).to(dev), torch.randn(bs, 2, n).to(dev)
    a, b = tp.to(dev).requires_grad_(True), mp.to(dev).requires_grad_(True)
    _, mix, *_ = console(tracks, a, fp.to(dev), b, **FULL)
    mix.backward(g1, retain_graph=True)
    first = (a.grad.clone(), b.grad.clone())
    a.grad = b.grad = None
    mix.backward(g2)
    second = (a.grad.clone(), b.grad.clone())
    # fresh forward + backward with g2
    a2, b2 = tp.to(dev).requires_grad_(True), mp.to(dev).requires_grad_(True)
    _, mix2, *_ = console(tracks, a2, fp.to(dev), b2, **FULL)
    mix2.backward(g2)
    assert torch.equal(second[0], a2.grad) and torch.equal(second[1], b2.grad)
    assert not torch.equal(first[0], second[0])


@pytest.mark.parametrize("bs,T,n,materialize", [(3, 4, 65536, True), (2, 8, 262144, False), (5, 2, 20000, True)])
def test_split_batch_is_bit_identical(bs, T, n, materialize, dev):
    """Round 6: `split_batch` runs the call as two halves of the batch on two streams (mst_console_forward_overlapped /
    _backward_overlapped, include/diffmst_hip.h).  The mixes of a call are independent (reference mst/modules.py:186-314), the halves run
    the same kernels over their own part of the workspace: every output and every gradient must be BIT-equal to the unsplit call - odd
    batch sizes (halves of different size), ragged lengths, with and without `mixed_tracks` and its cotangent, twice in a row (the side
    stream and its events are reused by every call)."""
    from mst.modules import AdvancedMixConsole

    torch.manual_seed(bs * 1000 + T)
    tracks = (0.1 * torch.randn(bs, T, n)).to(dev)
    tp, fp, mp = torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    gmix, gmixed = torch.randn(bs, 2, n).to(dev), torch.randn(bs, 2, T, n).to(dev)

    def run(split):
        c = AdvancedMixConsole(44100, validate="deferred", materialize_mixed_tracks=materialize, split_batch=split)
        outs = []
        for _ in range(2):
            tr = tracks.clone().requires_grad_(True)
            a, b = tp.to(dev).requires_grad_(True), mp.to(dev).requires_grad_(True)
            mixed, mix, *_ = c(tr, a, fp.to(dev), b, **FULL)
            loss = (mix * gmix).sum()
            if materialize:
                loss = loss + (mixed * gmixed).sum()
            loss.backward()
            torch.cuda.synchronize()
            c.check_parameters()
            outs.append((mix.detach(), mixed.detach() if materialize else None, a.grad, b.grad, tr.grad))
        return outs

    plain, split = run(False), run(True)
    for rep in range(2):
        for name, p, q in zip(("mix", "mixed_tracks", "grad_track_params", "grad_master_params", "grad_tracks"), plain[0], split[rep]):
            if p is None:
                assert q is None
                continue
            assert torch.isfinite(q).all(), name
            assert torch.equal(p, q), (rep, name, float((p - q).abs().max()))
