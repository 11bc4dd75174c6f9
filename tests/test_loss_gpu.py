"""GPU parity tests: MR-STFT loss and peak normalisation against the oracle.

Tolerance: the loss value (a mean over >1e6 well-conditioned terms) must agree to 1e-5 relative; its
gradient is checked three-way because the log-magnitude term is ill-conditioned on the two
reflect-padded edge frames (real-even frames -> many near-zero bins, d log|X| ~ 1/|X|): the fp32
reference itself sits ~1e-4 from float64 there."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from util import rel

RES = ((512, 256, 512), (2048, 1024, 2048), (8192, 4096, 8192))  # reference configs/models/naive.yaml:57-68


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from mst import _hip

    _hip.lib()
    return torch.device("cuda:0")


def make_loss(res=RES, **kw):
    from mst.loss import MultiResolutionSTFTLoss

    return MultiResolutionSTFTLoss(fft_sizes=[r[0] for r in res], hop_sizes=[r[1] for r in res],
                                   win_lengths=[r[2] for r in res], **kw)


@pytest.mark.parametrize("bs,n,kw,seeds", [
    (2, 65536, {}, 5),
    (8, 262144, {}, 3),                                                   # BASELINE cfg #2 loss shape
    (2, 131072, dict(w_sc=0.0, w_log_mag=1.0, w_lin_mag=1.0), 5),          # evaluation instance mst/system.py:61-69
    (1, 50000, dict(sc_per_example=False), 5),                            # pre-0.4.0 global spectral convergence
])
def test_mrstft_three_way(bs, n, kw, seeds, dev, record):
    """Loss to 1e-5 of float64; gradient three-way over several seeds.

    d log|X| / dX ~ 1/|X|: the rel-L2 error of the log-magnitude gradient is carried by a handful of near-zero bins of the two
    reflect-padded edge frames (real-even frames: X[k] = (-1)^k R[k] with R real, so |X| crosses zero between bins), for BOTH
    fp32 implementations.  Which of the two lands closer to float64 on one draw is a coin toss with a heavy tail - per seed and
    resolution the ratio (HIP distance) / (fp32 reference distance) ranges 0.08 .. 4.1 (tools/dbg_logmag.py, round 3) - so the
    bound is on the statistic that is stable: the MEDIAN ratio over the seeds must be <= 1.5 (HIP is as close to float64 as the
    path it replaces), every single draw within 10x + 2e-2 (measured 0.08x .. 19x: besides the 1/|X| tail a bin whose
    |X|^2 sits within rounding of the 1e-8 clamp switches its whole 1/|X| ~ 1e4 cotangent on or off - a step, for either path), and the smooth terms are pinned to 5e-6 separately below."""
    from oracle import loss_restated as ol

    ratios, worst = [], None
    for s in range(seeds):
        torch.manual_seed(bs * 7 + n + 1000 * s)
        x = 0.3 * torch.randn(bs, 2, n)
        y = 0.5 * x + 0.2 * torch.randn(bs, 2, n)
        xd = x.to(dev).requires_grad_(True)
        loss = make_loss(**kw)(xd, y.to(dev))
        loss.backward()
        outs = {}
        for dt in (torch.float32, torch.float64):
            xo = x.clone().to(dt).requires_grad_(True)
            lo = ol.mrstft_loss(xo, y.to(dt), RES, **kw)
            lo.backward()
            outs[dt] = (lo.item(), xo.grad)
        l32, g32 = outs[torch.float32]
        l64, g64 = outs[torch.float64]
        assert abs(loss.item() - l64) / l64 < 1e-5, (loss.item(), l32, l64)
        h32, h64, r = rel(xd.grad, g32), rel(xd.grad, g64), rel(g32, g64)
        ratios.append(h64 / r)
        if worst is None or h64 / r > worst[0]:
            worst = (h64 / r, h32, h64, r, abs(loss.item() - l64) / l64)
        print(f"\n[mrstft {bs}x2x{n} {kw} seed {s}] loss hip {loss.item():.7f} ref32 {l32:.7f} f64 {l64:.7f}; grad hip-ref32 {h32:.2e} hip-f64 {h64:.2e} ref32-f64 {r:.2e}")
        assert h64 <= 10 * r + 2e-2 and h32 <= 2 * (h64 + r)
    med = sorted(ratios)[len(ratios) // 2]
    record(loss_rel_err_vs_f64=worst[4], grad=worst[1:4], ratios_hip_over_ref32=ratios, median_ratio=med)
    assert med <= 1.5, ratios


@pytest.mark.parametrize("bs,n,seed", [(2, 65536, 21), (8, 262144, 22)])
def test_mrstft_log_magnitude_adjoint_away_from_the_clamp(bs, n, seed, dev, record):
    """The sign() / |X| adjoint of the log-magnitude term with a deterministic per-draw bound (round-3 review, weak 3): once the two
    things that make test_mrstft_three_way a statistic are out of the picture, every single draw must sit within the three-way bound
    (HIP no further from float64 than twice the fp32 reference path, + 1e-4) - no median, no 10x allowance.
      * the reflect-padded edge frames (real-even spectra: |X| crosses zero between bins) see silence - the signals are zero over the
        first and last 8192 samples, so every frame that touches the padding is all-zero, sits BELOW the 1e-8 clamp for both
        implementations and contributes exactly nothing;
      * no remaining bin is within a factor 4 of the clamp, where a cotangent of ~1e4 switches on or off with the last bit of |X|^2 -
        checked here in float64 on every bin of every resolution, for prediction and target (a self-check of the construction, not a
        mask applied to the result: with complex Gaussian spectra of variance >= 17 such a bin has probability 2e-9).
    What is left is the plain arithmetic of cotangent, inverse transform, window and overlap-add - and the 1 / |X| weighting of the
    fp32 transform's own round-off: for a complex Gaussian spectrum E[1 / |X|^2] diverges logarithmically, so the rel-L2 error of this
    gradient is ~1e-3 for ANY fp32 evaluation (measured: HIP 1.2e-3, the reference's fp32 path 8e-4 at 2 x 2 x 65536) and an absolute
    1e-4 is not a property fp32 has here; the smooth terms (test_mrstft_well_conditioned_terms) pin the linear part to 5e-6."""
    from oracle import loss_restated as ol

    # (the real-valued DC / Nyquist bins are chi-square with ONE degree of freedom: ~5e-5 of them fall into the band, i.e. a handful at
    # 8 x 2 x 262144 - so the first seed of a short deterministic sequence that has none is used)
    for attempt in range(16):
        torch.manual_seed(seed + 100 * attempt)
        x = 0.3 * torch.randn(bs, 2, n)
        y = 0.5 * x + 0.2 * torch.randn(bs, 2, n)
        for t in (x, y):
            t[..., :8192] = 0.0
            t[..., -8192:] = 0.0
        near = 0
        for n_fft, hop, win in RES:
            for t in (x, y):
                X = torch.stft(t.double().reshape(-1, n), n_fft, hop, win, torch.hann_window(win, dtype=torch.float64), return_complex=True)
                p2 = X.real**2 + X.imag**2
                near += int(((p2 > 0.25e-8) & (p2 < 4e-8)).sum())
        if near == 0:
            break
    assert near == 0, f"{near} bins within a factor 4 of the clamp after 16 seeds"
    kw = dict(w_sc=0.0, w_log_mag=1.0)
    xd = x.to(dev).requires_grad_(True)
    loss = make_loss(**kw)(xd, y.to(dev))
    loss.backward()
    outs = {}
    for dt in (torch.float32, torch.float64):
        xo = x.clone().to(dt).requires_grad_(True)
        lo = ol.mrstft_loss(xo, y.to(dt), RES, **kw)
        lo.backward()
        outs[dt] = (lo.item(), xo.grad)
    l64, g64 = outs[torch.float64]
    h64, r64 = rel(xd.grad, g64), rel(outs[torch.float32][1], g64)
    e_loss = abs(loss.item() - l64) / l64
    print(f"\n[log-magnitude adjoint, {bs}x2x{n}] loss {e_loss:.2e}; gradient HIP vs f64 {h64:.2e} (fp32 oracle vs f64 {r64:.2e})")
    record(loss=e_loss, grad_hip_vs_f64=h64, grad_ref32_vs_f64=r64)
    assert e_loss < 1e-5
    assert h64 <= 2 * r64 + 1e-4, (h64, r64)
    assert float(xd.grad[..., :4096].abs().max()) == 0.0 and float(xd.grad[..., -4096:].abs().max()) == 0.0  # silence below the clamp: no gradient


def _ill_posed_atoms(x64_row, res, tau_rel):
    """The time-domain atoms d X[t, k] / d x (real and imaginary part) of every bin of ONE row whose magnitude is below tau_rel x the
    row's rms bin magnitude at that resolution - the bins where the sign() / |X| adjoint of the log-magnitude term amplifies the
    transform's own round-off by 1 / |X| in ANY fp32 evaluation - plus every bin within a factor 4 of the 1e-8 clamp.  Returns
    (B (n, 2 * bins) float64, number of bins).  Reflect padding is folded back onto the row, as torch.stft(center=True) does."""
    n = x64_row.numel()
    cols, count = [], 0
    for n_fft, hop, win in res:
        w = torch.hann_window(win, dtype=torch.float64)
        X = torch.stft(x64_row[None], n_fft, hop, win, w, return_complex=True)[0]  # (bins, frames)
        p2 = X.real**2 + X.imag**2
        tau2 = tau_rel**2 * p2.mean()
        # (bins far BELOW the 1e-8 clamp - silence - have exactly zero cotangent in every implementation: nothing to project out)
        bad = ((p2 < tau2) & (p2 > 0.25e-8)) | ((p2 > 0.25e-8) & (p2 < 4e-8))
        ks, ts = torch.nonzero(bad, as_tuple=True)
        if ks.numel() == 0:
            continue
        m = torch.arange(n_fft, dtype=torch.float64)
        pos = torch.arange(n_fft)
        for k, t in zip(ks.tolist(), ts.tolist()):
            idx = t * hop + pos - n_fft // 2           # sample index before reflection
            idx = torch.where(idx < 0, -idx, idx)
            idx = torch.where(idx >= n, 2 * (n - 1) - idx, idx)
            ang = 2.0 * math.pi * k * m / n_fft
            for comp in (w * torch.cos(ang), -w * torch.sin(ang)):
                a = torch.zeros(n, dtype=torch.float64)
                a.index_add_(0, idx, comp)
                cols.append(a)
        count += ks.numel()
    return (torch.stack(cols, 1) if cols else torch.zeros(n, 0, dtype=torch.float64)), count


def _project_out(d_row, B):
    """d_row (n,) minus its least-squares projection onto the columns of B (n, m).  The atoms are far from independent (DC / Nyquist bins
    have no imaginary part, frame 0's spectrum is real, neighbouring bins of one frame overlap): the projector is formed from the
    eigen-decomposition of the Gram matrix with the directions below 1e-12 of the largest eigenvalue dropped - LAPACK's least-squares
    drivers either cut the rank at ~60 of ~1300 columns (gelsy, default tolerance) or fail to converge on some draws (gelsd)."""
    if not B.shape[1]:
        return d_row
    lam, V = torch.linalg.eigh(B.T @ B)
    keep = lam > 1e-12 * lam[-1]
    c = (V[:, keep].T @ (B.T @ d_row)) / lam[keep]
    return d_row - B @ (V[:, keep] @ c)


@pytest.mark.parametrize("case,bs,n,seed", [("tracking", 2, 32768, 31), ("tracking", 1, 65536, 33), ("independent", 2, 32768, 34),
                                            ("independent", 1, 65536, 35), ("frame0", 2, 32768, 36)])
def test_mrstft_gradient_every_draw_off_the_ill_posed_bins(case, bs, n, seed, dev, record):
    """A deterministic bound on EVERY draw of the full MR-STFT gradient (round-5 review, weak 3 / item 5b) - no median, no 10x allowance.

    The gradient is  sum_bins c[t, k] * atom[t, k]  with atom = d X[t, k] / d x and, for the log-magnitude term, c ~ sign(.) / |X|: at a
    bin with |X| << the row's level BOTH fp32 paths carry the transform's round-off amplified by 1 / |X| (and a bin within rounding of
    the 1e-8 clamp switches a ~1e4 cotangent on or off), so their error lives in the span of those bins' atoms.  That span is removed
    - by a least-squares projection in float64, the SAME bins (chosen from the float64 spectra of the prediction: |X| < 0.05 x the row's
    rms bin magnitude, ~0.25 % of the bins) for HIP and for the fp32 reference - and what is left must satisfy the three-way bound of
    every other gradient test: HIP no further from float64 than twice the fp32 reference, + 1e-5.  The un-projected numbers are recorded
    next to them: the projection takes the fp32 reference's own distance down by the same factor, which is the evidence that the
    complement is where the ill-posedness lives.
      tracking     target = 0.5 x + noise (the other gradient tests' construction)
      independent  target independent of the prediction: the saved-magnitude path of the backward (the forward keeps |Y| only) cannot
                   lean on Y ~ X
      frame0       energy only in the first 2048 samples: the gradient flows through frame 0 of every row and resolution, whose
                   reflect-padded real-even spectrum the backward projects onto the real axis, and frames 1..; everything else silent"""
    from oracle import loss_restated as ol

    torch.manual_seed(seed)
    x = 0.3 * torch.randn(bs, 2, n)
    y = 0.5 * x + 0.2 * torch.randn(bs, 2, n) if case != "independent" else 0.3 * torch.randn(bs, 2, n)
    if case == "frame0":
        for t in (x, y):
            t[..., 2048:] = 0.0
    xd = x.to(dev).requires_grad_(True)
    loss = make_loss()(xd, y.to(dev))
    loss.backward()
    outs = {}
    for dt in (torch.float32, torch.float64):
        xo = x.clone().to(dt).requires_grad_(True)
        lo = ol.mrstft_loss(xo, y.to(dt), RES)
        lo.backward()
        outs[dt] = (lo.item(), xo.grad.double().reshape(-1, n))
    l64, g64 = outs[torch.float64]
    g32 = outs[torch.float32][1]
    gh = xd.grad.detach().cpu().double().reshape(-1, n)
    threads = torch.get_num_threads()
    torch.set_num_threads(min(16, threads))  # the SVD of a tall 32768 x 1400 matrix does not scale to the GPU box's 128 threads
    # row by row (one row's atoms: n x ~2 % of n doubles - 0.4 GB at n = 32768, 1.4 GB at 65536 - are dropped before the next row's are built)
    nb, sq_h, sq_r = 0, 0.0, 0.0
    for i, row in enumerate(x.double().reshape(-1, n)):
        B, c = _ill_posed_atoms(row, RES, 0.05)
        nb += c
        sq_h += _project_out(gh[i] - g64[i], B).norm().item() ** 2
        sq_r += _project_out(g32[i] - g64[i], B).norm().item() ** 2
        del B
    torch.set_num_threads(threads)
    total_bins = x[..., 0].numel() * sum((1 + n // r[1]) * (r[0] // 2 + 1) for r in RES)
    norm = g64.norm().item()
    raw_h, raw_r = (gh - g64).norm().item() / norm, (g32 - g64).norm().item() / norm
    h, r = math.sqrt(sq_h) / norm, math.sqrt(sq_r) / norm
    e_loss = abs(loss.item() - l64) / l64
    print(f"\n[mrstft gradient off the ill-posed bins, {case} {bs}x2x{n} seed {seed}] {nb} of {total_bins} bins projected out "
          f"({100.0 * nb / total_bins:.2f} %); HIP vs f64 {h:.2e} (raw {raw_h:.2e}), fp32 reference vs f64 {r:.2e} (raw {raw_r:.2e}); loss {e_loss:.1e}")
    record(bins_projected=nb, bins=total_bins, hip_vs_f64=h, ref32_vs_f64=r, hip_vs_f64_raw=raw_h, ref32_vs_f64_raw=raw_r, loss=e_loss)
    assert e_loss < 1e-5
    assert nb < 0.01 * total_bins
    assert h <= 2 * r + 1e-5, (h, r, raw_h, raw_r)
    if case == "frame0":  # silence (below the clamp for both) beyond the reach of the frames that hold the burst
        assert float(xd.grad[..., 2048 + 8192:].abs().max()) == 0.0


@pytest.mark.parametrize("kw", [dict(w_sc=1.0, w_log_mag=0.0), dict(w_sc=1.0, w_log_mag=0.0, sc_per_example=False)])
def test_mrstft_well_conditioned_terms(kw, dev):
    """The spectral-convergence term is smooth (no 1/|X| factor, no sign()): its gradient pins the
    FFT / Hermitian split / adjoint / overlap-add machinery to fp32 round-off."""
    from oracle import loss_restated as ol

    torch.manual_seed(11)
    bs, n = 2, 65536
    x = 0.3 * torch.randn(bs, 2, n)
    y = 0.5 * x + 0.2 * torch.randn(bs, 2, n)
    xd = x.to(dev).requires_grad_(True)
    loss = make_loss(**kw)(xd, y.to(dev))
    loss.backward()
    xo = x.double().requires_grad_(True)
    lo = ol.mrstft_loss(xo, y.double(), RES, **kw)
    lo.backward()
    assert abs(loss.item() - lo.item()) / lo.item() < 1e-5
    assert rel(xd.grad, xo.grad) < 5e-6


@pytest.mark.parametrize("blocks,bs", [(4, 1), (5, 1), (6, 1), (7, 1), (9, 1), (12, 1), (17, 1), (31, 1), (16, 20), (33, 70)])
def test_mrstft_strip_layouts(blocks, bs, dev):
    """Row lengths of `blocks` 8192-point hops: every one cuts the rows into a different set of strips (seam strips of 2 frames
    with a longer last one for 8192, halo strips for 512 / 2048), and the seam halves parked by the 8192 launch must be picked up at
    exactly the right blocks by the next one.  Smooth term only, so the gradient is pinned to fp32 round-off."""
    from oracle import loss_restated as ol

    torch.manual_seed(blocks)
    n = blocks * 4096
    x = 0.3 * torch.randn(bs, 2, n)  # bs 20 / 70: 40 / 140 rows, where the 8192-point strips are lengthened to fit one round
    y = 0.5 * x + 0.2 * torch.randn(bs, 2, n)
    kw = dict(w_sc=1.0, w_log_mag=0.0)
    xd = x.to(dev).requires_grad_(True)
    loss = make_loss(**kw)(xd, y.to(dev))
    loss.backward()
    xo = x.double().requires_grad_(True)
    lo = ol.mrstft_loss(xo, y.double(), RES, **kw)
    lo.backward()
    assert abs(loss.item() - lo.item()) / lo.item() < 1e-5
    assert rel(xd.grad, xo.grad) < 5e-6
    again = x.to(dev).requires_grad_(True)
    make_loss(**kw)(again, y.to(dev)).backward()
    assert torch.equal(again.grad, xd.grad)  # no atomics left on this path: bit for bit


def test_mrstft_known_answers(dev):
    """MR-STFT(x, x) = 0 and MR-STFT(c*y, y) = |1 - c| + |ln c| (away from the 1e-8 clamp)."""
    torch.manual_seed(0)
    y = torch.randn(2, 2, 40000).to(dev)
    f = make_loss()
    assert abs(f(y, y).item()) < 1e-6
    for c in (0.5, 2.0, 1.25):
        assert abs(f(c * y, y).item() - (abs(1 - c) + abs(math.log(c)))) < 2e-5
    # gradient of an identical pair is finite (SC is 0/0 there; we return 0 for that term)
    x = y.clone().requires_grad_(True)
    f(x, y).backward()
    assert torch.isfinite(x.grad).all()


def test_mrstft_value_only_call(dev):
    """No gradient asked for (torch.no_grad(), or a prediction that does not require grad): `mst_mrstft_forward_eval` - the same launches
    without the kept spectra - returns the bit-identical loss, and a later differentiable call is unaffected."""
    torch.manual_seed(5)
    x = (0.1 * torch.randn(2, 2, 65536)).to(dev)
    y = (0.1 * torch.randn(2, 2, 65536)).to(dev)
    f = make_loss()
    xg = x.clone().requires_grad_(True)
    l_grad = f(xg, y)
    with torch.no_grad():
        l_eval = f(xg, y)
    l_detached = f(x, y)
    assert not l_eval.requires_grad and not l_detached.requires_grad
    assert torch.equal(l_eval, l_grad.detach()) and torch.equal(l_detached, l_grad.detach())
    l_grad.backward()
    g1 = xg.grad.clone()
    xg.grad = None
    f(xg, y).backward()
    assert torch.equal(g1, xg.grad) and torch.isfinite(g1).all()


def test_mrstft_odd_configuration(dev):
    """auraloss default-like resolutions: hop not dividing n_fft, win_length < n_fft, odd length."""
    from oracle import loss_restated as ol

    res = ((1024, 120, 600), (2048, 240, 1200), (512, 50, 240))
    torch.manual_seed(3)
    n = 30001
    x, y = 0.1 * torch.randn(1, 2, n), 0.1 * torch.randn(1, 2, n)
    xd = x.to(dev).requires_grad_(True)
    loss = make_loss(res)(xd, y.to(dev))
    loss.backward()
    xo = x.double().requires_grad_(True)
    lo = ol.mrstft_loss(xo, y.double(), res)
    lo.backward()
    assert abs(loss.item() - lo.item()) / lo.item() < 1e-5
    assert rel(xd.grad, xo.grad) < 2e-3


def test_mrstft_unsupported(dev):
    from mst.loss import MultiResolutionSTFTLoss

    with pytest.raises(NotImplementedError):
        MultiResolutionSTFTLoss(w_phs=1.0)
    with pytest.raises(ValueError):
        MultiResolutionSTFTLoss(fft_sizes=[1000], hop_sizes=[100], win_lengths=[1000])(
            torch.zeros(1, 2, 4096, device=dev), torch.zeros(1, 2, 4096, device=dev))


def test_peak_normalize(dev):
    from mst.utils import batch_stereo_peak_normalize
    from oracle import console_restated as oc

    torch.manual_seed(1)
    for n in (1000, 4099, 262144):
        x = torch.randn(3, 2, n) * torch.tensor([0.1, 3.0, 1e-3]).view(3, 1, 1)
        x[2] = 0.0  # silent item: clamp path
        xd = x.to(dev).requires_grad_(True)
        y = batch_stereo_peak_normalize(xd)
        ref_in = x.clone().requires_grad_(True)
        ref = oc.batch_stereo_peak_normalize(ref_in)
        assert torch.equal(y.detach().cpu(), ref.detach())  # same single division per element
        g = torch.randn(3, 2, n)
        y.backward(g.to(dev))
        ref.backward(g)
        assert rel(xd.grad[:2], ref_in.grad[:2]) < 1e-5
        assert torch.isfinite(xd.grad).all()


# ------------------------------------------------------------------------------------------------
# AudioFeatureLoss
# ------------------------------------------------------------------------------------------------
AF_WEIGHTS = [0.1, 0.001, 1.0, 1.0, 0.1]  # reference configs/models/unpaired+feat.yaml:55-60


def test_afloss_golden(dev, golden_dir, record):
    """Fixture produced by the REAL reference mst.loss.AudioFeatureLoss (tests/golden/make_golden.py)."""
    import numpy as np
    import os

    from mst.loss import AF_KEYS, AudioFeatureLoss
    from oracle import loss_restated as ol

    g = np.load(os.path.join(golden_dir, "af_loss.npz"))
    x = torch.from_numpy(g["input"]).to(dev).requires_grad_(True)
    y = torch.from_numpy(g["target"]).to(dev)
    ld = AudioFeatureLoss(weights=list(g["weights"]), sample_rate=44100)(x, y)
    assert tuple(ld.keys()) == AF_KEYS
    # three-way: the reference's own fp32 value sits up to 1e-5 from the float64 evaluation (stereo width: a squared difference
    # of two ratios of sums) - HIP must be no further from float64 than twice that, and within 5e-5 of the reference outright
    truth = ol.audio_feature_loss(torch.from_numpy(g["input"]).double(), y.double().cpu(), list(g["weights"]))
    for k in AF_KEYS:
        ref, t64 = float(g["loss." + k]), truth[k].item()
        assert abs(ld[k].item() - ref) <= 5e-5 * abs(ref) + 1e-12, (k, ld[k].item(), ref)
        assert abs(ld[k].item() - t64) <= 2 * abs(ref - t64) + 1e-5 * abs(t64), (k, ld[k].item(), ref, t64)
    sum(v.mean() for v in ld.values()).backward()  # reference mst/system.py:334-336
    gsub = torch.from_numpy(g["grad_input_sub"])
    record(grad_vs_reference=rel(x.grad[..., ::16], gsub),
           **{k.replace("-", "_"): abs(ld[k].item() - float(g["loss." + k])) / abs(float(g["loss." + k])) for k in AF_KEYS})
    assert rel(x.grad[..., ::16], gsub) < 2e-4
    assert abs(x.grad.double().pow(2).sum().sqrt().item() - float(g["grad_input_l2"])) / float(g["grad_input_l2"]) < 2e-4


@pytest.mark.parametrize("bs,n", [(2, 131072), (4, 262144), (1, 16385), (3, 50001)])
def test_afloss_three_way(bs, n, dev, record):
    from mst.loss import AF_KEYS, AudioFeatureLoss
    from oracle import loss_restated as ol

    torch.manual_seed(n + bs)
    x = 0.2 * torch.randn(bs, 2, n)
    x[:, 1] = 0.6 * x[:, 1] + 0.3 * x[:, 0]
    y = 0.3 * torch.randn(bs, 2, n) * torch.tensor([1.0, 0.5]).view(1, 2, 1)
    gw = torch.tensor([1.0, 2.0, 0.5, 1.5, 1.0])
    xd = x.to(dev).requires_grad_(True)
    ld = AudioFeatureLoss(AF_WEIGHTS, 44100)(xd, y.to(dev))
    vals = torch.stack([ld[k] for k in AF_KEYS])
    (vals * gw.to(dev)).sum().backward()
    res = {}
    for dt in (torch.float32, torch.float64):
        xo = x.clone().to(dt).requires_grad_(True)
        lo = ol.audio_feature_loss(xo, y.to(dt), AF_WEIGHTS)
        vo = torch.stack([lo[k] for k in ol.AF_KEYS])
        (vo * gw.to(dt)).sum().backward()
        res[dt] = (vo.detach().double(), xo.grad)
    v64, g64 = res[torch.float64]
    v32, g32 = res[torch.float32]
    err = ((vals.detach().cpu().double() - v64).abs() / v64.abs().clamp_min(1e-30))
    err32 = ((v32 - v64).abs() / v64.abs().clamp_min(1e-30))
    record(loss_rel_err_hip_vs_f64=err.tolist(), loss_rel_err_ref32_vs_f64=err32.tolist(),
           grad=(rel(xd.grad, g32), rel(xd.grad, g64), rel(g32, g64)))
    print(f"\n[af {bs}x2x{n}] loss rel err hip {err.tolist()} ref32 {err32.tolist()}; grad hip-f64 {rel(xd.grad, g64):.2e} ref32-f64 {rel(g32, g64):.2e}")
    assert (err <= 3 * err32 + 2e-5).all()
    assert rel(xd.grad, g64) <= 3 * rel(g32, g64) + 2e-5


def test_afloss_known_answers(dev):
    """L = R => width 0; silent left => imbalance +1; identical input and target => all five losses 0."""
    from mst.loss import AF_KEYS, AudioFeatureLoss

    torch.manual_seed(0)
    n = 40000
    f = AudioFeatureLoss([1.0] * 5, 44100)
    a = torch.randn(2, 2, n).to(dev)
    z = f(a, a)
    assert all(abs(z[k].item()) < 1e-10 for k in AF_KEYS)
    mono = a.clone()
    mono[:, 1] = mono[:, 0]           # width(mono) = 0
    wide = a.clone()
    wide[:, 1] = -wide[:, 0]          # sum channel silent => width = mean(4 L^2) / clamp(0, 1e-8)
    ld = f(mono, wide)
    e = (4 * a[:, 0] ** 2).mean(dim=-1) / 1e-8
    assert abs(ld["mix-stereo_width"].item() - (e ** 2).mean().item()) / (e ** 2).mean().item() < 1e-4
    left_silent = a.clone()
    left_silent[:, 0] = 0.0           # imbalance = +1
    right_silent = a.clone()
    right_silent[:, 1] = 0.0          # imbalance = -1
    assert abs(f(left_silent, right_silent)["mix-stereo_imbalance"].item() - 4.0) < 1e-5
    sq = torch.ones(1, 2, n, device=dev)
    sq[..., ::2] = -1.0               # +-1 square wave: crest factor 0 dB
    half = 0.5 * sq                   # same crest factor
    assert abs(f(sq, half)["mix-crest_factor"].item()) < 1e-8


def test_afloss_input_validation(dev):
    from mst.loss import AudioFeatureLoss

    f = AudioFeatureLoss(AF_WEIGHTS, 44100)
    with pytest.raises(ValueError):
        f(torch.zeros(1, 2, 1000, device=dev), torch.zeros(1, 2, 1000, device=dev))
    with pytest.raises(AssertionError):
        AudioFeatureLoss([1.0] * 6, 44100)  # reference asserts len(weights) == 5 (mst/loss.py:236)
