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


@pytest.mark.parametrize("bs,n,kw", [
    (2, 65536, {}),
    (8, 262144, {}),                                                   # BASELINE cfg #2 loss shape
    (2, 131072, dict(w_sc=0.0, w_log_mag=1.0, w_lin_mag=1.0)),          # evaluation instance mst/system.py:61-69
    (1, 50000, dict(sc_per_example=False)),                            # pre-0.4.0 global spectral convergence
])
def test_mrstft_three_way(bs, n, kw, dev):
    from oracle import loss_restated as ol

    torch.manual_seed(bs * 7 + n)
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
    print(f"\n[mrstft {bs}x2x{n} {kw}] loss hip {loss.item():.7f} ref32 {l32:.7f} f64 {l64:.7f}; grad hip-ref32 {h32:.2e} hip-f64 {h64:.2e} ref32-f64 {r:.2e}")
    # d log|X| / dX ~ 1/|X| : among 1e6..1e7 bins a few are nearly zero and dominate the fp32 error of
    # BOTH fp32 implementations (ref32 sits 5e-4..1e-2 from float64); HIP must not be worse than that
    assert h64 <= 6 * r + 5e-4 and h32 <= 2 * (h64 + r)


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
