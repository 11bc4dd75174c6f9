"""Known-answer tests of the oracle's restated third-party arithmetic.

The dasp-pytorch / auraloss sources are not available (parity unpinned), so the restatement is
anchored on analytic facts that any correct implementation must satisfy."""
import math

import numpy as np
import torch

from oracle import console_restated as oc
from oracle import dasp_restated as od
from oracle import loss_restated as ol

SR = 44100


def eq_params(rows, gain_db=0.0, dtype=torch.float64):
    ranges = oc.param_ranges(SR)["parametric_eq"]
    p = {}
    for name, (lo, hi) in ranges.items():
        v = gain_db if name.endswith("gain_db") else 0.5 * (lo + hi)
        p[name] = torch.full((rows,), float(v), dtype=dtype)
    return p


def test_gain_20db_is_times_ten():
    x = torch.randn(2, 1, 64)
    assert torch.allclose(od.gain(x, SR, torch.tensor([20.0, -20.0])), x * torch.tensor([10.0, 0.1]).view(2, 1, 1), rtol=1e-6)


def test_panner_triple():
    """pan 0 / 0.5 / 1 -> (1,0) / (0.594604, 0.594604) / (0,1)  (stimulus of reference tests/test_panner.py:8)."""
    y = od.stereo_panner(torch.ones(1, 4, 1), SR, torch.tensor([[0.0, 0.5, 1.0, 0.0]]))
    want = torch.tensor([[1.0, 0.594604, 0.0, 1.0], [0.0, 0.594604, 1.0, 0.0]])
    assert torch.allclose(y[0, :, :, 0], want, atol=2e-6)


def test_eq_zero_gain_is_identity():
    x = torch.randn(3, 1, 4096, dtype=torch.float64)
    y = od.parametric_eq(x, SR, **eq_params(3, 0.0))
    assert torch.allclose(y, x, atol=1e-12)
    sos = od.eq_sos(SR, **eq_params(3, 0.0))
    assert torch.allclose(sos[..., :3], sos[..., 3:], atol=1e-15)


def test_biquad_magnitudes():
    """low-shelf DC gain = high-shelf Nyquist gain = peaking gain at f0 = 10^(g/20)."""
    g = torch.tensor([7.0], dtype=torch.float64)
    f = torch.tensor([1000.0], dtype=torch.float64)
    q = torch.tensor([0.9], dtype=torch.float64)

    def mag(b, a, w):
        z = np.exp(-1j * w)
        b, a = b[0].numpy(), a[0].numpy()
        return abs((b[0] + b[1] * z + b[2] * z * z) / (a[0] + a[1] * z + a[2] * z * z))

    lin = 10 ** (7.0 / 20)
    b, a = od.biquad(g, f, q, SR, "low_shelf")
    assert abs(mag(b, a, 0.0) - lin) < 1e-9 and abs(mag(b, a, math.pi) - 1.0) < 1e-9
    b, a = od.biquad(g, f, q, SR, "high_shelf")
    assert abs(mag(b, a, math.pi) - lin) < 1e-9 and abs(mag(b, a, 0.0) - 1.0) < 1e-9
    b, a = od.biquad(g, f, q, SR, "peaking")
    assert abs(mag(b, a, 2 * math.pi * 1000.0 / SR) - lin) < 1e-9


def test_frequency_sampling_equals_time_domain_when_the_tail_fits():
    """The frequency-sampling cascade is the true IIR up to the wrapped tail; with a short impulse response
    and float64 both agree with scipy.signal.sosfilt to round-off."""
    torch.manual_seed(0)
    x = torch.randn(2, 1, 8192, dtype=torch.float64)
    p = eq_params(2, 6.0)
    a = od.parametric_eq(x, SR, **p)
    b = od.parametric_eq(x, SR, time_domain=True, **p)
    assert (a - b).abs().max() / b.abs().max() < 1e-10


def test_frequency_sampling_wraps_long_tails_on_short_clips():
    """Documents why short fixtures use short-impulse-response parameters: a 250 ms attack on a 0.19 s clip
    makes the (circular) frequency-sampling smoother differ from the recursion by percents."""
    torch.manual_seed(0)
    x = 0.5 * torch.randn(1, 1, 8192, dtype=torch.float64)
    kw = dict(threshold_db=torch.tensor([-30.0]).double(), ratio=torch.tensor([4.0]).double(),
              release_ms=torch.tensor([50.0]).double(), knee_db=torch.tensor([6.0]).double(),
              makeup_gain_db=torch.tensor([0.0]).double())
    slow = torch.tensor([250.0]).double()
    fast = torch.tensor([5.0]).double()
    d_slow = od.compressor(x, SR, attack_ms=slow, **kw) - od.compressor(x, SR, attack_ms=slow, time_domain=True, **kw)
    d_fast = od.compressor(x, SR, attack_ms=fast, **kw) - od.compressor(x, SR, attack_ms=fast, time_domain=True, **kw)
    assert d_slow.abs().max() > 1e-3 and d_fast.abs().max() < 1e-9


def test_compressor_ratio_one_is_delay_and_makeup():
    torch.manual_seed(1)
    x = torch.randn(2, 2, 6000, dtype=torch.float64)
    L = 1024
    y = od.compressor(x, SR, threshold_db=torch.tensor([-20.0, -10.0]).double(), ratio=torch.ones(2).double(),
                      attack_ms=torch.tensor([10.0, 50.0]).double(), release_ms=torch.tensor([10.0, 10.0]).double(),
                      knee_db=torch.tensor([6.0, 3.0]).double(), makeup_gain_db=torch.tensor([6.0, 0.0]).double(),
                      lookahead_samples=L, time_domain=True)
    want = torch.zeros_like(x)
    want[..., L:] = x[..., :-L]
    want = want * (10 ** (torch.tensor([6.0, 0.0]).double() / 20)).view(2, 1, 1)
    assert torch.allclose(y, want, atol=1e-12)
    assert float(y[..., :L].abs().max()) == 0.0


def test_one_pole_step_response():
    """g_s[n] = g (1 - a^(n+1)); reaches 8/9 of the step after `attack` (stimulus of reference tests/test_comp.py:18-36)."""
    n, attack_ms = 20000, 100.0
    x = torch.ones(1, 1, n, dtype=torch.float64)  # 0 dBFS step, threshold -12, ratio 4, knee 6 -> above the knee
    thr, ratio = -12.0, 4.0
    y = od.compressor(x, SR, threshold_db=torch.tensor([thr]).double(), ratio=torch.tensor([ratio]).double(),
                      attack_ms=torch.tensor([attack_ms]).double(), release_ms=torch.tensor([0.0]).double(),
                      knee_db=torch.tensor([6.0]).double(), makeup_gain_db=torch.tensor([0.0]).double(), time_domain=True)
    g_db = 20 * torch.log10(y[0, 0])
    g_static = (1 / ratio - 1) * (0.0 - thr)
    alpha = math.exp(-math.log(9.0) / (SR * attack_ms / 1e3))
    k = torch.arange(n, dtype=torch.float64)
    assert torch.allclose(g_db, g_static * (1 - alpha ** (k + 1)), atol=1e-9)
    n_att = int(SR * attack_ms / 1e3)
    assert abs(g_db[n_att - 1].item() / g_static - 8 / 9) < 1e-3


def test_soft_knee_is_continuous():
    xdb = torch.linspace(-40, 0, 4001, dtype=torch.float64).view(1, 1, -1)
    g = od.compressor_gain_computer(xdb, torch.tensor(-20.0).double(), torch.tensor(4.0).double(), torch.tensor(6.0).double())
    assert float(g.diff().abs().max()) < 0.01 and float(g[0, 0, 0]) == 0.0
    assert abs(float(g[0, 0, -1]) - (1 / 4 - 1) * 20.0) < 1e-12


def test_mrstft_known_answers():
    torch.manual_seed(0)
    y = torch.randn(2, 2, 20000, dtype=torch.float64)
    assert abs(ol.mrstft_loss(y, y).item()) < 1e-12
    for c in (0.5, 2.0):
        assert abs(ol.mrstft_loss(c * y, y).item() - (abs(1 - c) + abs(math.log(c)))) < 1e-9


def test_audio_feature_known_answers():
    torch.manual_seed(0)
    a = torch.randn(2, 2, 4096, dtype=torch.float64)
    mono = a.clone()
    mono[:, 1] = mono[:, 0]
    assert float(ol.feat_stereo_width(mono).abs().max()) == 0.0
    left_silent = a.clone()
    left_silent[:, 0] = 0
    assert torch.allclose(ol.feat_stereo_imbalance(left_silent), torch.ones(2, dtype=torch.float64))
    sq = torch.ones(1, 2, 4096, dtype=torch.float64)
    sq[..., ::2] = -1
    assert float(ol.feat_crest_factor(sq).abs().max()) < 1e-12
    assert torch.allclose(ol.feat_rms(3 * sq), torch.full((1, 2), 3.0, dtype=torch.float64))


def test_peak_normalize():
    x = torch.randn(3, 2, 100)
    y = oc.batch_stereo_peak_normalize(x)
    assert torch.allclose(y.abs().amax(dim=(1, 2)), torch.ones(3))
    assert float(oc.batch_stereo_peak_normalize(torch.zeros(1, 2, 8)).abs().max()) == 0.0


def test_out_of_range_raises_value_error():
    tp, fp, mp = torch.rand(1, 2, 27), torch.rand(1, 25), torch.rand(1, 26)
    tp[0, 0, 25] = -0.01
    try:
        oc.console_forward(torch.zeros(1, 2, 64), tp, fp, mp)
        raise AssertionError("expected ValueError")
    except ValueError as e:
        assert str(e) == "Parameter pan of effect stereo_panner is out of range."
