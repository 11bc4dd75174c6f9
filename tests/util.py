"""Shared helpers of the test-suite."""
import torch

FULL = dict(use_track_input_fader=True, use_track_eq=True, use_track_compressor=True, use_track_panner=True,
            use_fx_bus=False, use_master_bus=True, use_output_fader=True)


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-300)).item()


def short_ir(tp, mp):
    """Parameters whose impulse responses are far shorter than a short test clip.

    The reference's frequency-sampling filters are circular with period n_fft = 2^ceil(log2(2n-1));
    for clips much shorter than the production 131072..262144 samples a 250 ms attack or a 20 Hz
    shelf wraps around by percents (float64 time-domain vs float64 frequency-sampling differ by 4e-2
    at n = 16384).  The HIP console implements the true (linear) recursion, so short-clip tests that
    compare with the frequency-sampling oracle restrict the attack to <= 29.5 ms and the two
    low-frequency corners to >= ~300 Hz.  Same transformation as tests/golden/make_golden.py.
    """
    tp, mp = tp.clone(), mp.clone()
    tp[..., 21] *= 0.1
    mp[..., 20] *= 0.1
    tp[..., 2] = 0.15 + 0.85 * tp[..., 2]
    tp[..., 5] = 0.15 + 0.85 * tp[..., 5]
    mp[..., 1] = 0.15 + 0.85 * mp[..., 1]
    mp[..., 4] = 0.15 + 0.85 * mp[..., 4]
    return tp, mp


class StubModel(torch.nn.Module):
    """Stand-in for the reference's parameter-estimation model (mst/modules.py:17-68) with its call signature:
    ``model(tracks (bs,T,n), ref_mix (bs,2,n), track_padding_mask=...)`` -> three sigmoid tensors
    ``(bs,T,27), (bs,25), (bs,26)``.  Three tiny linear heads on crude level statistics; weights are drawn from a
    seeded CPU generator so the fixture generator (CPU, real reference) and the GPU test build the same model."""

    def __init__(self, seed: int = 0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        mk = lambda o: torch.nn.Parameter(0.8 * torch.randn(o, 4, generator=g))
        self.w_track, self.w_fx, self.w_master = mk(27), mk(25), mk(26)

    def forward(self, tracks, ref_mix, track_padding_mask=None):
        bs, T, _ = tracks.shape
        ref_rms = ref_mix.pow(2).mean(dim=(-1, -2)).sqrt()  # (bs,)
        ft = torch.stack((10 * tracks.pow(2).mean(-1).sqrt(), 10 * tracks.abs().mean(-1), ref_rms.view(bs, 1).expand(bs, T),
                          torch.ones(bs, T, device=tracks.device)), dim=-1)  # (bs,T,4)
        fm = ft.mean(dim=1)  # (bs,4)
        # squashed away from 0/1: the estimated parameters stay strictly inside the console's ranges
        sq = lambda z: 0.02 + 0.96 * torch.sigmoid(z)
        return sq(ft @ self.w_track.t()), sq(fm @ self.w_fx.t()), sq(fm @ self.w_master.t())


def simple_lufs(x):
    """Stand-in loudness meter for the inference-driver tests: ``x`` is the ``(n, channels)`` numpy array the reference hands
    to ``pyloudnorm.Meter.integrated_loudness`` (mst/utils.py:93-95); un-gated, un-weighted mean-square level with the
    BS.1770 offset.  pyloudnorm is absent from the image; the fixture generator installs THIS function behind the stubbed
    ``pyloudnorm.Meter`` of the real ``run_diffmst`` and the GPU test injects it as ``loudness_fn``."""
    import numpy as np

    return float(-0.691 + 10.0 * np.log10(np.mean(np.asarray(x, dtype=np.float64) ** 2) + 1e-12))


def seeded_controller(cls, seed_init, seed_pert, **kw):
    """tests/golden/make_golden.py controller_setup, call for call (same torch build: same CPU generator stream)."""
    torch.manual_seed(seed_init)
    ctrl = cls(512, 27, 25, 26, num_layers=3, nhead=8, **kw)
    torch.manual_seed(seed_pert)
    with torch.no_grad():
        for name, p in sorted(ctrl.named_parameters()):
            if "norm" in name or name.endswith("bias"):
                p.add_(0.1 * torch.randn_like(p))
    return ctrl
