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
