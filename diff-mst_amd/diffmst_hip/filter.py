"""``mst.filter`` - Bark filterbank table (reference mst/filter.py:107-161).

A constant (n_freqs, n_barks) matrix per (fft size, sample rate): it is host logic, built once with
plain tensor arithmetic and cached by the loss.  The arithmetic (fp32 linspace, the Traunmuller
Hz<->Bark maps and their *either/or* low / high corrections, triangular filters) follows the
reference so that the table is bit-identical (pinned by tests/golden/bark_fb.npz).
"""
import math
import warnings

import torch


def _hz_to_bark(freqs: float, bark_scale: str = "traunmuller") -> float:
    if bark_scale not in ("schroeder", "traunmuller", "wang"):
        raise ValueError('bark_scale should be one of "schroeder", "traunmuller" or "wang".')
    if bark_scale == "wang":
        return 6.0 * math.asinh(freqs / 600.0)
    if bark_scale == "schroeder":
        return 7.0 * math.asinh(freqs / 650.0)
    barks = ((26.81 * freqs) / (1960.0 + freqs)) - 0.53
    if barks < 2:
        barks += 0.15 * (2 - barks)
    elif barks > 20.1:
        barks += 0.22 * (barks - 20.1)
    return barks


def _bark_to_hz(barks: torch.Tensor, bark_scale: str = "traunmuller") -> torch.Tensor:
    if bark_scale not in ("schroeder", "traunmuller", "wang"):
        raise ValueError('bark_scale should be one of "traunmuller", "schroeder" or "wang".')
    if bark_scale == "wang":
        return 600.0 * torch.sinh(barks / 6.0)
    if bark_scale == "schroeder":
        return 650.0 * torch.sinh(barks / 7.0)
    # the reference corrects EITHER the low end OR the high end, in place (mst/filter.py:89-94)
    low = barks < 2
    if bool(low.any()):
        barks[low] = (barks[low] - 0.3) / 0.85
    else:
        high = barks > 20.1
        if bool(high.any()):
            barks[high] = (barks[high] + 4.422) / 1.22
    return 1960 * ((barks + 0.53) / (26.28 - barks))


def _create_triangular_filterbank(all_freqs: torch.Tensor, f_pts: torch.Tensor) -> torch.Tensor:
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def barkscale_fbanks(n_freqs: int, f_min: float, f_max: float, n_barks: int, sample_rate: int,
                     bark_scale: str = "traunmuller") -> torch.Tensor:
    """Triangular Bark filterbank of shape ``(n_freqs, n_barks)``."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_pts = torch.linspace(_hz_to_bark(f_min, bark_scale), _hz_to_bark(f_max, bark_scale), n_barks + 2)
    fb = _create_triangular_filterbank(all_freqs, _bark_to_hz(m_pts, bark_scale))
    if (fb.max(dim=0).values == 0.0).any():
        warnings.warn(
            "At least one bark filterbank has all zero values. "
            f"The value for `n_barks` ({n_barks}) may be set too high. "
            f"Or, the value for `n_freqs` ({n_freqs}) may be set too low."
        )
    return fb


def octave_band_filterbank(num_taps: int, sample_rate: float) -> torch.Tensor:
    """(12, num_taps) FIR filterbank of dasp-pytorch's ``noise_shaped_reverberation`` (``dasp_pytorch.signal``, restated
    from the published algorithm): a 12 Hz low-pass, ten octave band-passes centred on 31.5 Hz .. 16 kHz
    (fc / sqrt 2 .. fc sqrt 2, upper edge clipped below Nyquist), an 18 kHz high-pass - ``scipy.signal.firwin`` designs in
    float64, rounded to float32.  A constant table per (taps, sample rate), uploaded once by the console."""
    import numpy as np
    from scipy.signal import firwin  # the same designer the reference's dependency calls

    bands = [31.5, 63, 125, 250, 500, 1000, 2000, 4000, 8000, 16000]
    filts = [firwin(num_taps, 12, fs=sample_rate)]
    for fc in bands:
        f_min, f_max = fc / np.sqrt(2), fc * np.sqrt(2)
        f_max = float(np.clip(f_max, a_min=0, a_max=(sample_rate / 2) * 0.999))
        filts.append(firwin(num_taps, [f_min, f_max], fs=sample_rate, pass_zero=False))
    filts.append(firwin(num_taps, 18000, fs=sample_rate, pass_zero=False))
    # dasp flips each (symmetric) filter and applies it with conv1d (a correlation): the flip is kept for fidelity
    return torch.stack([torch.flip(torch.from_numpy(f.astype("float32")), dims=[0]) for f in filts], dim=0)
