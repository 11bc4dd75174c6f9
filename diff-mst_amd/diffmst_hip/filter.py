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
