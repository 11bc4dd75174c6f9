"""Restatement of the two losses on the hot path.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

* ``MultiResolutionSTFTLoss`` - ``auraloss==0.4.0`` (``requirements.txt:7``), absent
  from this image => **parity unpinned**; restates SURVEY.md Appendix A.7 and is
  anchored on the reference's configuration
  (``/root/reference/configs/models/naive.yaml:54-68``, ``mst/system.py:61-69``).
  ``sc_per_example`` selects between the per-example spectral-convergence ratio
  (recalled 0.4.0 behaviour, default) and the batch-global ratio of older releases.
* ``audio_feature_loss`` and the five feature transforms follow
  ``/root/reference/mst/loss.py:62-260``; the Bark filterbank follows
  ``/root/reference/mst/filter.py:8-161`` including its quirks.
  These are mst-owned and **pinned** by ``tests/golden/make_golden.py``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

MRSTFT_DEFAULT = ((512, 256, 512), (2048, 1024, 2048), (8192, 4096, 8192))  # naive.yaml:57-68


def stft_mag(x2d: torch.Tensor, n_fft: int, hop: int, win: int, eps: float = 1e-8) -> torch.Tensor:
    window = torch.hann_window(win, dtype=x2d.dtype, device=x2d.device)
    X = torch.stft(x2d, n_fft, hop, win, window, return_complex=True)
    return torch.sqrt(torch.clamp(X.real**2 + X.imag**2, min=eps))


def stft_loss(x, y, n_fft, hop, win, w_sc=1.0, w_log_mag=1.0, w_lin_mag=0.0, sc_per_example=True):
    xm = stft_mag(x.reshape(-1, x.size(-1)), n_fft, hop, win)
    ym = stft_mag(y.reshape(-1, y.size(-1)), n_fft, hop, win)
    loss = 0.0
    if w_sc:
        if sc_per_example:
            sc = (torch.norm(ym - xm, p="fro", dim=[-1, -2]) / torch.norm(ym, p="fro", dim=[-1, -2])).mean()
        else:
            sc = torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
        loss = loss + w_sc * sc
    if w_log_mag:
        loss = loss + w_log_mag * F.l1_loss(torch.log(xm), torch.log(ym))
    if w_lin_mag:
        loss = loss + w_lin_mag * F.l1_loss(xm, ym)
    return loss


def mrstft_loss(x, y, resolutions=MRSTFT_DEFAULT, **kw):
    total = 0.0
    for n_fft, hop, win in resolutions:
        total = total + stft_loss(x, y, n_fft, hop, win, **kw)
    return total / len(resolutions)


# ---------------------------------------------------------------- Bark filterbank (filter.py)
def _hz_to_bark_traunmuller(f: float) -> float:
    z = 26.81 * f / (1960.0 + f) - 0.53
    if z < 2:
        z += 0.15 * (2 - z)
    elif z > 20.1:
        z += 0.22 * (z - 20.1)
    return z


def bark_filterbank(n_freqs: int, f_min: float, f_max: float, n_barks: int, sample_rate: int) -> torch.Tensor:
    """(n_freqs, n_barks) triangular filters; reproduces filter.py:107-161 for the traunmuller scale."""
    freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    z = torch.linspace(_hz_to_bark_traunmuller(f_min), _hz_to_bark_traunmuller(f_max), n_barks + 2)
    # filter.py:89-94: EITHER the low OR the high correction is applied, never both
    if bool((z < 2).any()):
        m = z < 2
        z[m] = (z[m] - 0.3) / 0.85
    elif bool((z > 20.1).any()):
        m = z > 20.1
        z[m] = (z[m] + 4.422) / 1.22
    pts = 1960 * ((z + 0.53) / (26.28 - z))
    gaps = pts[1:] - pts[:-1]
    slopes = pts.unsqueeze(0) - freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / gaps[:-1]
    up = slopes[:, 2:] / gaps[1:]
    return torch.clamp(torch.min(down, up), min=0.0)


# ---------------------------------------------------------------- AudioFeatureLoss (loss.py)
def feat_rms(x):
    return torch.sqrt(torch.mean(x**2, dim=-1).clamp(min=1e-8))


def feat_crest_factor(x):
    peak = x.abs().max(dim=-1)[0]
    return 20 * torch.log10((peak / feat_rms(x).clamp(min=1e-8)).clamp(min=1e-8))


def feat_stereo_width(x):
    s = x[:, 0, :] + x[:, 1, :]
    d = x[:, 0, :] - x[:, 1, :]
    return torch.mean(d**2, dim=-1) / torch.mean(s**2, dim=-1).clamp(min=1e-8)


def feat_stereo_imbalance(x):
    el = torch.mean(x[:, 0, :] ** 2, dim=-1)
    er = torch.mean(x[:, 1, :] ** 2, dim=-1)
    return (er - el) / (er + el).clamp(min=1e-8)


def feat_barkspectrum(x, sample_rate=44100, fft_size=32768, n_bands=24, f_min=20.0, f_max=20000.0):
    fb = bark_filterbank(fft_size // 2 + 1, f_min, f_max, n_bands, sample_rate).to(dtype=x.dtype, device=x.device).t().unsqueeze(0)
    window = torch.hann_window(fft_size).to(dtype=x.dtype, device=x.device)  # made on the host like the reference's, then moved: the checker may run on any torch device
    outs = []
    for sig in (x[:, 0, :] + x[:, 1, :], x[:, 0, :] - x[:, 1, :]):
        X = torch.stft(sig, n_fft=fft_size, hop_length=fft_size // 4, window=window, return_complex=True)
        mean_mag = X.abs().mean(dim=-1, keepdim=True)
        outs.append(torch.log(torch.matmul(fb, mean_mag) + 1e-8))
    return torch.cat(outs, dim=-1)


AF_KEYS = ("mix-rms", "mix-crest_factor", "mix-stereo_width", "mix-stereo_imbalance", "mix-barkspectrum")


def audio_feature_loss(inp, tgt, weights, sample_rate=44100) -> dict:
    feats = (
        feat_rms,
        feat_crest_factor,
        feat_stereo_width,
        feat_stereo_imbalance,
        lambda t: feat_barkspectrum(t, sample_rate=sample_rate),
    )
    assert len(weights) == len(feats)
    return {k: w * F.mse_loss(f(inp), f(tgt)) * 1.0 for k, w, f in zip(AF_KEYS, weights, feats)}
