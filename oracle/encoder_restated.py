"""Restatement of the spectrogram encoder.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Plain PyTorch (CPU or any device torch runs on), functional in the
reference's ``state_dict``: ``cnn14(x, sd, ...)`` follows ``/root/reference/mst/panns.py:126-209`` (ConvBlock :27-85) and
``spectrogram_encoder(wave, sd, ...)`` follows ``/root/reference/mst/modules.py:772-806``.  Both are mst-owned, pure-torch
code of the reference: **pinned** - ``tests/golden/make_golden.py encoder`` imports the real classes, loads the same
``state_dict`` and asserts equal outputs and gradients before it writes the fixture.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

POOLS = ((2, 2), (4, 4), (4, 2), (4, 2), (4, 2), (2, 2))  # panns.py:186-197, over (bins, frames)


class _RoundBf16(torch.autograd.Function):
    """Value rounded to bfloat16 (nearest even) in the forward AND the cotangent rounded the same way in the backward: what a
    tensor that is STORED in bf16 between two kernels goes through in both directions."""

    @staticmethod
    def forward(ctx, t):
        return t.to(torch.bfloat16).to(t.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def cnn14(x: torch.Tensor, sd: dict, training: bool = True, prefix: str = "", momentum: float = 0.1, eps: float = 1e-5,
          emulate_bf16: bool = False) -> torch.Tensor:
    """x (bs, 1, bins, frames) -> (bs, num_classes).  ``sd`` maps the reference's parameter names to tensors (leaves that
    require grad get their gradients); running statistics in ``sd`` are updated in place when ``training``.

    ``emulate_bf16`` (not a reference mode - the checker of the build's bf16 setting): the same network with a bf16 rounding
    at every point where the MI355X kernels STORE a tensor in bf16 - convolution weights (except the first layer's), every
    convolution output, every ReLU / pooling output, and the cotangents of those tensors on the way back - while all
    arithmetic between two stores stays in the working precision (use float64); BatchNorm statistics come from the unrounded
    convolution output, as in the kernels' epilogue."""
    q = _RoundBf16.apply if emulate_bf16 else (lambda t: t)
    for i, pool in enumerate(POOLS, start=1):
        for k in (1, 2):
            p = f"{prefix}conv_block{i}."
            w = sd[p + f"conv{k}.weight"]
            raw = F.conv2d(x, w if (i == 1 and k == 1) else q(w), None, stride=1, padding=1)
            if emulate_bf16:
                ga, be = sd[p + f"bn{k}.weight"].view(1, -1, 1, 1), sd[p + f"bn{k}.bias"].view(1, -1, 1, 1)
                if training:
                    mean, var = raw.mean(dim=(0, 2, 3), keepdim=True), raw.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
                else:
                    mean, var = sd[p + f"bn{k}.running_mean"].view(1, -1, 1, 1), sd[p + f"bn{k}.running_var"].view(1, -1, 1, 1)
                x = (q(raw) - mean) * torch.rsqrt(var + eps) * ga + be
            else:
                x = F.batch_norm(raw, sd[p + f"bn{k}.running_mean"], sd[p + f"bn{k}.running_var"], sd[p + f"bn{k}.weight"],
                                 sd[p + f"bn{k}.bias"], training, momentum, eps)
            x = F.relu(x)
            if k == 1:
                x = q(x)
        x = q(F.avg_pool2d(x, pool))
    x = torch.mean(x, dim=2)           # over bins
    x = torch.max(x, dim=2)[0] + torch.mean(x, dim=2)  # over frames
    return F.linear(x, sd[prefix + "fc.weight"], sd[prefix + "fc.bias"])


def spectrogram(wave2d: torch.Tensor, n_fft: int = 2048, hop_length: int = 512) -> torch.Tensor:
    """(rows, seq_len) -> (rows, bins, frames) = (|STFT| + 1e-8)^0.3 (modules.py:789-800)."""
    window = torch.hann_window(n_fft, dtype=wave2d.dtype, device=wave2d.device)
    X = torch.stft(wave2d, n_fft=n_fft, hop_length=hop_length, window=window, return_complex=True)
    return torch.pow(X.abs() + 1e-8, 0.3)


def spectrogram_encoder(wave: torch.Tensor, sd: dict, training: bool = True, n_fft: int = 2048, hop_length: int = 512,
                        emulate_bf16: bool = False) -> torch.Tensor:
    """wave (bs, chs, seq_len) -> (bs, embed_dim); ``sd`` holds the encoder's ``model.*`` entries."""
    bs, chs, n = wave.shape
    X = spectrogram(wave.reshape(-1, n), n_fft, hop_length).view(bs, chs, n_fft // 2 + 1, -1)
    return cnn14(X, sd, training, prefix="model.", emulate_bf16=emulate_bf16)
