"""Restatement of the ``dasp_pytorch.functional`` ops the console calls.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  **Parity unpinned**: the real
package (``dasp-pytorch==0.0.1``, ``/root/reference/requirements.txt:15``,
``/root/reference/setup.py:35``) is absent from this image, so what follows
restates its published algorithm and is anchored
on the reference's call sites in ``/root/reference/mst/modules.py``:

* ``gain``            <- modules.py:231, 288, 308
* ``parametric_eq``   <- modules.py:237, 293
* ``compressor``      <- modules.py:246 (look-ahead 2048), 300 (look-ahead 1024)
* ``stereo_panner``   <- modules.py:263
* ``stereo_bus``      <- modules.py:276

Every function works in the dtype of its inputs, so the same code is the
fp32 "reference algorithm" (frequency-sampling IIR through ``torch.fft``) and,
fed float64 tensors, a high-precision version of it.  ``time_domain=True``
variants run the literal recursions (float64 recommended) and serve as the
independent truth the HIP kernels - which are time-domain - are compared with.
"""
from __future__ import annotations

import math

import torch

LOG9 = math.log(9.0)


# --------------------------------------------------------------------------- A.1
def gain(x: torch.Tensor, sample_rate: float, gain_db: torch.Tensor) -> torch.Tensor:
    rows = x.size(0)
    return x * 10 ** (gain_db.view(rows, 1, 1) / 20.0)


# --------------------------------------------------------------------------- A.2
def stereo_panner(x: torch.Tensor, sample_rate: float, pan: torch.Tensor) -> torch.Tensor:
    """(bs, T, n) , (bs, T) -> (bs, 2, T, n) with the constant-power-ish sin/cos law."""
    bs, n_tracks, _ = x.size()
    half_pi = math.pi / 2
    theta = pan.view(bs, n_tracks) * half_pi
    left = torch.sqrt((half_pi - theta) * (2 / math.pi) * torch.cos(theta))
    right = torch.sqrt(theta * (2 / math.pi) * torch.sin(theta))
    lr = torch.stack((left, right), dim=1).unsqueeze(-1)  # (bs, 2, T, 1)
    return x.unsqueeze(1).repeat(1, 2, 1, 1) * lr


def stereo_bus(x: torch.Tensor, sample_rate: float, send_db: torch.Tensor) -> torch.Tensor:
    bs, _, n_tracks, _ = x.size()
    return (x * 10 ** (send_db.view(bs, 1, n_tracks, 1) / 20.0)).sum(dim=2)


# --------------------------------------------------------------------------- A.3
def biquad(gain_db, cutoff_freq, q_factor, sample_rate: float, filter_type: str):
    """RBJ cookbook sections, normalised by a0.  Inputs (rows,) -> b (rows,3), a (rows,3)."""
    gain_db = gain_db.reshape(-1)
    cutoff_freq = cutoff_freq.reshape(-1)
    q_factor = q_factor.reshape(-1)

    A = 10 ** (gain_db / 40.0)
    w0 = 2 * math.pi * (cutoff_freq / sample_rate)
    alpha = torch.sin(w0) / (2 * q_factor)
    cw = torch.cos(w0)
    sA = torch.sqrt(A)

    if filter_type == "low_shelf":
        b0 = A * ((A + 1) - (A - 1) * cw + 2 * sA * alpha)
        b1 = 2 * A * ((A - 1) - (A + 1) * cw)
        b2 = A * ((A + 1) - (A - 1) * cw - 2 * sA * alpha)
        a0 = (A + 1) + (A - 1) * cw + 2 * sA * alpha
        a1 = -2 * ((A - 1) + (A + 1) * cw)
        a2 = (A + 1) + (A - 1) * cw - 2 * sA * alpha
    elif filter_type == "high_shelf":
        b0 = A * ((A + 1) + (A - 1) * cw + 2 * sA * alpha)
        b1 = -2 * A * ((A - 1) + (A + 1) * cw)
        b2 = A * ((A + 1) + (A - 1) * cw - 2 * sA * alpha)
        a0 = (A + 1) - (A - 1) * cw + 2 * sA * alpha
        a1 = 2 * ((A - 1) - (A + 1) * cw)
        a2 = (A + 1) - (A - 1) * cw - 2 * sA * alpha
    elif filter_type == "peaking":
        b0 = 1 + alpha * A
        b1 = -2 * cw
        b2 = 1 - alpha * A
        a0 = 1 + alpha / A
        a1 = -2 * cw
        a2 = 1 - alpha / A
    else:
        raise ValueError(filter_type)

    b = torch.stack((b0, b1, b2), dim=-1) / a0.unsqueeze(-1)
    a = torch.stack((a0, a1, a2), dim=-1) / a0.unsqueeze(-1)
    return b, a


EQ_BANDS = (
    ("low_shelf", "low_shelf"),
    ("band0", "peaking"),
    ("band1", "peaking"),
    ("band2", "peaking"),
    ("band3", "peaking"),
    ("high_shelf", "high_shelf"),
)


def eq_sos(sample_rate: float, **p) -> torch.Tensor:
    """18 parameter tensors (any shape, numel == rows) -> sos (rows, 6, 6) = [b0 b1 b2 a0 a1 a2]."""
    sections = []
    for name, kind in EQ_BANDS:
        b, a = biquad(
            p[f"{name}_gain_db"], p[f"{name}_cutoff_freq"], p[f"{name}_q_factor"], sample_rate, kind
        )
        sections.append(torch.cat((b, a), dim=-1))
    return torch.stack(sections, dim=1)


def fsm_nfft(n: int) -> int:
    return 1 << int(math.ceil(math.log2(2 * n - 1)))


# --------------------------------------------------------------------------- A.4
def sosfilt_via_fsm(sos: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Frequency-sampling cascade: H = prod rfft(b)/rfft(a); y = irfft(rfft(x) H)[:n]."""
    n = x.shape[-1]
    n_fft = fsm_nfft(n)
    H = None
    for s in range(sos.shape[1]):
        B = torch.fft.rfft(sos[:, s, :3], n_fft)
        A = torch.fft.rfft(sos[:, s, 3:], n_fft)
        H = B / A if H is None else H * (B / A)
    X = torch.fft.rfft(x, n_fft)
    y = torch.fft.irfft(X * H.unsqueeze(1), n_fft)
    return y[..., :n]


def lfilter_via_fsm(x: torch.Tensor, b: torch.Tensor, a: torch.Tensor) -> torch.Tensor:
    """x (rows, n), b/a (rows, taps)."""
    n = x.shape[-1]
    n_fft = fsm_nfft(n)
    H = torch.fft.rfft(b, n_fft) / torch.fft.rfft(a, n_fft)
    return torch.fft.irfft(torch.fft.rfft(x, n_fft) * H, n_fft)[..., :n]


def sosfilt_time_domain(sos: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Literal DF2T recursion (no autograd use intended; float64 truth)."""
    import numpy as np
    from scipy.signal import sosfilt

    out = np.empty(x.shape, dtype=np.float64)
    s = sos.detach().double().numpy()
    xn = x.detach().double().numpy()
    for r in range(x.shape[0]):
        out[r] = sosfilt(s[r], xn[r], axis=-1)
    return torch.from_numpy(out).to(x.dtype)


def parametric_eq(x: torch.Tensor, sample_rate: float, time_domain: bool = False, **p) -> torch.Tensor:
    sos = eq_sos(sample_rate, **p)
    if time_domain:
        return sosfilt_time_domain(sos, x)
    return sosfilt_via_fsm(sos, x)


# --------------------------------------------------------------------------- A.5
def compressor_gain_computer(x_db, threshold_db, ratio, knee_db):
    """Soft-knee static curve; returns g_c = x_sc - x_db (<= 0)."""
    x_sc = x_db.clone()
    lo = threshold_db - knee_db / 2
    hi = threshold_db + knee_db / 2
    in_knee = torch.logical_and(x_db >= lo, x_db <= hi)
    knee_val = x_db + ((1 / ratio) - 1) * ((x_db - threshold_db + knee_db / 2) ** 2) / (2 * knee_db)
    x_sc[in_knee] = knee_val[in_knee]
    above = x_db > hi
    lin_val = threshold_db + (x_db - threshold_db) / ratio
    x_sc[above] = lin_val[above]
    return x_sc - x_db


def compressor(
    x: torch.Tensor,
    sample_rate: float,
    threshold_db: torch.Tensor,
    ratio: torch.Tensor,
    attack_ms: torch.Tensor,
    release_ms: torch.Tensor,
    knee_db: torch.Tensor,
    makeup_gain_db: torch.Tensor,
    eps: float = 1e-8,
    lookahead_samples: int = 0,
    time_domain: bool = False,
) -> torch.Tensor:
    rows, chs, n = x.size()
    side = x.sum(dim=1, keepdim=True)
    threshold_db = threshold_db.view(rows, 1, 1)
    ratio = ratio.view(rows, 1, 1)
    attack_ms = attack_ms.view(rows, 1, 1)
    knee_db = knee_db.view(rows, 1, 1)
    makeup_gain_db = makeup_gain_db.view(rows, 1, 1)

    # one time constant for attack and release (release_ms accepted, unused)
    alpha = torch.exp(-LOG9 / (sample_rate * (attack_ms / 1e3)))

    x_db = 20 * torch.log10(side.abs().clamp(eps))
    g_c = compressor_gain_computer(x_db, threshold_db, ratio, knee_db)

    if time_domain:
        import numpy as np
        from scipy.signal import lfilter

        g_np = g_c.detach().double().numpy()[:, 0, :]
        al = alpha.detach().double().numpy().reshape(rows)
        out = np.empty_like(g_np)
        for r in range(rows):
            out[r] = lfilter([1 - al[r]], [1.0, -al[r]], g_np[r])
        g_s = torch.from_numpy(out).to(x.dtype).view(rows, 1, n)
    else:
        b = torch.cat((1 - alpha, torch.zeros_like(alpha)), dim=-1).view(rows, 2)
        a = torch.cat((torch.ones_like(alpha), -alpha), dim=-1).view(rows, 2)
        g_s = lfilter_via_fsm(g_c[:, 0, :], b, a).view(rows, 1, n)

    if lookahead_samples > 0:
        x = torch.roll(x, lookahead_samples, dims=-1)
        x[..., :lookahead_samples] = 0

    return x * 10 ** ((g_s + makeup_gain_db) / 20.0)


# --------------------------------------------------------------------------- A.6
def octave_band_filterbank(num_taps: int, sample_rate: float) -> torch.Tensor:
    """(12, 1, num_taps): 12 Hz low-pass, ten octave band-passes 31.5 Hz .. 16 kHz, 18 kHz high-pass (scipy firwin,
    float64 design rounded to float32, each filter flipped) - dasp_pytorch.signal.octave_band_filterbank."""
    import numpy as np
    import scipy.signal

    bands = [31.5, 63, 125, 250, 500, 1000, 2000, 4000, 8000, 16000]
    filts = []
    filt = scipy.signal.firwin(num_taps, 12, fs=sample_rate)
    filts.append(torch.flip(torch.from_numpy(filt.astype("float32")), dims=[0]))
    for fc in bands:
        f_min = fc / np.sqrt(2)
        f_max = fc * np.sqrt(2)
        f_max = np.clip(f_max, a_min=0, a_max=(sample_rate / 2) * 0.999)
        filt = scipy.signal.firwin(num_taps, [f_min, f_max], fs=sample_rate, pass_zero=False)
        filts.append(torch.flip(torch.from_numpy(filt.astype("float32")), dims=[0]))
    filt = scipy.signal.firwin(num_taps, 18000, fs=sample_rate, pass_zero=False)
    filts.append(torch.flip(torch.from_numpy(filt.astype("float32")), dims=[0]))
    return torch.stack(filts, dim=0).unsqueeze(1)


def noise_shaped_reverberation(
    x: torch.Tensor,
    sample_rate: float,
    band0_gain, band1_gain, band2_gain, band3_gain, band4_gain, band5_gain,
    band6_gain, band7_gain, band8_gain, band9_gain, band10_gain, band11_gain,
    band0_decay, band1_decay, band2_decay, band3_decay, band4_decay, band5_decay,
    band6_decay, band7_decay, band8_decay, band9_decay, band10_decay, band11_decay,
    mix,
    num_samples: int = 65536,
    num_bandpass_taps: int = 1023,
    noise: torch.Tensor = None,
) -> torch.Tensor:
    """modules.py:277-283.  Band-passed white noise (12 bands), per-band exponential decay exp(-(10 d + 1) t), t in [0,1],
    per-band gain, mean over bands = a stereo impulse response of `num_samples` taps; causal convolution with the input;
    wet / dry mix.  `noise` (bs*2, 12, num_samples + num_bandpass_taps - 1): the op draws it with torch.randn when absent -
    an explicit argument makes the GPU kernels testable against this restatement."""
    assert num_bandpass_taps % 2 == 1
    bs, chs, seq_len = x.size()
    assert chs == 2
    gains = torch.stack([band0_gain, band1_gain, band2_gain, band3_gain, band4_gain, band5_gain, band6_gain, band7_gain,
                         band8_gain, band9_gain, band10_gain, band11_gain], dim=1).view(bs, 12)
    decays = torch.stack([band0_decay, band1_decay, band2_decay, band3_decay, band4_decay, band5_decay, band6_decay,
                          band7_decay, band8_decay, band9_decay, band10_decay, band11_decay], dim=1).view(bs, 12)
    filters = octave_band_filterbank(num_bandpass_taps, sample_rate).type_as(x)
    num_bands = filters.shape[0]
    pad_size = num_bandpass_taps - 1
    wn = torch.randn(bs * 2, num_bands, num_samples + pad_size).type_as(x) if noise is None else noise.to(x.dtype)
    wn_filt = torch.nn.functional.conv1d(wn, filters, groups=num_bands)  # (bs*2, 12, num_samples)
    wn_filt = wn_filt.view(bs, 2, num_bands, num_samples)
    t = torch.linspace(0, 1, steps=num_samples).type_as(x)
    rates = (decays * 10.0) + 1.0
    env = torch.exp(-rates.view(bs, 1, num_bands, 1) * t.view(1, 1, 1, -1))
    wn_filt = wn_filt * env * gains.view(bs, 1, num_bands, 1)
    ir = wn_filt.mean(2)  # (bs, 2, num_samples)
    # y[n] = sum_j ir[j] x[n - j] (conv1d of the left-padded input with the flipped response), evaluated with FFTs
    n_fft = 1 << int(math.ceil(math.log2(seq_len + num_samples - 1)))
    y = torch.fft.irfft(torch.fft.rfft(x, n_fft) * torch.fft.rfft(ir, n_fft), n_fft)[..., :seq_len]
    mix = mix.view(bs, 1, 1)
    return (1 - mix) * x + mix * y
