"""``diffmst_hip.loss`` (alias ``mst.loss``) - the losses on the hot path, MI355X-native.

* ``MultiResolutionSTFTLoss`` - drop-in for ``auraloss.freq.MultiResolutionSTFTLoss`` as the reference
  configures it (configs/models/naive.yaml:54-68; evaluation instance mst/system.py:61-69): same
  constructor keywords for the supported subset, called as ``loss(pred, target)`` -> scalar tensor.
  The windowed real-FFT spectrograms, magnitude / log / norm reductions and their adjoints run in
  ``diff-mst_amd/csrc/mst_stft.hip``; no spectrogram is ever materialised.
* ``AudioFeatureLoss`` - reference mst/loss.py:198-260 (see below).
"""
from __future__ import annotations

import ctypes
from typing import List

import torch
from torch.autograd.function import once_differentiable

from . import _cabi, _hip

_TABLE_CACHE = {}


def _mrstft_desc(rows, n, resolutions, w_sc, w_log_mag, w_lin_mag, sc_per_example, eps):
    d = _cabi.MrstftDesc()
    d.rows, d.n_samples, d.n_res = int(rows), int(n), len(resolutions)
    for i, (nf, hop, win) in enumerate(resolutions):
        d.fft_size[i], d.hop_size[i], d.win_length[i] = int(nf), int(hop), int(win)
    d.w_sc, d.w_log_mag, d.w_lin_mag = float(w_sc), float(w_log_mag), float(w_lin_mag)
    d.sc_per_example, d.eps = int(bool(sc_per_example)), float(eps)
    return d


def _tables(desc, resolutions, device):
    """Twiddle + window tables: depend only on (fft_size, win_length); built once per device."""
    key = (str(device), tuple((r[0], r[2]) for r in resolutions))
    t = _TABLE_CACHE.get(key)
    if t is None:
        lib = _hip.lib()
        nbytes = lib.mst_mrstft_tables_bytes(ctypes.byref(desc))
        if nbytes == 0:
            raise ValueError("unsupported STFT configuration (fft sizes must be powers of two in 128..8192, "
                             "win_length <= fft_size, n_samples > fft_size/2)")
        t = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            _hip.check(lib.mst_mrstft_init_tables(ctypes.byref(desc), _cabi.ptr(t), _hip.current_stream_ptr(device)),
                       "mst_mrstft_init_tables")
        _TABLE_CACHE[key] = t
    return t


class _MrstftFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, cfg, want_grad=True):
        _hip.require_cuda(pred, target)
        if ctx.needs_input_grad[1]:
            # auraloss differentiates w.r.t. both arguments; no in-repo caller of the reference asks for the target's gradient
            # (mst/system.py:331-338: the target is the detached reference mix) and the adjoint kernels do not form it
            raise NotImplementedError("MultiResolutionSTFTLoss (MI355X build): the target's gradient is not implemented; detach the target")
        # the backward is wanted when the prediction requires grad AND grad mode is on at the call (forward() itself runs with grad
        # mode off, so the module hands the caller's mode over): under torch.no_grad() the value-only forward keeps no spectra
        want_grad = bool(want_grad) and ctx.needs_input_grad[0]
        ctx.saved = False
        lib = _hip.lib()
        n = pred.shape[-1]
        x = pred.float().reshape(-1, n).contiguous()
        y = target.float().reshape(-1, n).contiguous()
        if x.shape != y.shape:
            raise ValueError(f"input {tuple(pred.shape)} and target {tuple(target.shape)} differ")
        dev = x.device
        desc = _mrstft_desc(x.shape[0], n, cfg["resolutions"], cfg["w_sc"], cfg["w_log_mag"], cfg["w_lin_mag"],
                            cfg["sc_per_example"], cfg["eps"])
        tables = _tables(desc, cfg["resolutions"], dev)
        nbytes = lib.mst_mrstft_workspace_bytes(ctypes.byref(desc))
        if nbytes == 0:
            raise ValueError("unsupported STFT configuration for this input length")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        group = cfg.get("sync_group")
        if group is not None and not cfg["sc_per_example"]:
            # batch rows sharded over ranks + batch-global spectral convergence: the two squared norms are summed over the
            # ranks before the division (include/diffmst_hip.h, mst_mrstft_forward_partial / _finish)
            import torch.distributed as dist

            totals = torch.empty(len(cfg["resolutions"]) * 4, dtype=torch.float64, device=dev)
            with torch.cuda.device(dev):
                _hip.check(lib.mst_mrstft_forward_partial(ctypes.byref(desc), _cabi.ptr(x), _cabi.ptr(y), _cabi.ptr(tables),
                                                          _cabi.ptr(totals), _cabi.ptr(ws), nbytes, _hip.current_stream_ptr(dev)),
                           "mst_mrstft_forward_partial")
            grp = None if group is True else group
            if dist.get_backend(grp) == "gloo":  # CPU collectives: a 96-byte round trip through the host
                host = totals.cpu()
                dist.all_reduce(host, group=grp)
                totals.copy_(host)
            else:
                dist.all_reduce(totals, group=grp)
            with torch.cuda.device(dev):
                _hip.check(lib.mst_mrstft_forward_finish(ctypes.byref(desc), _cabi.ptr(totals), dist.get_world_size(grp),
                                                         _cabi.ptr(loss), _cabi.ptr(ws), nbytes, _hip.current_stream_ptr(dev)),
                           "mst_mrstft_forward_finish")
        else:
            # no gradient asked for (torch.no_grad(), a detached prediction): the value only - the forward then keeps no spectra
            fwd = lib.mst_mrstft_forward if want_grad else lib.mst_mrstft_forward_eval
            with torch.cuda.device(dev):
                _hip.check(fwd(ctypes.byref(desc), _cabi.ptr(x), _cabi.ptr(y), _cabi.ptr(tables), _cabi.ptr(loss),
                               _cabi.ptr(ws), nbytes, _hip.current_stream_ptr(dev)), "mst_mrstft_forward")
        if want_grad:
            ctx.desc, ctx.nbytes, ctx.shape = desc, nbytes, pred.shape
            ctx.save_for_backward(x, y, tables, ws)
            ctx.saved = True
        return loss.reshape(())

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_loss):
        if not ctx.saved:  # nothing was kept (value-only forward): there is no gradient to hand back
            return None, None, None, None
        x, y, tables, ws = ctx.saved_tensors
        lib = _hip.lib()
        dev = x.device
        g = grad_loss.float().reshape(1).contiguous()
        gx = torch.empty_like(x)
        with torch.cuda.device(dev):
            _hip.check(lib.mst_mrstft_backward(ctypes.byref(ctx.desc), _cabi.ptr(x), _cabi.ptr(y), _cabi.ptr(tables), _cabi.ptr(g),
                                               _cabi.ptr(gx), _cabi.ptr(ws), ctx.nbytes, _hip.current_stream_ptr(dev)),
                       "mst_mrstft_backward")
        return gx.view(ctx.shape), None, None, None


class MultiResolutionSTFTLoss(torch.nn.Module):
    """auraloss-compatible multi-resolution STFT loss (spectral convergence + log / linear magnitude L1).

    ``sc_per_example`` selects auraloss 0.4.0's per-example spectral-convergence ratio (default) or the
    batch-global ratio of older releases (SURVEY A.7 - the pinned package is not available to verify).
    Phase loss, mel / chroma scaling, perceptual weighting and scale invariance are not part of the
    reference's configuration and raise ``NotImplementedError``.

    ``sync_group`` (``True`` = the default process group, or a ``torch.distributed`` group; only matters with
    ``sc_per_example=False``): the batch is sharded over the ranks of the group and the batch-global ratio is formed from
    norms summed over all ranks, so that the mean of the rank losses - and rank-averaged gradients - equal the
    single-process values over the global batch.  Every other term is a mean over examples and needs no exchange.
    """

    def __init__(
        self,
        fft_sizes: List[int] = [1024, 2048, 512],
        hop_sizes: List[int] = [120, 240, 50],
        win_lengths: List[int] = [600, 1200, 240],
        window: str = "hann_window",
        w_sc: float = 1.0,
        w_log_mag: float = 1.0,
        w_lin_mag: float = 0.0,
        w_phs: float = 0.0,
        sample_rate: float = None,
        scale: str = None,
        n_bins: int = None,
        perceptual_weighting: bool = False,
        scale_invariance: bool = False,
        eps: float = 1e-8,
        sc_per_example: bool = True,
        sync_group=None,
        **kwargs,
    ):
        super().__init__()
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        if window != "hann_window":
            raise NotImplementedError("only the (default) periodic Hann window is built")
        if w_phs or scale is not None or perceptual_weighting or scale_invariance:
            raise NotImplementedError("phase loss / mel-chroma scaling / perceptual weighting / scale invariance "
                                      "are outside the reference's configuration of this loss")
        self.fft_sizes, self.hop_sizes, self.win_lengths = list(fft_sizes), list(hop_sizes), list(win_lengths)
        self.cfg = dict(
            resolutions=tuple(zip(self.fft_sizes, self.hop_sizes, self.win_lengths)),
            w_sc=w_sc, w_log_mag=w_log_mag, w_lin_mag=w_lin_mag, sc_per_example=sc_per_example, eps=eps,
            sync_group=sync_group,
        )

    def forward(self, x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
        return _MrstftFunction.apply(x, y, self.cfg, torch.is_grad_enabled())


# ------------------------------------------------------------------------------------------------
# AudioFeatureLoss
# ------------------------------------------------------------------------------------------------
AF_KEYS = ("mix-rms", "mix-crest_factor", "mix-stereo_width", "mix-stereo_imbalance", "mix-barkspectrum")
_AF_CACHE = {}


def _af_constants(device, sample_rate):
    """(twiddle/window tables, Bark filterbank) on `device`; built once."""
    key = (str(device), int(sample_rate))
    c = _AF_CACHE.get(key)
    if c is None:
        from .filter import barkscale_fbanks

        lib = _hip.lib()
        tables = torch.empty(lib.mst_afloss_tables_bytes() // 4, dtype=torch.float32, device=device)
        with torch.cuda.device(device):
            _hip.check(lib.mst_afloss_init_tables(_cabi.ptr(tables), _hip.current_stream_ptr(device)), "mst_afloss_init_tables")
        fb = barkscale_fbanks(16385, 20.0, 20000.0, 24, sample_rate).contiguous().to(device)  # reference mst/loss.py:88
        c = _AF_CACHE[key] = (tables, fb)
    return c


class _AudioFeatureFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weights, sample_rate):
        _hip.require_cuda(pred, target)
        lib = _hip.lib()
        if pred.dim() != 3 or pred.shape[1] != 2 or pred.shape != target.shape:
            raise ValueError("AudioFeatureLoss expects (bs, 2, seq_len) input and target of equal shape")
        x = pred.float().contiguous()
        y = target.float().contiguous()
        bs, _, n = x.shape
        dev = x.device
        tables, fb = _af_constants(dev, sample_rate)
        nbytes = lib.mst_afloss_workspace_bytes(bs, n)
        if nbytes == 0:
            raise ValueError("AudioFeatureLoss needs seq_len > 16384 (reflect padding of the 32768-point Bark STFT)")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        losses = torch.empty(5, dtype=torch.float32, device=dev)
        w = (ctypes.c_float * 5)(*[float(v) for v in weights])
        with torch.cuda.device(dev):
            _hip.check(lib.mst_afloss_forward(_cabi.ptr(x), _cabi.ptr(y), bs, n, w, _cabi.ptr(tables), _cabi.ptr(fb),
                                              _cabi.ptr(losses), _cabi.ptr(ws), nbytes, _hip.current_stream_ptr(dev)),
                       "mst_afloss_forward")
        ctx.meta = (bs, n, w, nbytes, pred.shape)
        ctx.save_for_backward(x, y, tables, fb, ws)
        # five 0-dim outputs (views of one buffer) instead of one (5,) tensor the caller would index: indexing costs
        # ~3 tiny kernels per key in forward + backward, 5 x that is more than the closed-form features themselves
        return tuple(losses.unbind(0))

    @staticmethod
    @once_differentiable
    def backward(ctx, *grad_each):
        x, y, tables, fb, ws = ctx.saved_tensors
        bs, n, w, nbytes, shape = ctx.meta
        lib = _hip.lib()
        dev = x.device
        g = torch.stack([gi.float().reshape(()) for gi in grad_each]).contiguous()
        gx = torch.empty_like(x)
        with torch.cuda.device(dev):
            _hip.check(lib.mst_afloss_backward(_cabi.ptr(x), _cabi.ptr(y), bs, n, w, _cabi.ptr(tables), _cabi.ptr(fb), _cabi.ptr(g),
                                               _cabi.ptr(gx), _cabi.ptr(ws), nbytes, _hip.current_stream_ptr(dev)),
                       "mst_afloss_backward")
        return gx.view(shape), None, None, None


class AudioFeatureLoss(torch.nn.Module):
    """Drop-in for reference ``mst.loss.AudioFeatureLoss`` (:198-260).

    ``forward(input, target)`` returns ``{key: weight * mse(feature(input), feature(target))}`` with the
    reference's five keys (``System`` sums ``val.mean()`` over them, mst/system.py:334-336).  All five
    features and their gradients are computed by the kernels of ``csrc/mst_af.hip`` in one pass.
    """

    def __init__(self, weights: List[float], sample_rate: int, stem_separation: bool = False, use_clap: bool = False) -> None:
        super().__init__()
        self.weights = weights
        self.sample_rate = sample_rate
        self.stem_separation = stem_separation
        self.sources_list = ["mix"]
        self.source_weights = [1.0]
        self.use_clap = use_clap
        self.transform_names = ["rms", "crest_factor", "stereo_width", "stereo_imbalance", "barkspectrum"]
        assert len(self.transform_names) == len(weights)

    def forward(self, input: torch.Tensor, target: torch.Tensor):
        losses = _AudioFeatureFunction.apply(input, target, tuple(self.weights), self.sample_rate)
        return dict(zip(AF_KEYS, losses))
