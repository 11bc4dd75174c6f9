"""``mst.utils`` - the helper on the hot path (reference mst/utils.py:14-29)."""
from __future__ import annotations

import torch

from . import _cabi, _hip


class _PeakNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _hip.require_cuda(x)
        lib = _hip.lib()
        if x.dim() != 3 or x.shape[1] != 2:
            raise ValueError("expected a (bs, 2, seq_len) tensor")
        xc = x.float().contiguous()
        bs, _, n = xc.shape
        dev = xc.device
        nbytes = lib.mst_peak_normalize_workspace_bytes(bs, n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        y = torch.empty_like(xc)
        with torch.cuda.device(dev):
            _hip.check(lib.mst_peak_normalize_forward(_cabi.ptr(xc), _cabi.ptr(y), bs, n, _cabi.ptr(ws), nbytes,
                                                      _hip.current_stream_ptr(dev)), "mst_peak_normalize_forward")
        ctx.save_for_backward(xc, ws)
        ctx.nbytes = nbytes
        return y

    @staticmethod
    def backward(ctx, g):
        xc, ws = ctx.saved_tensors
        lib = _hip.lib()
        bs, _, n = xc.shape
        dev = xc.device
        g = g.float().contiguous()
        dx = torch.empty_like(xc)
        with torch.cuda.device(dev):
            _hip.check(lib.mst_peak_normalize_backward(_cabi.ptr(xc), _cabi.ptr(g), _cabi.ptr(dx), bs, n, _cabi.ptr(ws),
                                                       ctx.nbytes, _hip.current_stream_ptr(dev)), "mst_peak_normalize_backward")
        return dx


def batch_stereo_peak_normalize(x: torch.Tensor):
    """Normalize a batch of stereo mixes ``(bs, 2, seq_len)`` by their peak value (per batch item)."""
    return _PeakNormalize.apply(x)
