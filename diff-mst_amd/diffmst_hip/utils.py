"""``mst.utils`` - the helper on the hot path (reference mst/utils.py:14-29)."""
from __future__ import annotations

import torch
from torch.autograd.function import once_differentiable

from . import _cabi, _hip


class _PeakNormalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _hip.require_cuda(x)
        lib = _hip.lib()
        if x.dim() != 3:
            raise ValueError("expected a (bs, chs, seq_len) tensor")
        shape = x.shape
        xc = x.float().contiguous()
        if shape[1] != 2:
            # the peak runs over (channels, time) of a batch item, so any channel count is the stereo kernel on a
            # (bs, 2, chs*n/2) view of the same memory (reference mst/system.py:390-391 normalises a MONO sum)
            flat = xc.view(shape[0], -1)
            if flat.size(1) % 2:  # odd element count: one zero sample of padding leaves the peak unchanged
                flat = torch.nn.functional.pad(flat, (0, 1))
            xc = flat.view(shape[0], 2, flat.size(1) // 2)
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
        ctx.shape = shape
        return y.view(shape[0], -1)[:, : shape[1] * shape[2]].reshape(shape)

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        xc, ws = ctx.saved_tensors
        lib = _hip.lib()
        bs, _, n = xc.shape
        dev = xc.device
        g = g.float().contiguous().view(ctx.shape[0], -1)
        if g.size(1) != 2 * n:
            g = torch.nn.functional.pad(g, (0, 2 * n - g.size(1)))
        g = g.contiguous().view(xc.shape)
        dx = torch.empty_like(xc)
        with torch.cuda.device(dev):
            _hip.check(lib.mst_peak_normalize_backward(_cabi.ptr(xc), _cabi.ptr(g), _cabi.ptr(dx), bs, n, _cabi.ptr(ws),
                                                       ctx.nbytes, _hip.current_stream_ptr(dev)), "mst_peak_normalize_backward")
        shape = ctx.shape
        return dx.view(shape[0], -1)[:, : shape[1] * shape[2]].reshape(shape)


def batch_stereo_peak_normalize(x: torch.Tensor):
    """Normalize a batch of mixes ``(bs, chs, seq_len)`` by their peak value over (chs, seq_len), per batch item."""
    return _PeakNormalize.apply(x)
