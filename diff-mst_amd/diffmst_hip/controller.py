"""The TransformerController's encoder stack on the HIP kernels of ``csrc/mst_ctrl.hip`` (C ABI ``mst_ctrl_forward`` /
``mst_ctrl_backward``): one ``torch.autograd.Function`` for all layers of a ``torch.nn.TransformerEncoder`` built the way the
reference builds it (mst/modules.py:848-854: post-norm layers, relu, dropout 0, batch_first, no final norm)."""
import ctypes

import torch
from torch.autograd.function import once_differentiable

from . import _cabi, _hip

_PARAM_NAMES = ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias",
                "norm2.weight", "norm2.bias")


def layer_parameters(encoder: torch.nn.TransformerEncoder):
    """The twelve parameters of every layer, in ``mst_ctrl_layer`` order (a flat list, layer-major)."""
    # 144 attribute walks per call are not free, so the list is cached - and validated by IDENTITY against the modules' own parameter
    # dictionaries (load_state_dict(assign=True), parametrizations, swap_tensors or a plain `layer.linear1.weight = ...` replace the
    # objects; a stale list would compute with, and route gradients to, tensors the module no longer owns)
    cached = encoder.__dict__.get("_mst_layer_parameters")
    if cached is not None and len(cached[0]) == len(_PARAM_NAMES) * len(encoder.layers):
        flat, owners = cached
        if all(o[k] is p for (o, k), p in zip(owners, flat)):
            return flat
    flat, owners = [], []
    for layer in encoder.layers:
        for n in _PARAM_NAMES:
            *path, leaf = n.split(".")
            mod = layer
            for part in path:
                mod = getattr(mod, part)
            flat.append(mod._parameters[leaf])
            owners.append((mod._parameters, leaf))
    encoder.__dict__["_mst_layer_parameters"] = (flat, owners)
    return flat


def supported(encoder: torch.nn.TransformerEncoder, bs: int, seq: int) -> bool:
    """Whether the kernels cover this stack (else the caller stays on torch's own layers)."""
    layer = encoder.layers[0]
    if encoder.norm is not None or layer.norm_first or layer.dropout.p != 0.0 or not layer.self_attn.batch_first:
        return False
    if getattr(layer, "activation_relu_or_gelu", 0) != 1 or layer.self_attn.in_proj_weight is None:
        return False
    return _desc(encoder, bs, seq, probe=True) is not None


def _desc(encoder, bs, seq, probe=False):
    layer = encoder.layers[0]
    d = _cabi.CtrlDesc(bs, seq, layer.self_attn.embed_dim, layer.self_attn.num_heads, layer.linear1.out_features,
                       len(encoder.layers), float(layer.norm1.eps))
    if probe:
        return d if _hip.lib().mst_ctrl_workspace_bytes(ctypes.byref(d)) else None
    return d


def _layer_array(tensors, n_layers):
    arr = (_cabi.CtrlLayer * n_layers)()
    k = len(_cabi.CTRL_FIELDS)
    for l in range(n_layers):
        for j, name in enumerate(_cabi.CTRL_FIELDS):
            setattr(arr[l], name, tensors[l * k + j].data_ptr())
    return arr


class _EncoderStack(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tokens, mask, desc, *params):
        _hip.require_cuda(tokens, mask, *params)  # host pointers must raise here, not fault on the device
        lib = _hip.lib()
        dev = tokens.device
        _hip.require_same_device(dev, mask, *params)
        x = tokens.float().contiguous()
        ps = [p.detach() if (p.dtype is torch.float32 and p.is_contiguous()) else p.detach().float().contiguous() for p in params]
        m = None  # (bs, seq) bytes, non-zero = padded key; a bool tensor is viewed, not converted
        if mask is not None:
            m = mask.contiguous().view(torch.uint8) if mask.dtype is torch.bool else (mask != 0).contiguous().view(torch.uint8)
        nbytes = lib.mst_ctrl_workspace_bytes(ctypes.byref(desc))
        if nbytes == 0:
            raise ValueError("TransformerController: this encoder stack is outside the kernels' limits (controller.supported)")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        out = torch.empty_like(x)
        with torch.cuda.device(dev):
            rc = lib.mst_ctrl_forward(ctypes.byref(desc), _cabi.ptr(x), _cabi.ptr(m), _layer_array(ps, desc.n_layers), _cabi.ptr(out),
                                      _cabi.ptr(ws), nbytes, _hip.current_stream_ptr(dev))
        _hip.check(rc, "mst_ctrl_forward")
        ctx.desc, ctx.nbytes = desc, nbytes
        ctx.save_for_backward(x, ws, *ps)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, ws, *ps = ctx.saved_tensors
        lib = _hip.lib()
        dev = x.device
        g = grad_out.float().contiguous()
        grads = [torch.empty_like(p) for p in ps]
        gx = torch.empty_like(x)
        with torch.cuda.device(dev):
            rc = lib.mst_ctrl_backward(ctypes.byref(ctx.desc), _cabi.ptr(x), _layer_array(ps, ctx.desc.n_layers), _cabi.ptr(g),
                                       _layer_array(grads, ctx.desc.n_layers), _cabi.ptr(gx), _cabi.ptr(ws), ctx.nbytes,
                                       _hip.current_stream_ptr(dev))
        _hip.check(rc, "mst_ctrl_backward")
        return (gx, None, None, *grads)


def encoder_stack(encoder: torch.nn.TransformerEncoder, tokens: torch.Tensor, key_padding_mask=None) -> torch.Tensor:
    """``encoder(tokens, src_key_padding_mask=key_padding_mask)`` on the HIP kernels (training and eval: dropout is 0)."""
    bs, seq, _ = tokens.shape
    return _EncoderStack.apply(tokens, key_padding_mask, _desc(encoder, bs, seq), *layer_parameters(encoder))
