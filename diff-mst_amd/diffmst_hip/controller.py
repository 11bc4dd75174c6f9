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


# ---------------------------------------------------------------------------------------------------------------------------------
# The whole TransformerController.forward as ONE autograd node (round 4): token assembly + mask extension, the encoder stack, the three
# sigmoid heads - nothing of it on torch / rocBLAS (csrc/mst_ctrl.hip: k_ctrl_tokens, k_ctrl_heads_*; reference mst/modules.py:866-914)
# ---------------------------------------------------------------------------------------------------------------------------------
_IO_NAMES = ("track_embedding", "mix_embedding", "fx_bus_embedding", "master_bus_embedding", "track_projection.weight",
             "track_projection.bias", "fx_bus_projection.weight", "fx_bus_projection.bias", "master_bus_projection.weight",
             "master_bus_projection.bias")


def io_parameters(ctrl: torch.nn.Module):
    """The controller's own ten parameters in ``mst_ctrl_io`` order (looked up per call: ten attribute walks)."""
    out = []
    for n in _IO_NAMES:
        obj = ctrl
        for part in n.split("."):
            obj = getattr(obj, part)
        out.append(obj)
    return out


def _io_struct(tensors):
    s = _cabi.CtrlIO()
    for name, t in zip(_cabi.CTRL_IO_FIELDS, tensors):
        setattr(s, name, t.data_ptr() if t is not None else None)
    return s


def _f32c(p):
    return p.detach() if (p.dtype is torch.float32 and p.is_contiguous()) else p.detach().float().contiguous()


class _Controller(torch.autograd.Function):
    @staticmethod
    def forward(ctx, track_embeds, mix_embeds, mask, desc, *params):
        _hip.require_cuda(track_embeds, mix_embeds, mask, *params)
        lib = _hip.lib()
        dev = track_embeds.device
        _hip.require_same_device(dev, mix_embeds, mask, *params)
        bs, T, D = track_embeds.shape
        te, me = track_embeds.float().contiguous(), mix_embeds.float().contiguous()
        ps = [_f32c(p) for p in params]
        io, layers = ps[:10], ps[10:]
        nt, nf, nm = io[4].shape[0], io[6].shape[0], io[8].shape[0]
        m_in = None
        if mask is not None:
            m_in = mask.contiguous().view(torch.uint8) if mask.dtype is torch.bool else (mask != 0).contiguous().view(torch.uint8)
        tokens = torch.empty(bs, T + 4, D, dtype=torch.float32, device=dev)
        m_ext = torch.empty(bs, T + 4, dtype=torch.uint8, device=dev) if mask is not None else None
        nbytes = lib.mst_ctrl_workspace_bytes(ctypes.byref(desc))
        if nbytes == 0:
            raise ValueError("TransformerController: this encoder stack is outside the kernels' limits (controller.supported)")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        z = torch.empty_like(tokens)
        out_t = torch.empty(bs, T, nt, dtype=torch.float32, device=dev)
        out_f = torch.empty(bs, nf, dtype=torch.float32, device=dev)
        out_m = torch.empty(bs, nm, dtype=torch.float32, device=dev)
        io_s = _io_struct(io)
        st = _hip.current_stream_ptr(dev)
        with torch.cuda.device(dev):
            _hip.check(lib.mst_ctrl_tokens_forward(ctypes.byref(desc), T, _cabi.ptr(te), _cabi.ptr(me), _cabi.ptr(m_in), ctypes.byref(io_s),
                                                   _cabi.ptr(tokens), _cabi.ptr(m_ext), st), "mst_ctrl_tokens_forward")
            _hip.check(lib.mst_ctrl_forward(ctypes.byref(desc), _cabi.ptr(tokens), _cabi.ptr(m_ext), _layer_array(layers, desc.n_layers),
                                            _cabi.ptr(z), _cabi.ptr(ws), nbytes, st), "mst_ctrl_forward")
            _hip.check(lib.mst_ctrl_heads_forward(ctypes.byref(desc), T, _cabi.ptr(z), ctypes.byref(io_s), nt, nf, nm, _cabi.ptr(out_t),
                                                  _cabi.ptr(out_f), _cabi.ptr(out_m), st), "mst_ctrl_heads_forward")
        ctx.desc, ctx.nbytes, ctx.T, ctx.heads = desc, nbytes, T, (nt, nf, nm)
        ctx.save_for_backward(tokens, z, ws, out_t, out_f, out_m, *ps)
        ctx.set_materialize_grads(False)  # an unused head (fx bus off) hands None down, and its projection reports None like autograd
        return out_t, out_f, out_m

    @staticmethod
    @once_differentiable
    def backward(ctx, g_t, g_f, g_m):
        tokens, z, ws, out_t, out_f, out_m, *ps = ctx.saved_tensors
        io, layers = ps[:10], ps[10:]
        lib = _hip.lib()
        dev = tokens.device
        desc, T = ctx.desc, ctx.T
        nt, nf, nm = ctx.heads
        g_t = torch.zeros_like(out_t) if g_t is None else g_t.float().contiguous()
        g_f = None if g_f is None else g_f.float().contiguous()
        g_m = None if g_m is None else g_m.float().contiguous()
        io_g = [torch.empty_like(p) for p in io]
        if g_f is None:
            io_g[6] = io_g[7] = None
        if g_m is None:
            io_g[8] = io_g[9] = None
        layer_g = [torch.empty_like(p) for p in layers]
        gz = torch.empty_like(z)
        gtok = torch.empty_like(tokens)
        scratch = torch.empty(lib.mst_ctrl_heads_scratch_bytes(ctypes.byref(desc), T), dtype=torch.uint8, device=dev)
        io_s, iog_s = _io_struct(io), _io_struct(io_g)
        st = _hip.current_stream_ptr(dev)
        with torch.cuda.device(dev):
            _hip.check(lib.mst_ctrl_heads_backward(ctypes.byref(desc), T, _cabi.ptr(z), ctypes.byref(io_s), nt, nf, nm, _cabi.ptr(out_t),
                                                   _cabi.ptr(out_f), _cabi.ptr(out_m), _cabi.ptr(g_t), _cabi.ptr(g_f), _cabi.ptr(g_m),
                                                   ctypes.byref(iog_s), _cabi.ptr(gz), _cabi.ptr(scratch), st), "mst_ctrl_heads_backward")
            _hip.check(lib.mst_ctrl_backward(ctypes.byref(desc), _cabi.ptr(tokens), _layer_array(layers, desc.n_layers), _cabi.ptr(gz),
                                             _layer_array(layer_g, desc.n_layers), _cabi.ptr(gtok), _cabi.ptr(ws), ctx.nbytes, st),
                       "mst_ctrl_backward")
            _hip.check(lib.mst_ctrl_tokens_backward(ctypes.byref(desc), T, _cabi.ptr(gtok), ctypes.byref(iog_s), st), "mst_ctrl_tokens_backward")
        return (gtok[:, :T], gtok[:, T:T + 2], None, None, *io_g, *layer_g)


def controller_forward(ctrl: torch.nn.Module, track_embeds: torch.Tensor, mix_embeds: torch.Tensor, track_padding_mask=None):
    """``TransformerController.forward`` entirely on csrc/mst_ctrl.hip: (track params, fx-bus params, master-bus params)."""
    bs, T, _ = track_embeds.shape
    return _Controller.apply(track_embeds, mix_embeds, track_padding_mask, _desc(ctrl.transformer_encoder, bs, T + 4), *io_parameters(ctrl),
                             *layer_parameters(ctrl.transformer_encoder))


def heads_supported(ctrl: torch.nn.Module) -> bool:
    return (ctrl.embed_dim <= 1024 and max(ctrl.num_track_control_params, ctrl.num_fx_bus_control_params, ctrl.num_master_bus_control_params) <= 32)
