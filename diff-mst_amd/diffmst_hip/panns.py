"""``diffmst_hip.panns`` (alias ``mst.panns``) - the Cnn14 spectrogram encoder on the MI355X matrix cores.

Same module tree and parameter names as the reference (mst/panns.py:27-85 ``ConvBlock``, :126-209 ``Cnn14``):
``conv_block{1..6}.{conv1,conv2}.weight``, ``.bn{1,2}.{weight,bias,running_mean,running_var,num_batches_tracked}``,
``fc.{weight,bias}`` - a reference checkpoint loads with ``load_state_dict`` unchanged.  The modules only HOLD the parameters:
the twelve 3x3 convolutions (implicit GEMM on MFMA), BatchNorm, ReLU, average pooling, the pooling head and the final Linear
all run in ``csrc/mst_cnn*.hip`` through ``mst_cnn14_forward`` / ``mst_cnn14_backward`` (include/diffmst_hip.h), forward and
reverse mode; autograd sees ONE function.

``precision``: ``"fp32"`` (default, the reference's ``precision: 32``, configs/config.yaml:42) - fp32 operands on the fp32 MFMA;
``"bf16"`` (opt-in: constructor keyword, or ``MST_ENCODER_PRECISION=bf16`` in the environment for an unmodified YAML) - bf16 operands and
activations, fp32 accumulation / statistics / gradients: 5x faster, training-mode weight gradients 20-40 % from the fp32 ones;
``"bf16x6"`` / ``"bf16x3"`` - fp32 tensors everywhere, the convolutions' operands split into bf16 (hi, mid, lo) / (hi, lo) pieces in
registers and multiplied term by term on the bf16 MFMA with fp32 accumulation: bf16x6 keeps every product term down to 2^-24 and passes
the fp32 path's parity bounds at 0.84x its time; bf16x3 (~2^-17 per product) runs at 0.68x.
"""
from __future__ import annotations

import ctypes

import torch
import torch.nn as nn
from torch.autograd.function import once_differentiable

from . import _cabi, _hip

_CHANNELS = (64, 128, 256, 512, 1024, 2048)
# pool sizes over (bins, frames) as the reference passes them (mst/panns.py:186-197)
POOL_SIZES = ((2, 2), (4, 4), (4, 2), (4, 2), (4, 2), (2, 2))


# descriptor codes (include/diffmst_hip.h, mst_cnn14_desc::precision)
PRECISIONS = {"bf16": 0, "fp32": 1, "bf16x3": 2, "bf16x6": 3}


def default_precision() -> str:
    """The encoder precision of a model built without the keyword: fp32 like the reference, unless the environment opts in."""
    import os

    p = os.environ.get("MST_ENCODER_PRECISION", "fp32")
    if p not in PRECISIONS:
        raise ValueError("MST_ENCODER_PRECISION must be 'bf16', 'bf16x3', 'bf16x6' or 'fp32'")
    return p


def init_layer(layer):
    nn.init.xavier_uniform_(layer.weight)
    if getattr(layer, "bias", None) is not None:
        layer.bias.data.fill_(0.0)


def init_bn(bn):
    bn.bias.data.fill_(0.0)
    bn.weight.data.fill_(1.0)


class ConvBlock(nn.Module):
    """Parameter holder of one block (reference mst/panns.py:27-85); evaluated inside the fused Cnn14 call."""

    def __init__(self, in_channels: int, out_channels: int, use_batchnorm: bool = True, pool_type: str = "avg"):
        super().__init__()
        if not use_batchnorm or pool_type != "avg":
            raise NotImplementedError("the MI355X encoder builds the reference's configuration: BatchNorm on, average pooling")
        self.use_batchnorm = use_batchnorm
        self.conv1 = nn.Conv2d(in_channels, out_channels, (3, 3), (1, 1), (1, 1), bias=False)
        self.conv2 = nn.Conv2d(out_channels, out_channels, (3, 3), (1, 1), (1, 1), bias=False)
        self.bn1 = nn.BatchNorm2d(out_channels)
        self.bn2 = nn.BatchNorm2d(out_channels)
        for conv in (self.conv1, self.conv2):
            init_layer(conv)
        for bn in (self.bn1, self.bn2):
            init_bn(bn)

    def forward(self, *_a, **_k):
        raise RuntimeError("ConvBlock is evaluated by Cnn14's fused kernels; call the Cnn14 module")


def sync_group_of(module: "Cnn14"):
    """(process group, world size) over which this encoder's BatchNorm statistics are shared, (None, 1) when they are not.

    The reference trains with Lightning's ``sync_batchnorm: true`` (configs/config.yaml:41), i.e.
    ``torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)``: the ``nn.BatchNorm2d`` parameter holders of the ConvBlocks become
    ``nn.SyncBatchNorm`` instances (same parameter names).  The fused kernels honour exactly that: if the holders are SyncBatchNorm
    modules and a process group of more than one rank is up, every layer's statistics - and their adjoints - are all-reduced over
    the holders' ``process_group`` (``mst_cnn14_forward_sync``)."""
    import torch.distributed as dist

    bns = module._bns()
    if not any(isinstance(bn, nn.SyncBatchNorm) for bn in bns):
        return None, 1
    if not all(isinstance(bn, nn.SyncBatchNorm) for bn in bns):
        raise RuntimeError("Cnn14: either every BatchNorm of the encoder is a SyncBatchNorm or none is")
    if not (dist.is_available() and dist.is_initialized()):
        return None, 1
    group = bns[0].process_group
    world = dist.get_world_size(group)
    return (group, world) if world > 1 else (None, 1)


_COUNT_STREAMS = {}


def _sync_count_check(group, n: int, dev):
    """The cross-rank statistics count ``world x n`` signals and the per-layer exchanges are sized by ``n`` (torch's SyncBatchNorm
    exchanges the counts; these kernels assume them equal): ranks that feed different ``n`` - a short last batch with drop_last=False, a
    variable track count - must not reach the statistics exchange at all (gloo aborts on the size mismatch, RCCL hangs or corrupts).
    EVERY training call of EVERY rank therefore runs the same tiny collective first - an all-reduce (MAX) of ``[n, -n]`` - and reads its
    verdict before anything else is enqueued.  No collective is gated on rank-local state (a per-``n`` cache let a rank that had seen its
    ``n`` skip the exchange its peer was waiting in: advisor, round 5), and the verdict is a function of the reduced values only
    (max n != min n): every rank raises, at the same call, in front of the same collective.  The exchange runs on a stream of its own,
    so the host waits for one collective's latency, not for the compute stream's backlog."""
    import torch.distributed as dist

    key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
    side = _COUNT_STREAMS.get(key)
    if side is None:
        side = _COUNT_STREAMS[key] = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        t = torch.tensor([n, -n], dtype=torch.int64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        hi, neg_lo = (int(v) for v in t.tolist())
    if hi != -neg_lo:
        raise RuntimeError(f"Cnn14 with SyncBatchNorm: every rank must feed the same number of signals per call (this rank {n}, "
                           f"ranks between {-neg_lo} and {hi}); pad the batch or use drop_last=True")

class _StatSync:
    """The ``mst_sync_fn`` of one kernel call: all-reduces (SUM) a span of the call's workspace over the process group."""

    def __init__(self, ws: torch.Tensor, group):
        self.ws, self.group, self.base, self.error = ws, group, ws.data_ptr(), None
        self.fn = _cabi.SYNC_FN(self._call)

    def _call(self, _user, ptr, n_doubles, _stream):
        try:  # an exception must not unwind through the C frames: keep it, the caller re-raises
            import torch.distributed as dist

            off = ptr - self.base
            dist.all_reduce(self.ws[off:off + 8 * n_doubles].view(torch.float64), group=self.group)
        except BaseException as e:  # noqa: BLE001
            self.error = self.error or e

    def check(self):
        if self.error is not None:
            raise self.error


class _Cnn14Function(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec, module, training, *params):
        """spec (n, frames, bins) fp32; params = 12 conv weights, 12 BN weights, 12 BN biases, fc weight, fc bias."""
        bns = module._bns()
        # every pointer below goes straight to the kernels: a model left on the host (load_diffmst maps checkpoints to the CPU) must
        # raise here, not fault on the device
        _hip.require_cuda(spec, *params, *(bn.running_mean for bn in bns), *(bn.running_var for bn in bns))
        lib = _hip.lib()
        dev = spec.device
        _hip.require_same_device(dev, *params, *(bn.running_mean for bn in bns), *(bn.running_var for bn in bns))
        n, frames, bins = spec.shape
        group, world = sync_group_of(module) if training else (None, 1)
        if world > 1:
            _sync_count_check(group, n, dev)
        desc = _cabi.Cnn14Desc(n, frames, bins, module.fc.out_features, PRECISIONS[module.precision], int(training),
                               float(module.conv_block1.bn1.eps), world)
        nbytes = lib.mst_cnn14_workspace_bytes(ctypes.byref(desc))
        if nbytes == 0:
            raise ValueError(f"Cnn14: unsupported spectrogram size {(frames, bins)} (six pooling stages need >= 128 frames x 1024 bins)")
        convs, gammas, betas = params[0:12], params[12:24], params[24:36]
        fc_w, fc_b = params[36], params[37]
        keep = [t.detach().float().contiguous() for t in (*convs, *gammas, *betas, fc_w, fc_b)]
        rmean = [bn.running_mean.detach().float().contiguous() for bn in bns]
        rvar = [bn.running_var.detach().float().contiguous() for bn in bns]
        prm = _cabi.Cnn14Params()
        for i in range(12):
            prm.conv_w[i] = keep[i].data_ptr()
            prm.bn_gamma[i] = keep[12 + i].data_ptr()
            prm.bn_beta[i] = keep[24 + i].data_ptr()
            prm.bn_mean[i] = rmean[i].data_ptr()
            prm.bn_var[i] = rvar[i].data_ptr()
        prm.fc_w, prm.fc_b = keep[36].data_ptr(), keep[37].data_ptr()
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        embed = torch.empty(n, module.fc.out_features, dtype=torch.float32, device=dev)
        stats = torch.empty(12, 2, 2048, dtype=torch.float32, device=dev) if training else None
        spec = spec.float().contiguous()
        sync = _StatSync(ws, group) if world > 1 else None
        with torch.cuda.device(dev):
            rc = lib.mst_cnn14_forward_sync(ctypes.byref(desc), _cabi.ptr(spec), ctypes.byref(prm), _cabi.ptr(embed), _cabi.ptr(stats),
                                            _cabi.ptr(ws), nbytes, _hip.current_stream_ptr(dev), sync.fn if sync else _cabi.SYNC_FN(), None)
        if sync:
            sync.check()
        _hip.check(rc, "mst_cnn14_forward")
        ctx.desc, ctx.nbytes, ctx.dev, ctx.group = desc, nbytes, dev, group
        # the backward reads mean / invstd from the workspace; the running statistics (updated in place by the module right
        # after a training-mode call) are only handed over again, never read in training mode
        ctx.running = (rmean, rvar)
        ctx.save_for_backward(spec, ws, *keep)
        ctx.mark_non_differentiable(*([stats] if stats is not None else []))
        return (embed, stats) if training else (embed, None)

    @staticmethod
    @once_differentiable
    def backward(ctx, g_embed, _g_stats):
        spec, ws, *rest = ctx.saved_tensors
        keep = rest[:38]
        rmean, rvar = ctx.running
        lib = _hip.lib()
        dev = ctx.dev
        prm, gr = _cabi.Cnn14Params(), _cabi.Cnn14Grads()
        grads = [torch.empty_like(t) for t in keep]
        for i in range(12):
            prm.conv_w[i], prm.bn_gamma[i], prm.bn_beta[i] = keep[i].data_ptr(), keep[12 + i].data_ptr(), keep[24 + i].data_ptr()
            prm.bn_mean[i], prm.bn_var[i] = rmean[i].data_ptr(), rvar[i].data_ptr()
            gr.conv_w[i], gr.bn_gamma[i], gr.bn_beta[i] = grads[i].data_ptr(), grads[12 + i].data_ptr(), grads[24 + i].data_ptr()
        prm.fc_w, prm.fc_b = keep[36].data_ptr(), keep[37].data_ptr()
        gr.fc_w, gr.fc_b = grads[36].data_ptr(), grads[37].data_ptr()
        g = g_embed.float().contiguous()
        sync = _StatSync(ws, ctx.group) if ctx.desc.world > 1 else None
        with torch.cuda.device(dev):
            rc = lib.mst_cnn14_backward_sync(ctypes.byref(ctx.desc), _cabi.ptr(spec), ctypes.byref(prm), _cabi.ptr(g), ctypes.byref(gr),
                                             _cabi.ptr(ws), ctx.nbytes, _hip.current_stream_ptr(dev), sync.fn if sync else _cabi.SYNC_FN(), None)
        if sync:
            sync.check()
        _hip.check(rc, "mst_cnn14_backward")
        return (None, None, None, *grads)


class Cnn14(nn.Module):
    """Drop-in for reference ``mst.panns.Cnn14`` (:126-209): ``(bs, 1, bins, frames)`` spectrogram -> ``(bs, num_classes)``."""

    def __init__(self, num_classes: int, n_inputs: int = 1, use_batchnorm: bool = True, precision: str | None = None):
        super().__init__()
        precision = default_precision() if precision is None else precision
        if n_inputs != 1:
            raise NotImplementedError("n_inputs = 1 (the reference's SpectrogramEncoder default) is what the first-layer kernel is built for")
        if precision not in PRECISIONS:
            raise ValueError("precision must be 'bf16', 'bf16x3', 'bf16x6' or 'fp32'")
        self.precision = precision
        c_in = n_inputs
        for i, c_out in enumerate(_CHANNELS, start=1):
            setattr(self, f"conv_block{i}", ConvBlock(c_in, c_out, use_batchnorm=use_batchnorm))
            c_in = c_out
        self.fc = nn.Linear(2048, num_classes, bias=True)
        init_layer(self.fc)

    def _blocks(self):
        return [getattr(self, f"conv_block{i}") for i in range(1, 7)]

    def _bns(self):
        return [bn for b in self._blocks() for bn in (b.bn1, b.bn2)]

    def _parameters_in_abi_order(self):
        blocks = self._blocks()
        convs = [c.weight for b in blocks for c in (b.conv1, b.conv2)]
        bns = self._bns()
        return (*convs, *(bn.weight for bn in bns), *(bn.bias for bn in bns), self.fc.weight, self.fc.bias)

    def forward_frames_major(self, spec: torch.Tensor) -> torch.Tensor:
        """spec (n, frames, bins): the layout the STFT kernel emits and the kernels consume."""
        training = self.training
        embed, stats = _Cnn14Function.apply(spec, self, training, *self._parameters_in_abi_order())
        if training:  # nn.BatchNorm2d bookkeeping (momentum 0.1, unbiased running variance), on the batch statistics of this call
            n, frames, bins = spec.shape
            n = n * sync_group_of(self)[1]  # SyncBatchNorm: the statistics (and the unbiased correction) span every rank's signals
            h, w = frames, bins
            with torch.no_grad():
                # all twelve layers in four multi-tensor launches (48 one-tensor kernels otherwise); same arithmetic as
                # nn.BatchNorm2d: running.lerp_(batch, momentum) with the unbiased batch variance
                rmeans, rvars, means, vars_, scales, counters, moms = [], [], [], [], [], [], []
                for i, block in enumerate(self._blocks()):
                    count = n * h * w
                    for k, bn in enumerate((block.bn1, block.bn2)):
                        c = bn.num_features
                        rmeans.append(bn.running_mean)
                        rvars.append(bn.running_var)
                        means.append(stats[2 * i + k, 0, :c])
                        vars_.append(stats[2 * i + k, 1, :c])
                        scales.append(count / max(count - 1, 1))
                        counters.append(bn.num_batches_tracked)
                        moms.append(bn.momentum)
                    h, w = h // POOL_SIZES[i][1], w // POOL_SIZES[i][0]
                if all(m is not None and m == moms[0] for m in moms):
                    torch._foreach_lerp_(rmeans, means, moms[0])
                    torch._foreach_lerp_(rvars, torch._foreach_mul(vars_, scales), moms[0])
                    torch._foreach_add_(counters, 1)
                else:  # cumulative averages (momentum None) or per-layer momenta: one layer at a time
                    for rm, rv, mean, var, sc, nb, mom in zip(rmeans, rvars, means, vars_, scales, counters, moms):
                        m = mom if mom is not None else 1.0 / float(nb + 1)
                        rm.lerp_(mean, m)
                        rv.lerp_(var * sc, m)
                        nb += 1
        return embed

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        bs, chs, bins, frames = x.size()
        if chs != 1:
            raise NotImplementedError("n_inputs = 1")
        return self.forward_frames_major(x.reshape(bs, bins, frames).transpose(1, 2).contiguous())
