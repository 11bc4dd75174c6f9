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


# ------------------------------------------------------------------------------------------------
# Inference driver (reference mst/utils.py:32-258): forward-only, batch 1, long songs
# ------------------------------------------------------------------------------------------------
ANALYSIS_LEN = 262144  # reference mst/utils.py:66


def _default_loudness_fn(sample_rate=44100):
    """``pyloudnorm.Meter(44100).integrated_loudness`` like the reference (mst/utils.py:67, :93); the package is a host-side
    dependency of the reference, not part of the device path - inject ``loudness_fn`` where it is not installed."""
    try:
        import pyloudnorm as pyln
    except ImportError as e:  # fail loudly: there is no silent stand-in for BS.1770 loudness
        raise ImportError("run_diffmst needs pyloudnorm (reference requirements.txt) or an explicit loudness_fn(ndarray (n, ch)) -> LUFS") from e
    return pyln.Meter(sample_rate).integrated_loudness


def run_diffmst(tracks: torch.Tensor, ref: torch.Tensor, model: torch.nn.Module, mix_console: torch.nn.Module,
                track_start_idx: int = 0, ref_start_idx: int = 0, loudness_fn=None, device=None, verbose: bool = False):
    """Reference ``mst.utils.run_diffmst`` (mst/utils.py:32-173) on the HIP console.

    ``tracks (1, T, n)``, ``ref (1, 2, n_ref)`` (host or device tensors) ->
    ``(pred_mix (1, 2, n), track / fx-bus / master-bus parameter dictionaries)`` with the reference's steps: crop an analysis
    window of 262144 samples, normalise every track to -48 LUFS from its analysis crop (tracks below -80 LUFS are dropped),
    ONE parameter estimate ``model(analysis_tracks, analysis_ref)``, then console forwards over 262144-sample windows
    hopping by 131072, each faded with a periodic Hann window (the first window's first half held at 1) and overlap-added.

    Differences, all outside the arithmetic: the caller's ``tracks`` is not scaled in place (the reference's ``track *= ...``
    writes through a view); model and console windows run on ``device`` (default: the model's device if it is a GPU, else the
    current GPU; a model still on the host - what ``load_diffmst`` returns - is moved there with ``model.to(device)``) and
    ``pred_mix`` comes back on ``tracks.device``; ``loudness_fn(ndarray (n, 1)) -> float`` replaces the
    pyloudnorm meter where that package is absent (it is host-side in the reference too)."""
    if tracks.dim() != 3 or tracks.shape[0] != 1:
        raise ValueError("tracks must be (1, num_tracks, seq_len)")  # the reference's squeeze(0) / zeros(1, 2, n) fix bs = 1
    if loudness_fn is None:
        loudness_fn = _default_loudness_fn()
    if device is None:
        p = next(iter(model.parameters()), None) if isinstance(model, torch.nn.Module) else None
        device = p.device if p is not None and p.is_cuda else torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    if isinstance(model, torch.nn.Module):
        # load_diffmst returns the model on the host (map_location="cpu", like the reference, which runs there); this package has no
        # host path, so the model follows the audio onto the device - in place, like nn.Module.to
        if any(t.device != device for t in (*model.parameters(), *model.buffers())):
            model.to(device)
    n = tracks.shape[-1]
    if n >= ANALYSIS_LEN:
        analysis_tracks = tracks[..., track_start_idx:track_start_idx + ANALYSIS_LEN]
    else:
        analysis_tracks = tracks
    analysis_ref = ref[..., ref_start_idx:ref_start_idx + ANALYSIS_LEN] if ref.shape[-1] >= ANALYSIS_LEN else ref

    # loudness-normalise to -48 LUFS (host side, like the reference: the meter works on numpy)
    keep, gains = [], []
    host_analysis = analysis_tracks.detach().float().cpu()
    for t in range(tracks.shape[1]):
        lufs_db = float(loudness_fn(host_analysis[0, t:t + 1].permute(1, 0).numpy()))
        if lufs_db < -80.0:
            if verbose:
                print(f"Skipping track {t} due to low loudness {lufs_db}.")
            continue
        keep.append(t)
        gains.append(10 ** ((-48 - lufs_db) / 20))
    if not keep:
        raise RuntimeError("every track is below -80 LUFS")  # the reference fails in torch.cat([]) here
    idx = torch.tensor(keep, device=tracks.device)
    g = torch.tensor(gains, dtype=torch.float32).view(1, -1, 1)
    norm_tracks = (tracks.detach().float().index_select(1, idx) * g.to(tracks.device)).to(device).contiguous()
    norm_analysis = (analysis_tracks.detach().float().index_select(1, idx) * g.to(tracks.device)).to(device).contiguous()

    # ---- one parameter estimate from the analysis audio
    pred_track_params, pred_fx_bus_params, pred_master_bus_params = model(norm_analysis, analysis_ref.float().to(device))

    # ---- overlap-add of windowed console forwards
    pred_mix = torch.zeros(1, 2, n, dtype=torch.float32, device=device)
    window = torch.hann_window(ANALYSIS_LEN, device=device)
    first = window.clone()
    first[:ANALYSIS_LEN // 2] = 1.0
    dicts = None
    with torch.no_grad():
        for i in range(0, n, ANALYSIS_LEN // 2):
            win_tracks = norm_tracks[..., i:i + ANALYSIS_LEN]
            _, mix_w, *dicts = mix_console(
                win_tracks, pred_track_params, pred_fx_bus_params, pred_master_bus_params,
                use_track_input_fader=True, use_track_panner=True, use_track_eq=True, use_track_compressor=True,
                use_fx_bus=False, use_master_bus=True, use_output_fader=True,
            )
            m = mix_w.shape[-1]  # the last window is shorter: the reference pads it to 262144 before the fade
            pred_mix[..., i:i + m] += mix_w * (first if i == 0 else window)[:m]
    return (pred_mix.to(tracks.device), *dicts)


def load_diffmst(config_path: str, ckpt_path: str, map_location: str = "cpu"):
    """Reference ``mst.utils.load_diffmst`` (mst/utils.py:176-258): build the encoders, controller and console named by a
    training YAML (class paths resolved by import, so ``mst.modules.*`` means whatever ``mst`` is importable - this
    package's alias or a reference checkout after ``diffmst_hip.install()``), split a Lightning checkpoint's ``state_dict``
    by the ``model.track_encoder.`` / ``model.mix_encoder.`` / ``model.controller.`` / ``model.mix_console.`` prefixes and
    return ``(MixStyleTransferModel in eval mode, mix_console)``."""
    import yaml
    from importlib import import_module

    with open(config_path) as f:
        config = yaml.safe_load(f)

    def build(spec):
        module_path, class_name = spec["class_path"].rsplit(".", 1)
        return getattr(import_module(module_path), class_name)(**spec.get("init_args", {}))

    core = config["model"]["init_args"]["model"]
    sub = core["init_args"]
    parts = {name: build(sub[name]) for name in ("track_encoder", "mix_encoder", "controller")}
    mix_console = build(config["model"]["init_args"]["mix_console"])
    checkpoint = torch.load(ckpt_path, map_location=map_location)
    sd = checkpoint["state_dict"]
    for name, module in (*parts.items(), ("mix_console", mix_console)):
        prefix = f"model.{name}."
        module.load_state_dict({k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)})
    module_path, class_name = core["class_path"].rsplit(".", 1)
    model = getattr(import_module(module_path), class_name)(parts["track_encoder"], parts["mix_encoder"], parts["controller"])
    model.eval()
    return model, mix_console
