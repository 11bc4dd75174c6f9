"""Test harness: runs the C ABI of the kernel sources built against the host-side HIP simulator.

TEST INFRASTRUCTURE ONLY - the product package never loads this library.
"""
import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "build", "libdiffmst_hostsim.so")


def build():
    subprocess.run(["make", "-s", "-j8", "-C", HERE], check=True)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        from mst import _cabi

        build()
        _lib = _cabi.bind(C.CDLL(LIB))
    return _lib


def console(param_ranges, tracks, tp, fp, mp, flags, grad_mix=None, want_mixed=True, grad_mixed=None,
            want_grad_tracks=False, sample_rate=44100, multipass_eq=False, denormalized=False, fx_noise=None,
            fx_ir_samples=65536, fx_bandpass_taps=1023):
    """CPU tensors in; returns dict(mix, mixed, status, grad_tp, grad_mp, grad_tracks)."""
    from mst import _cabi, _desc

    L = lib()
    bs, T, n = tracks.shape
    tracks = tracks.contiguous().float()
    word = _desc.flag_word(save_for_backward=grad_mix is not None, **flags)
    if multipass_eq:
        word |= _cabi.DEV_MULTIPASS_EQ
    if denormalized:  # tp / mp hold denormalised values (forward_mix_console): identity ranges, no range check
        word |= _cabi.NO_RANGE_CHECK
    d = _desc.make_desc(param_ranges, sample_rate, bs, T, n, tracks.stride(1), word, identity_ranges=denormalized,
                        fx_ir_samples=fx_ir_samples, fx_bandpass_taps=fx_bandpass_taps)
    nbytes = L.mst_console_workspace_bytes(C.byref(d))
    assert nbytes > 0
    ws = torch.zeros(nbytes // 4 + 64, dtype=torch.float32)
    off = (-ws.data_ptr() % 256) // 4
    ws = ws[off:]
    mix = torch.zeros(bs, 2, n)
    mixed = torch.zeros(bs, 2, T, n) if want_mixed else None
    status = torch.zeros(1, dtype=torch.int32)
    tp, fp, mp = tp.contiguous().float(), fp.contiguous().float(), mp.contiguous().float()
    fx = None
    if flags.get("use_fx_bus", True):
        from mst.filter import octave_band_filterbank

        noise = fx_noise.contiguous().float()
        assert tuple(noise.shape) == (bs * 2, 12, fx_ir_samples + fx_bandpass_taps - 1)
        filters = octave_band_filterbank(fx_bandpass_taps, sample_rate).contiguous()
        tables = torch.zeros(L.mst_console_fx_tables_bytes() // 4)
        assert L.mst_console_fx_init_tables(_cabi.ptr(tables), None) == 0
        fx = _cabi.ConsoleFx(noise.data_ptr(), filters.data_ptr(), tables.data_ptr())
    fxp = C.byref(fx) if fx is not None else None
    rc = L.mst_console_forward(C.byref(d), _cabi.ptr(tracks), _cabi.ptr(tp), _cabi.ptr(fp), _cabi.ptr(mp), fxp, _cabi.ptr(mix),
                               _cabi.ptr(mixed), _cabi.ptr(status), _cabi.ptr(ws), nbytes, None)
    assert rc == 0, rc
    out = dict(mix=mix, mixed=mixed, status=int(status.item()))
    if grad_mix is not None:
        gtp = torch.full((bs, T, 27), float("nan"))
        gmp = torch.full((bs, 26), float("nan"))
        gtr = torch.zeros(bs, T, n) if want_grad_tracks else None
        gm = grad_mix.contiguous().float()
        gmx = None if grad_mixed is None else grad_mixed.contiguous().float()
        gfp = torch.full((bs, 25), float("nan")) if fx is not None else None
        rc = L.mst_console_backward(C.byref(d), _cabi.ptr(tracks), _cabi.ptr(tp), _cabi.ptr(fp), _cabi.ptr(mp), fxp, _cabi.ptr(gm),
                                    _cabi.ptr(gmx), _cabi.ptr(gtp), _cabi.ptr(gfp), _cabi.ptr(gmp), _cabi.ptr(gtr), _cabi.ptr(status),
                                    _cabi.ptr(ws), nbytes, None)
        assert rc == 0, rc
        out.update(grad_tp=gtp, grad_mp=gmp, grad_tracks=gtr, grad_fp=gfp, status=int(status.item()))
    return out


def mrstft(pred, target, resolutions, w_sc=1.0, w_log_mag=1.0, w_lin_mag=0.0, sc_per_example=True, grad=True):
    """(bs, chs, n) CPU tensors -> dict(loss, grad_pred)."""
    from mst import _cabi

    L = lib()
    x = pred.reshape(-1, pred.shape[-1]).contiguous().float()
    y = target.reshape(-1, target.shape[-1]).contiguous().float()
    d = _cabi.MrstftDesc()
    d.rows, d.n_samples, d.n_res = x.shape[0], x.shape[1], len(resolutions)
    for i, (nf, hop, win) in enumerate(resolutions):
        d.fft_size[i], d.hop_size[i], d.win_length[i] = nf, hop, win
    d.w_sc, d.w_log_mag, d.w_lin_mag = w_sc, w_log_mag, w_lin_mag
    d.sc_per_example, d.eps = int(sc_per_example), 1e-8
    tb, wb = L.mst_mrstft_tables_bytes(C.byref(d)), L.mst_mrstft_workspace_bytes(C.byref(d))
    assert tb > 0 and wb > 0
    tables = torch.zeros(tb // 4)
    ws = torch.zeros(wb // 4)
    assert L.mst_mrstft_init_tables(C.byref(d), _cabi.ptr(tables), None) == 0
    loss = torch.zeros(1)
    assert L.mst_mrstft_forward(C.byref(d), _cabi.ptr(x), _cabi.ptr(y), _cabi.ptr(tables), _cabi.ptr(loss), _cabi.ptr(ws), wb, None) == 0
    out = dict(loss=loss.clone())
    if grad:
        gl = torch.ones(1)
        gx = torch.full_like(x, float("nan"))
        assert L.mst_mrstft_backward(C.byref(d), _cabi.ptr(x), _cabi.ptr(y), _cabi.ptr(tables), _cabi.ptr(gl), _cabi.ptr(gx),
                                     _cabi.ptr(ws), wb, None) == 0
        out["grad_pred"] = gx.view_as(pred)
    return out


def afloss(pred, target, weights, sample_rate=44100, grad=True, grad_losses=None):
    """(bs, 2, n) CPU tensors -> dict(losses (5,), grad_pred)."""
    from mst import _cabi
    from mst.filter import barkscale_fbanks

    L = lib()
    x, y = pred.contiguous().float(), target.contiguous().float()
    bs, _, n = x.shape
    tables = torch.zeros(L.mst_afloss_tables_bytes() // 4)
    assert L.mst_afloss_init_tables(_cabi.ptr(tables), None) == 0
    fb = barkscale_fbanks(16385, 20.0, 20000.0, 24, sample_rate).contiguous()
    wb = L.mst_afloss_workspace_bytes(bs, n)
    assert wb > 0
    ws = torch.zeros(wb // 4)
    w = (C.c_float * 5)(*weights)
    losses = torch.zeros(5)
    assert L.mst_afloss_forward(_cabi.ptr(x), _cabi.ptr(y), bs, n, w, _cabi.ptr(tables), _cabi.ptr(fb), _cabi.ptr(losses),
                                _cabi.ptr(ws), wb, None) == 0
    out = dict(losses=losses.clone())
    if grad:
        g = torch.ones(5) if grad_losses is None else grad_losses.float()
        gx = torch.full_like(x, float("nan"))
        assert L.mst_afloss_backward(_cabi.ptr(x), _cabi.ptr(y), bs, n, w, _cabi.ptr(tables), _cabi.ptr(fb), _cabi.ptr(g),
                                     _cabi.ptr(gx), _cabi.ptr(ws), wb, None) == 0
        out["grad_pred"] = gx
    return out
