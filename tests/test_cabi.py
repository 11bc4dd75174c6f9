"""The C-ABI library: loads without a GPU, exports every symbol include/diffmst_hip.h declares, and its
host-side entry points (sizes / validation) behave.  No compute call is made here."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "diffmst_hip.h")


@pytest.fixture(scope="module")
def lib():
    from mst import _hip

    if not os.path.exists(_hip.LIB_PATH):
        subprocess.run(["make", "-s", "-j8", "-C", os.path.join(ROOT, "diff-mst_amd", "csrc")], check=True)
    return _hip.lib()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mst_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for need in ("mst_console_forward", "mst_console_backward", "mst_mrstft_forward", "mst_mrstft_backward",
                 "mst_afloss_forward", "mst_afloss_backward", "mst_peak_normalize_forward"):
        assert need in syms


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), f"{name} declared in include/diffmst_hip.h but not exported"


def test_binding_covers_the_header(lib):
    from mst import _cabi

    assert sorted(_cabi.SIGNATURES) == declared_symbols()
    assert lib.mst_abi_version() == _cabi.ABI_VERSION


def test_struct_layout_matches_the_header():
    """ctypes mirror of mst_console_desc / mst_mrstft_desc: sizes as the C compiler lays them out."""
    from mst import _cabi

    src = r'''
    #include <stdio.h>
    #include "%s"
    int main(void) { printf("%%zu %%zu %%zu %%zu %%zu\n", sizeof(mst_console_desc), sizeof(mst_mrstft_desc), sizeof(mst_cnn14_desc),
                            sizeof(mst_cnn14_params), sizeof(mst_cnn14_grads)); return 0; }
    ''' % HEADER
    exe = "/tmp/mst_sizeof_test"
    subprocess.run(["gcc", "-x", "c", "-", "-o", exe], input=src.encode(), check=True)
    a, b, c, d, e = (int(v) for v in subprocess.run([exe], capture_output=True, check=True).stdout.split())
    assert ctypes.sizeof(_cabi.ConsoleDesc) == a
    assert ctypes.sizeof(_cabi.MrstftDesc) == b
    assert (ctypes.sizeof(_cabi.Cnn14Desc), ctypes.sizeof(_cabi.Cnn14Params), ctypes.sizeof(_cabi.Cnn14Grads)) == (c, d, e)


def test_workspace_sizes_and_validation(lib):
    from mst import _cabi, _desc
    from mst.modules import AdvancedMixConsole

    ranges = AdvancedMixConsole(44100).param_ranges
    ok = _desc.make_desc(ranges, 44100, 8, 8, 262144, 262144, _desc.flag_word(use_fx_bus=False))
    nbytes = lib.mst_console_workspace_bytes(ctypes.byref(ok))
    assert 100e6 < nbytes < 2e9  # BASELINE cfg #2: a few hundred MB of saved intermediates
    fx = _desc.make_desc(ranges, 44100, 8, 8, 262144, 262144, _desc.flag_word(use_fx_bus=True))
    assert lib.mst_console_workspace_bytes(ctypes.byref(fx)) > nbytes  # fx bus: spectra of the partitioned convolution on top
    bad = _desc.make_desc(ranges, 44100, 8, 8, 262144, 262144, _desc.flag_word(use_fx_bus=True), fx_ir_samples=65000)
    assert lib.mst_console_workspace_bytes(ctypes.byref(bad)) == 0  # impulse response must be whole 4096-sample partitions
    bad = _desc.make_desc(ranges, 44100, 8, 8, 262144, 262144, _desc.flag_word(use_fx_bus=True), fx_bandpass_taps=1024)
    assert lib.mst_console_workspace_bytes(ctypes.byref(bad)) == 0  # odd band-pass length (dasp asserts it)
    nopan = _desc.make_desc(ranges, 44100, 1, 1, 1000, 1000, _desc.flag_word(use_fx_bus=False, use_track_panner=False))
    assert lib.mst_console_workspace_bytes(ctypes.byref(nopan)) == 0
    # launchers refuse bad arguments before touching the device
    assert lib.mst_console_forward(ctypes.byref(fx), None, None, None, None, None, None, None, None, None, 0, None) != 0
    d = _cabi.MrstftDesc()
    d.rows, d.n_samples, d.n_res = 16, 262144, 3
    for i, (nf, hop) in enumerate(((512, 256), (2048, 1024), (8192, 4096))):
        d.fft_size[i], d.hop_size[i], d.win_length[i] = nf, hop, nf
    d.eps = 1e-8
    assert lib.mst_mrstft_tables_bytes(ctypes.byref(d)) == 4 * 3 * (512 + 2048 + 8192)
    assert lib.mst_mrstft_workspace_bytes(ctypes.byref(d)) > 0
    d.fft_size[0] = 1000  # not a power of two
    assert lib.mst_mrstft_workspace_bytes(ctypes.byref(d)) == 0
    assert lib.mst_afloss_workspace_bytes(8, 262144) > 0
    assert lib.mst_afloss_workspace_bytes(8, 16384) == 0  # reflect padding needs n > 16384
    assert lib.mst_peak_normalize_workspace_bytes(8, 262144) > 0
    # spectrogram encoder: 513 x 1025 images (262144 samples), 16 signals - a few GB of bf16 activations; too few frames for six pools -> 0
    e = _cabi.Cnn14Desc(16, 513, 1025, 512, 0, 1, 1e-5, 1)
    assert 2e9 < lib.mst_cnn14_workspace_bytes(ctypes.byref(e)) < 2e10
    e32 = _cabi.Cnn14Desc(16, 513, 1025, 512, 1, 1, 1e-5, 1)
    assert lib.mst_cnn14_workspace_bytes(ctypes.byref(e32)) > lib.mst_cnn14_workspace_bytes(ctypes.byref(e))
    assert lib.mst_cnn14_workspace_bytes(ctypes.byref(_cabi.Cnn14Desc(1, 100, 1025, 512, 0, 1, 1e-5, 1))) == 0
    assert lib.mst_spectrogram_tables_bytes() == 3 * 2048 * 4
