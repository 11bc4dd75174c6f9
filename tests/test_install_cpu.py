"""``diffmst_hip.install()`` against the REAL reference package (build container only: needs /root/reference).

The reference's own ``mst.system.System`` is imported through stand-ins for its absent third-party imports
(tests/refstubs.py), the five hot-path symbols are swapped, and ``System.common_step`` is driven up to the first
device call - which must be OUR console refusing a CPU tensor, i.e. the rebinding reached the call sites that
``System`` really uses.  Runs in a subprocess: the reference's ``mst`` namespace package must not meet the
stand-alone alias package that the rest of the test-suite imports.
"""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

SCRIPT = textwrap.dedent(
    """
    import sys
    sys.dont_write_bytecode = True
    sys.path[:0] = [{root!r}, {root!r} + "/tests"]
    import refstubs
    refstubs.install_stubs()
    sys.path.insert(0, {ref!r})                       # the reference checkout, as a user has it
    sys.path.append({root!r} + "/diff-mst_amd")       # diffmst_hip only - NOT diff-mst_amd/standalone
    import torch
    import mst.system, mst.modules, mst.mixing, mst.loss, mst.utils      # the reference's own modules
    import auraloss
    assert set(mst.__path__) == {{{ref!r} + "/mst"}}, list(mst.__path__)
    ref_console_cls, ref_peak = mst.modules.AdvancedMixConsole, mst.utils.batch_stereo_peak_normalize
    import diffmst_hip
    swapped = diffmst_hip.install()
    assert len(swapped) == 5, swapped
    # the five hot-path symbols are ours ...
    assert mst.modules.AdvancedMixConsole is diffmst_hip.modules.AdvancedMixConsole
    assert mst.mixing.naive_random_mix is diffmst_hip.mixing.naive_random_mix
    assert mst.loss.AudioFeatureLoss is diffmst_hip.loss.AudioFeatureLoss
    assert auraloss.freq.MultiResolutionSTFTLoss is diffmst_hip.loss.MultiResolutionSTFTLoss
    assert mst.system.batch_stereo_peak_normalize is mst.utils.batch_stereo_peak_normalize   # `from mst.utils import` copy too
    assert mst.utils.batch_stereo_peak_normalize.__wrapped__ is ref_peak
    # ... and everything else is still the reference's
    for mod, name in ((mst.modules, "MixStyleTransferModel"), (mst.modules, "SpectrogramEncoder"), (mst.modules, "TransformerController"),
                      (mst.mixing, "knowledge_engineering_mix"), (mst.system, "System"), (mst.loss, "compute_barkspectrum")):
        assert getattr(mod, name).__module__.startswith("mst."), (name, getattr(mod, name).__module__)
    # host tensors of the plotting path stay with the reference's function (mst/system.py:390-391 passes a mono CPU tensor)
    x = torch.randn(2, 1, 64)
    assert torch.equal(mst.utils.batch_stereo_peak_normalize(x), ref_peak(x))

    # the reference's System, constructed the way its YAML does (class paths resolved by attribute lookup)
    class Model(torch.nn.Module):
        def forward(self, tracks, ref_mix, track_padding_mask=None):
            bs, T, _ = tracks.shape
            return torch.rand(bs, T, 27), torch.rand(bs, 25), torch.rand(bs, 26)
    console = mst.modules.AdvancedMixConsole(sample_rate=44100)
    assert type(console) is not ref_console_cls
    system = mst.system.System(model=Model(), mix_console=console, mix_fn=mst.mixing.naive_random_mix,
                               loss=auraloss.freq.MultiResolutionSTFTLoss(fft_sizes=[512], hop_sizes=[256], win_lengths=[512]),
                               generate_mix=True, active_eq_epoch=0, active_compressor_epoch=0, active_fx_bus_epoch=1000,
                               active_master_bus_epoch=0)
    batch = (0.1 * torch.randn(2, 3, 8192), None, None, torch.zeros(2, 3, dtype=torch.bool), None, ["a", "b"])
    try:
        system.common_step(batch, 0, train=True)
    except RuntimeError as e:       # our console, reached through System's own call at mst/system.py:159
        assert "no CPU path" in str(e), e
        print("REACHED_DEVICE_CALL")
    diffmst_hip.uninstall()
    assert mst.modules.AdvancedMixConsole is ref_console_cls and mst.system.batch_stereo_peak_normalize is ref_peak

    # install(models=True): the parameter-estimation model too (reference mst/modules.py:17-68, :740-914, mst/panns.py:126-209)
    import mst.panns
    ref_model_classes = (mst.modules.MixStyleTransferModel, mst.modules.SpectrogramEncoder, mst.modules.TransformerController, mst.panns.Cnn14)
    ref_sd_keys = set(mst.modules.SpectrogramEncoder(embed_dim=16).state_dict().keys())
    swapped = diffmst_hip.install(models=True)
    assert len(swapped) == 9, swapped
    assert mst.modules.SpectrogramEncoder is diffmst_hip.modules.SpectrogramEncoder and mst.panns.Cnn14 is diffmst_hip.panns.Cnn14
    assert mst.modules.TransformerController is diffmst_hip.modules.TransformerController
    assert mst.modules.MixStyleTransferModel is diffmst_hip.modules.MixStyleTransferModel
    # built the way configs/models/naive+feat.yaml builds it, class paths resolved through the patched modules
    enc = lambda: mst.modules.SpectrogramEncoder(embed_dim=128, n_inputs=1, n_fft=2048, hop_length=512, input_batchnorm=False, encoder_batchnorm=True)
    model = mst.modules.MixStyleTransferModel(enc(), enc(), mst.modules.TransformerController(
        embed_dim=128, num_track_control_params=27, num_fx_bus_control_params=25, num_master_bus_control_params=26, num_layers=2, nhead=8))
    assert set(model.track_encoder.state_dict().keys()) == ref_sd_keys  # reference checkpoints load
    assert model.track_encoder.model.precision == "fp32" and model.controller.native is None
    system = mst.system.System(model=model, mix_console=mst.modules.AdvancedMixConsole(sample_rate=44100), mix_fn=mst.mixing.naive_random_mix,
                               loss=auraloss.freq.MultiResolutionSTFTLoss(fft_sizes=[512], hop_sizes=[256], win_lengths=[512]),
                               generate_mix=False, active_eq_epoch=0, active_compressor_epoch=0, active_fx_bus_epoch=1000,
                               active_master_bus_epoch=0)
    batch = (0.1 * torch.randn(2, 3, 131072), None, None, torch.zeros(2, 3, dtype=torch.bool), 0.1 * torch.randn(2, 2, 131072), ["a", "b"])
    try:
        system.common_step(batch, 0, train=True)
    except RuntimeError as e:       # OUR SpectrogramEncoder, reached through System's own model call (mst/system.py:263-271)
        assert "no CPU path" in str(e), e
        print("REACHED_ENCODER_CALL")
    diffmst_hip.uninstall()
    assert (mst.modules.MixStyleTransferModel, mst.modules.SpectrogramEncoder, mst.modules.TransformerController, mst.panns.Cnn14) == ref_model_classes
    print("INSTALL_OK")
    """
)


@pytest.mark.skipif(not os.path.isdir(REF), reason="needs the reference checkout (build container only)")
def test_install_rebinds_only_the_hot_path_of_the_real_reference():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, "-B", "-c", SCRIPT.format(root=ROOT, ref=REF)], capture_output=True, text=True, env=env,
                       cwd="/tmp", timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "REACHED_DEVICE_CALL" in r.stdout and "REACHED_ENCODER_CALL" in r.stdout and "INSTALL_OK" in r.stdout, r.stdout + r.stderr


def test_install_refuses_the_alias_package():
    """With only the stand-alone alias importable as `mst`, install() must say so instead of patching itself."""
    code = textwrap.dedent(
        f"""
        import sys
        sys.path[:0] = [{ROOT!r} + "/diff-mst_amd", {ROOT!r} + "/diff-mst_amd/standalone"]
        import diffmst_hip
        try:
            diffmst_hip.install()
        except RuntimeError as e:
            assert "alias package" in str(e)
            print("REFUSED")
        """
    )
    r = subprocess.run([sys.executable, "-B", "-c", code], capture_output=True, text=True, cwd="/tmp", timeout=600)
    assert r.returncode == 0 and "REFUSED" in r.stdout, r.stdout + r.stderr
