"""TEST INFRASTRUCTURE: ``sys.modules`` stand-ins for the third-party packages the reference imports but this image
lacks (torchaudio, librosa, dasp_pytorch, pytorch_lightning, auraloss, pyloudnorm), so that the REAL reference
modules under /root/reference can be imported in the build container - by tests/golden/make_golden.py to generate
fixtures and by tests/test_install_cpu.py to exercise ``diffmst_hip.install()``.  The dasp stand-in exposes the
oracle's restated ops, i.e. they are patched in exactly at the reference's import seam (mst/modules.py:7-14).
Nothing here travels into the product, and nothing here is reference source."""
import sys
import types

import torch


def install_stubs(dasp_ops=None, auraloss_mrstft=None):
    ta = types.ModuleType("torchaudio")
    ta.pipelines = types.ModuleType("torchaudio.pipelines")
    ta.pipelines.HDEMUCS_HIGH_MUSDB_PLUS = None
    ta.transforms = types.ModuleType("torchaudio.transforms")
    ta.functional = types.ModuleType("torchaudio.functional")
    for name in ("torchaudio", "torchaudio.pipelines", "torchaudio.transforms", "torchaudio.functional"):
        sys.modules[name] = {"torchaudio": ta, "torchaudio.pipelines": ta.pipelines, "torchaudio.transforms": ta.transforms,
                             "torchaudio.functional": ta.functional}[name]
    sys.modules["librosa"] = types.ModuleType("librosa")

    dp = types.ModuleType("dasp_pytorch")
    dpf = types.ModuleType("dasp_pytorch.functional")
    if dasp_ops is None:
        from oracle import dasp_restated as dasp_ops
    for name in ("gain", "stereo_panner", "compressor", "parametric_eq", "stereo_bus", "noise_shaped_reverberation"):
        setattr(dpf, name, getattr(dasp_ops, name))
    dp.functional = dpf
    sys.modules["dasp_pytorch"] = dp
    sys.modules["dasp_pytorch.functional"] = dpf

    # pytorch_lightning: only what `class System(pl.LightningModule)` touches at construction / in common_step
    pl = types.ModuleType("pytorch_lightning")

    class LightningModule(torch.nn.Module):
        current_epoch = 0

        def save_hyperparameters(self, *a, **k):
            self.hparams = types.SimpleNamespace()

        def log(self, *a, **k):
            self.__dict__.setdefault("logged", []).append((a, k))

    pl.LightningModule = LightningModule
    pl.LightningDataModule = object
    pl.Callback = object
    sys.modules["pytorch_lightning"] = pl
    cb = types.ModuleType("pytorch_lightning.callbacks")
    cb.Callback = object
    sys.modules["pytorch_lightning.callbacks"] = cb
    pl.callbacks = cb

    au = types.ModuleType("auraloss")
    au.time = types.ModuleType("auraloss.time")
    au.freq = types.ModuleType("auraloss.freq")

    class SISDRLoss(torch.nn.Module):
        pass

    class MultiResolutionSTFTLoss(torch.nn.Module):  # identity marker of "the original"; the oracle's restatement if given
        def __init__(self, **kw):
            super().__init__()
            self.kw = kw

        def forward(self, x, y):
            if auraloss_mrstft is None:
                raise RuntimeError("auraloss stand-in called")
            return auraloss_mrstft(x, y, **self.kw)

    au.time.SISDRLoss = SISDRLoss
    au.freq.MultiResolutionSTFTLoss = MultiResolutionSTFTLoss
    sys.modules["auraloss"] = au
    sys.modules["auraloss.time"] = au.time
    sys.modules["auraloss.freq"] = au.freq

    pyln = types.ModuleType("pyloudnorm")

    class Meter:
        def __init__(self, rate):
            self.rate = rate

    pyln.Meter = Meter
    sys.modules["pyloudnorm"] = pyln
    return dict(torchaudio=ta, dasp=dpf, pl=pl, auraloss=au, pyloudnorm=pyln)
