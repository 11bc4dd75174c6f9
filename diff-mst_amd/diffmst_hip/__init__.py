"""MI355X-native implementation of the Diff-MST mix-console hot path (package ``diffmst_hip``).

The hot-path classes and functions of the reference, under their reference names:

    diffmst_hip.modules.AdvancedMixConsole          <- reference mst/modules.py:100-487
    diffmst_hip.mixing.naive_random_mix             <- reference mst/mixing.py:35-94
    diffmst_hip.loss.AudioFeatureLoss               <- reference mst/loss.py:198-260
    diffmst_hip.loss.MultiResolutionSTFTLoss        <- auraloss.freq.MultiResolutionSTFTLoss (configs/models/naive.yaml:55)
    diffmst_hip.utils.batch_stereo_peak_normalize   <- reference mst/utils.py:14-29
    diffmst_hip.system.CommonStep                   <- call order of reference mst/system.py:102-407, without Lightning

All numerical work runs in hand-written HIP kernels for gfx950 behind the C ABI of
``include/diffmst_hip.h``; there is no CPU fallback.

``install()`` rebinds exactly those symbols inside an importable checkout of the reference, leaving the
rest of its ``mst`` package (``System``, the controller model, the data modules, ...) untouched - see
INTEGRATION.md; ``install(models=True)`` also swaps the parameter-estimation model classes (encoders on the
matrix cores, controller stack on HIP).  No reference source travels with this package.
"""
from __future__ import annotations

import importlib
import sys

__version__ = "0.2.0"

from . import _cabi, _desc, _hip, filter, loss, mixing, modules, panns, utils  # noqa: E402,F401
from . import system  # noqa: E402,F401

# (module of the reference, attribute, replacement)
_TARGETS = (
    ("mst.modules", "AdvancedMixConsole", modules.AdvancedMixConsole),
    ("mst.mixing", "naive_random_mix", mixing.naive_random_mix),
    ("mst.loss", "AudioFeatureLoss", loss.AudioFeatureLoss),
    ("mst.utils", "batch_stereo_peak_normalize", utils.batch_stereo_peak_normalize),
    ("auraloss.freq", "MultiResolutionSTFTLoss", loss.MultiResolutionSTFTLoss),
)
# the parameter-estimation model: opt-in, install(models=True) - state-dict compatible with the reference's classes
_MODEL_TARGETS = (
    ("mst.modules", "MixStyleTransferModel", modules.MixStyleTransferModel),
    ("mst.modules", "SpectrogramEncoder", modules.SpectrogramEncoder),
    ("mst.modules", "TransformerController", modules.TransformerController),
    ("mst.panns", "Cnn14", panns.Cnn14),
)
_installed = {}


def _peak_normalize_dispatch(reference_fn):
    """``System.common_step`` also peak-normalises a MONO, already-on-the-host plotting copy
    (reference mst/system.py:390-391).  Device tensors go to the HIP kernel; host tensors stay with the reference's own
    function (it is the reference's code that runs, nothing is restated here)."""

    def batch_stereo_peak_normalize(x):
        if x.is_cuda:
            return utils.batch_stereo_peak_normalize(x)
        return reference_fn(x)

    batch_stereo_peak_normalize.__wrapped__ = reference_fn
    return batch_stereo_peak_normalize


def install(strict: bool = True, models: bool = False) -> dict:
    """Swap the five hot-path symbols of the reference for the HIP implementations.

    ``models=True`` also rebinds the parameter-estimation model - ``mst.modules.{MixStyleTransferModel, SpectrogramEncoder,
    TransformerController}`` and ``mst.panns.Cnn14`` (reference mst/modules.py:17-68, :740-914, mst/panns.py:126-209): same constructor
    keywords, same parameter names (reference checkpoints load), so an unchanged ``System`` + unchanged YAML runs the STFT front end and
    Cnn14 on the matrix cores (fp32 operands unless ``MST_ENCODER_PRECISION=bf16``) and the controller's encoder stack on
    ``csrc/mst_ctrl.hip`` whenever its shape is inside the kernels' limits.

    Requires the reference's ``mst`` package (and ``auraloss``) to be importable as usual; every other name
    of the reference (``mst.system.System``, ``mst.modules.MixStyleTransferModel``,
    ``mst.mixing.knowledge_engineering_mix``, ...) stays the reference's own.  Modules that already did
    ``from mst.utils import batch_stereo_peak_normalize`` (e.g. ``mst.system``) are patched as well, and the
    class paths in the reference's YAML configs (``mst.modules.AdvancedMixConsole``,
    ``auraloss.freq.MultiResolutionSTFTLoss``) resolve to the replacements from then on.

    Returns ``{(module, attribute): original object}``; ``uninstall()`` restores them.
    ``strict=False`` skips targets whose module cannot be imported instead of raising.
    """
    replaced, current = {}, {}
    for modname, attr, new in (_TARGETS + _MODEL_TARGETS if models else _TARGETS):
        try:
            mod = importlib.import_module(modname)
        except ImportError:
            if strict:
                raise
            continue
        if getattr(sys.modules.get(modname.split(".")[0]), "__diffmst_alias__", False):
            raise RuntimeError(
                f"`{modname}` resolves to diffmst_hip's own alias package, not to a checkout of the reference: put the "
                "reference on sys.path (and drop diff-mst_amd/standalone from it, or import the reference first)"
            )
        old = getattr(mod, attr, None)
        if old is new or (modname, attr) in _installed:
            continue
        if attr == "batch_stereo_peak_normalize" and old is not None:
            new = _peak_normalize_dispatch(old)
        setattr(mod, attr, new)
        replaced[(modname, attr)] = old
        current[(modname, attr)] = new
    # names that other modules of the reference bound at import time (`from mst.utils import ...`)
    for name, m in list(sys.modules.items()):
        if m is None or not (name == "mst" or name.startswith("mst.") or name == "auraloss" or name.startswith("auraloss.")):
            continue
        for (modname, attr), old in replaced.items():
            if old is not None and name != modname and getattr(m, attr, None) is old:
                setattr(m, attr, current[(modname, attr)])
                _installed[(name, attr)] = old
    _installed.update(replaced)
    return replaced


def uninstall() -> None:
    """Undo ``install()``."""
    for (modname, attr), old in list(_installed.items()):
        m = sys.modules.get(modname)
        if m is not None and old is not None:
            setattr(m, attr, old)
        _installed.pop((modname, attr))
