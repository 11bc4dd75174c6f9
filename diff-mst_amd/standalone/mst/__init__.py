"""Alias package: ``mst.modules`` / ``mst.mixing`` / ``mst.loss`` / ``mst.utils`` / ``mst.filter`` of the MI355X build.

The implementation lives in ``diffmst_hip`` (``diff-mst_amd/diffmst_hip``).  This alias lives in its own directory
(``diff-mst_amd/standalone``) because the reference's ``mst`` is a namespace package (it has no ``__init__.py``) and any
regular package of that name on ``sys.path`` would shadow ALL of it.  It exists so that stand-alone users of the
hot path (the tests, ``bench.py``, scripts that only need the console and the losses) can keep the reference's
import lines - ``from mst.modules import AdvancedMixConsole`` - on a machine that has NO checkout of the
reference.  It does NOT shadow the rest of the reference: to run the reference's own ``mst.system.System`` on
the HIP console use ``diffmst_hip.install()`` with the reference's ``mst`` on ``sys.path`` instead
(INTEGRATION.md section 1).
"""
import os
import sys

try:
    import diffmst_hip
except ImportError:  # this directory alone was put on sys.path: the implementation package sits two levels up
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import diffmst_hip
from diffmst_hip import _cabi, _desc, _hip, filter, loss, mixing, modules, panns, system, utils  # noqa: F401

__diffmst_alias__ = True
__version__ = diffmst_hip.__version__
for _name in ("_cabi", "_desc", "_hip", "filter", "loss", "mixing", "modules", "panns", "system", "utils"):
    sys.modules[__name__ + "." + _name] = getattr(diffmst_hip, _name)
del _name
