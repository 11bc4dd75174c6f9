"""A timed-out in-launch exchange must be an ERROR, not a silent NaN (round-4 review item 4, advisor finding on mst_common.h).

The run kernels of the console exchange block / tile aggregates between the workgroups of ONE launch (granules, mst_common.h).  Every
wait is bounded; a wait that gives up poisons its outputs with NaN AND raises MST_STATUS_EXCHANGE_TIMEOUT in the call's status word,
which the binding turns into RuntimeError - the reference's convention for a bad state is a raised exception
(/root/reference/mst/modules.py:86-89, mst/system.py:178-180).  The test loads a developer build of the SAME sources in which no
workgroup ever publishes (-DMST_GRAN_DROP_PUBLISH -DMST_GRAN_SPINS=64, `make -C diff-mst_amd/csrc dev`, built by
__graft_entry__.build()), so every wait gives up deterministically, and checks what the caller sees in both validation modes.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_LIB = os.path.join(ROOT, "diff-mst_amd", "lib", "libdiffmst_hip_dev_nopublish.so")

SCRIPT = r"""
import sys, torch
from mst.modules import AdvancedMixConsole
dev = torch.device("cuda:0")
torch.manual_seed(0)
bs, T, n = 2, 4, 65536
tracks = 0.1 * torch.randn(bs, T, n, device=dev)
def params():
    return (torch.rand(bs, T, 27, device=dev, requires_grad=True), torch.rand(bs, 25, device=dev),
            torch.rand(bs, 26, device=dev, requires_grad=True))
flags = dict(use_track_input_fader=True, use_track_eq=True, use_track_compressor=True, use_track_panner=True,
             use_fx_bus=False, use_master_bus=True, use_output_fader=True)
# validate="sync": the host reads the status word as mirrored right behind the parameter check (ABI v8) - a time-out of a LATER launch of
# the same call is in the sticky device word and raises at the next call at the latest
tp, fp, mp = params()
c = AdvancedMixConsole(44100)
raised = None
for attempt in range(2):
    try:
        c(tracks, tp, fp, mp, **flags)
    except RuntimeError as e:
        raised = (attempt, "timed out" in str(e))
        break
print("SYNC RuntimeError", raised is not None and raised[1], "attempt", None if raised is None else raised[0])
# validate="deferred": the call returns (poisoned), check_parameters() raises; the status word is cleared by the read
c = AdvancedMixConsole(44100, validate="deferred")
tp, fp, mp = params()
mixed, mix, *_ = c(tracks, tp, fp, mp, **flags)
print("DEFERRED finite", bool(torch.isfinite(mix).all()))
try:
    c.check_parameters()
    print("DEFERRED no-raise")
except RuntimeError as e:
    print("DEFERRED RuntimeError", "timed out" in str(e))
c.check_parameters()  # cleared: no second raise
# the backward reports through the same (sticky) word: its own exchanges give up as well; read at the next check
mix.sum().backward()
torch.cuda.synchronize()
try:
    c.check_parameters()
    print("BACKWARD no-raise")
except RuntimeError as e:
    print("BACKWARD RuntimeError", "timed out" in str(e))
"""


@pytest.mark.gpu
def test_timed_out_exchange_raises():
    if not os.path.exists(DEV_LIB):
        pytest.skip("developer library not built (make -C diff-mst_amd/csrc dev)")
    env = dict(os.environ, MST_HIP_LIB=DEV_LIB)
    env["PYTHONPATH"] = os.pathsep.join([ROOT, os.path.join(ROOT, "diff-mst_amd"), os.path.join(ROOT, "diff-mst_amd", "standalone"),
                                         env.get("PYTHONPATH", "")])
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = out.stdout.strip().splitlines()
    assert any(l.startswith("SYNC RuntimeError True") for l in lines), out.stdout
    assert "DEFERRED finite False" in lines, out.stdout  # the poisoned call is visibly poisoned ...
    assert "DEFERRED RuntimeError True" in lines, out.stdout  # ... and the deferred check raises
    assert "BACKWARD RuntimeError True" in lines, out.stdout


def test_status_code_decodes_to_runtime_error():
    from diffmst_hip import _desc

    err = _desc.status_to_error(_desc.EXCHANGE_TIMEOUT)
    assert isinstance(err, RuntimeError) and "timed out" in str(err)
    # range-check codes keep their meaning, and the time-out outranks them in the device's atomic max
    assert isinstance(_desc.status_to_error(1000 - 1 - 0), ValueError)
    assert _desc.EXCHANGE_TIMEOUT > 1000
