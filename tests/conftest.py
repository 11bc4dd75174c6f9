import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "diff-mst_amd")
for p in (ROOT, PKG, os.path.join(PKG, "standalone")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def pytest_collection_modifyitems(config, items):
    """`slow` CPU cases (host-simulated kernels at the reference's full sizes, minutes each) run with MST_RUN_SLOW=1 only, so that the
    default `-m "not gpu"` suite stays within a few minutes; the same sizes are covered on the device by the -m gpu tests."""
    if os.environ.get("MST_RUN_SLOW") == "1":
        return
    skip = pytest.mark.skip(reason="slow host-simulation case: set MST_RUN_SLOW=1")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


# ---- measured parity numbers of the -m gpu tests -> gpurun_out/parity_r06.json (committed copy: profiles/) -----------
_PARITY = {}


@pytest.fixture
def record(request):
    """record(key=value, ...) files the numbers a parity test measured under the test's id."""

    def rec(**values):
        import re

        key = re.sub(r"\[/.*?/tests/golden/", "[", request.node.nodeid.split("::", 1)[-1])
        d = _PARITY.setdefault(key, {})
        for k, v in values.items():
            d[k] = [float(x) for x in v] if isinstance(v, (tuple, list)) else float(v)

    return rec


def pytest_sessionfinish(session, exitstatus):
    if not _PARITY:
        return
    import json

    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", ROOT), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    legend = ("rel-L2 errors measured by the -m gpu tests on the MI355X; triples are (HIP vs fp32 reference algorithm, "
              "HIP vs float64, fp32 reference algorithm vs float64)")
    with open(os.path.join(out, "parity_r06.json"), "w") as f:
        json.dump({"legend": legend, "tests": _PARITY}, f, indent=1, sort_keys=True)
