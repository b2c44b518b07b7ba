"""pytest config: registers the `gpu` marker, puts the repo root on sys.path and
offers helpers to load the committed golden fixtures (tests/golden/*.npz)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name, dtype=None):
    """-> dict group -> dict key -> torch tensor."""
    import torch
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    out = {}
    for k in z.files:
        g, key = k.split("/", 1)
        t = torch.from_numpy(z[k])
        if dtype is not None and t.is_floating_point():
            t = t.to(dtype)
        out.setdefault(g, {})[key] = t
    return out


@pytest.fixture(scope="session")
def golden():
    return load_golden


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Achieved parity errors (worst tensor per test) go into the test log and into gpurun_out/parity_report.json."""
    import parity
    if not parity.RECORDS:
        return
    terminalreporter.section("achieved parity errors (max-abs relative to the tensor's max; tests/parity.py)")
    for line in parity.summary_lines():
        terminalreporter.write_line(line)
    path = parity.write_report(ROOT)
    terminalreporter.write_line(f"{len(parity.RECORDS)} comparisons -> {path}")


@pytest.fixture
def split_wgrads():
    """Tests that compare two schedules of the SAME computation to rounding (1e-5) run the bf16 mode's weight gradients in their
    fp32-grade "split" form: with plain bf16 operands a last-bit difference of an operand can flip its bf16 rounding."""
    import vln_amd
    before = vln_amd.ops.get_wgrad_precision()
    vln_amd.ops.set_wgrad_precision("split")
    yield
    vln_amd.ops.set_wgrad_precision(before)
