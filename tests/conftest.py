import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(params=["row4", "wave"])
def both_step_kernels(request, monkeypatch):
    """Runs a GPU parity test twice: with the four-envs-per-wave step kernel wherever it exists (row4: 15x10 and 32x32
    boards with 16 bullet slots; other boards fall back) and with the wave-per-env kernel everywhere (wave).
    Modules opt in with pytest.mark.usefixtures("both_step_kernels")."""
    monkeypatch.setenv("INVERSUS_STEP_KERNEL", "row4_if_any" if request.param == "row4" else "wave")
    return request.param
