"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """A fresh escape-hatch log per session (tests/helpers.py::hatch, tests/test_zz_hatch_budget.py)."""
    from tests.helpers import HATCH_REPORT, MEASURED_REPORT

    for f in (HATCH_REPORT, MEASURED_REPORT):
        try:
            f.unlink()
        except OSError:
            pass
