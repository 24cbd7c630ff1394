"""Runs last (file order): the escape hatches of the random sweeps, counted.

Every criterion of tests/test_hip_fuzz.py (and of the sweeps built on it: test_hip_wide.py, test_hip_wide_train.py,
test_hip_bwd_fused.py) that skips a case or relaxes a bar because the REFERENCE is ill-conditioned logs its use through
tests/helpers.py::hatch.  The counts of a full `-m gpu` session at SDEH_FUZZ_SCALE = 1 are recorded in
tests/golden/hatch_budget.json; a session in which any hatch fires more often than recorded (+ its slack) fails here -- a real
defect hiding behind an allowance shows up as a count that grows."""
import json
import os
from collections import Counter
from pathlib import Path

import pytest

from tests.helpers import HATCH_REPORT

BUDGET = Path(__file__).parent / "golden" / "hatch_budget.json"


@pytest.mark.gpu
def test_escape_hatches_stay_within_the_recorded_budget():
    if int(os.environ.get("SDEH_FUZZ_SCALE", "1")) != 1:
        pytest.skip("the budget is recorded for SDEH_FUZZ_SCALE = 1")
    counts = Counter()
    if HATCH_REPORT.exists():
        for line in HATCH_REPORT.read_text().splitlines():
            counts[line.split("\t")[0]] += 1
    summary = HATCH_REPORT.with_name("fuzz_hatch_counts.json")
    try:
        summary.write_text(json.dumps(dict(sorted(counts.items())), indent=1) + "\n")
    except OSError:
        pass
    if not BUDGET.exists():
        pytest.skip(f"no recorded budget ({BUDGET.name}); this session: {dict(counts)}")
    rec = json.loads(BUDGET.read_text())
    over = {k: (v, rec["counts"].get(k, 0)) for k, v in counts.items() if v > rec["counts"].get(k, 0) + rec.get("slack", 0)}
    assert not over, f"escape hatches fired more often than recorded (count, budget): {over}"
