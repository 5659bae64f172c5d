"""Randomised parity on the GPU (tests/fuzz_parity.py): genome shape x read length x cost model x noise x tier sizes, the
product through the C ABI against the CPU oracle, SAM line by line and hit-list lengths.  1,960 such cases were run in round 1
(seeds 1-5, 21, 101-108: all identical, including ragged lengths after reads of one cost class began to share a launch);
the test keeps a 30-case sample in the suite."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_random_cases_identical_to_oracle(monkeypatch):
    import fuzz_parity
    assert fuzz_parity.run(20, 11) == 20
    # ... and with the effort-ordered hand-out (ps_effort.hip; launches below 4,096 reads skip it by default) switched on for the
    # few-thousand-read launches of the sweep: the order of the queue must never show in a result
    # (with the order comes the estimated best score that spares the search entries in profile mode, ps_narrow.h: nt_tail)
    monkeypatch.setenv("PS_ORDER_MIN", "1")
    assert fuzz_parity.run(20, 12) == 20
    # ... and with every estimate 8 units too low: most reads start over inside the launch (nt_restart_without_estimate)
    monkeypatch.setenv("PS_CAP_BIAS", "8")
    assert fuzz_parity.run(10, 13) == 10
