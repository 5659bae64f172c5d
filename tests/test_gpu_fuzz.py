"""Randomised parity on the GPU (tests/fuzz_parity.py): genome shape x read length x cost model x noise x tier sizes, the
product through the C ABI against the CPU oracle, SAM line by line and hit-list lengths.  600 such cases were run when
this was written (seeds 2-5, 150 cases each: all identical); the test keeps a 30-case sample in the suite."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


def test_random_cases_identical_to_oracle():
    import fuzz_parity
    assert fuzz_parity.run(30, 11) == 30
