"""A short run of the randomised differential test (tools/fuzz.py): shapes, data families, penalties over five decades,
pinned and adaptive geometry modes, every 2-D solver -- against the CPU oracle."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_fuzz_against_oracle(ptv, oracle):
    import fuzz
    cases, worst, where = fuzz.run(budget=12.0, seed=2026, sizes=(2, 3, 17, 95, 96, 97, 130, 257, 400))
    assert cases > 50
    assert worst <= 1e-9, where


def test_fuzz_volumes_against_oracle(ptv, oracle):
    import fuzz
    cases, worst, where = fuzz.run_nd(budget=10.0, seed=2027, sizes=(2, 5, 33, 96, 130))
    assert cases > 10
    assert worst <= 1e-9, where
