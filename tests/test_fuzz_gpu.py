"""A short randomised soak inside the GPU suite: tools/fuzz_parity.py's scenarios (random map sizes, poses, clouds, batches, moves,
lowest tracking + ray tracing, pipeline knobs on top of the suite's pipeline variant), every step bit for bit against the oracle."""
import sys
from pathlib import Path

import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))


@pytest.mark.gpu
@pytest.mark.parametrize("first_seed", [7_000_000, 7_000_100])
def test_random_scenarios_match_the_oracle(oracle_mod, first_seed):
    import fuzz_parity
    points = 0
    for seed in range(first_seed, first_seed + 12):
        points += fuzz_parity.scenario(seed)[4]              # raises AssertionError with the seed on a mismatch
    assert points > 0
