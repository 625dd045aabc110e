"""bench.py's bookkeeping (no GPU): the algorithmic-bytes yardstick of SURVEY.md §8(d) and the LDS-atomic floor that
DESIGN.md §4 reports next to it."""
import pytest

import bench


def test_algorithmic_bytes_per_visibility():
    # 40 B stream + per tap 16 B kernel read + 16 B grid read + 16 B grid write
    assert bench.alg_bytes_per_vis(15) == 10840
    assert bench.alg_bytes_per_vis(7) == 2392
    assert bench.alg_bytes_per_vis(1) == 88


@pytest.mark.parametrize("S,cycles", [
    (8, 16),        # 64 taps: one full step, two 8-cycle instructions
    (16, 64),       # 256 taps: four full steps
    (15, 56.25),    # 225 taps: three full steps, a 32-tap step shared by two visibilities, 1 tap per block of 64
    (7, 16),        # 49 taps: four 16-lane groups -> 8 cycles per instruction
    (5, 12),        # 25 taps: two groups -> 6 cycles
    (11, 32),       # 121 taps: 64 + 57
    (13, 46),       # 169 taps: 2 x 64 + 41 (three groups: 7 cycles)
])
def test_lds_atomic_cycles_per_visibility(S, cycles):
    assert bench.lds_atomic_cycles_per_vis(S) == cycles


def test_workloads_are_the_baseline_configs():
    assert bench.WORKLOADS["cfg3"] == (100_000_000, 4096, 128, 8, 15)
    assert bench.WORKLOADS["cfg2"] == (1_000_000, 2048, 16, 8, 7)
