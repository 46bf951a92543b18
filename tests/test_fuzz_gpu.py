"""GPU: a short seeded random sweep of the HIP path against the CPU oracle (tools/fuzz_parity.py):
shapes around the 64 / 128 / 1024 boundaries, d up to 32, k up to 64, both kernels, ARD, panel
widths, device groups in both solve modes, mean-only predict, the analytic gradient.  The long
sweeps (3 seeds, 580 cases: worst mean error 2.7e-10) are recorded in profiles/r02_fuzz_parity.json."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [11, 12])
def test_random_shapes_match_the_oracle(seed):
    import fuzz_parity
    res = fuzz_parity.sweep(30, seed, verbose=False)
    assert not res["failed"], res["failed"]
    assert res["worst_relative_errors"]["mean"] <= 1e-8 and res["worst_relative_errors"]["var"] <= 1e-10
