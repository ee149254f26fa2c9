"""One seed each of the randomized soaks under tools_dev/ inside the driver's GPU run: the persistent pair kernel's
hand-counted vmcnt bookkeeping, hand-issued loads / stores and tile composition are guarded by row blocks at arbitrary
(also non-granule-aligned) offsets x kernels x forms x tile orders x granule orders, each compared bit for bit with a
reference matrix of the one-tile kernel in bin form; the fused k-nearest consumer likewise against matrix rows."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools_dev"))


def test_pair_kernel_soak_one_seed(engine):
    import fuzz_pair
    blocks, bad = fuzz_pair.soak(engine, seed=20261004, budget_s=100.0, max_blocks=200, n_max=2600)
    assert bad is None, bad
    assert blocks >= 60          # (a slow box may not reach all 200 within the time cap)


def test_topk_soak_one_seed(engine):
    import fuzz_topk
    calls, bad = fuzz_topk.soak(engine, seed=20261004, budget_s=60.0, max_calls=50, n_max=2600)
    assert bad is None, bad
    assert calls >= 20
