"""Round 6: several lone contexts of one process compressing arrays with regression blocks at the same time (the coefficient chains of one call beside the sweeps of
the others).  Every stream must be the single call's, no call may fail.  The long form of this test -- 20 000 calls -- is tools/gpu_r6_multictx.py
(profiles/r06_multi_context_stress.txt)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.gpu
@pytest.mark.parametrize("threads, calls", [(2, 40), (4, 20)])
def test_lone_contexts_on_threads_give_the_single_call_stream(threads, calls):
    import gpu_r6_multictx
    r = gpu_r6_multictx.run(threads, calls, 512)
    assert r["errors"] == 0 and r["mismatches"] == 0, r
