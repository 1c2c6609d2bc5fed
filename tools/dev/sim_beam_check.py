#!/usr/bin/env python3
"""Development check (CPU shim, no GPU): the beam sweep (sz_amd/csrc/szh_beam.h) against the oracle, streams byte for byte and decoded values bit
for bit, over shapes that exercise several tiles along k and j, ragged extents, both types, the mean shortcut and regression blocks.
usage: python tools/dev/sim_beam_check.py [quick]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("SZ_HIP_BEAM", "2")
import oracle_lib as O  # noqa: E402
import sim_lib  # noqa: E402
import sz_amd  # noqa: E402
from sz_amd import api  # noqa: E402
from sz_amd.fields import m_field, s_field  # noqa: E402


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    api._lib = api._bind(ctypes.CDLL(os.path.join(ROOT, "tests", "sim", "libszhip_sim.so")))
    assert sz_amd.SZ_Init(os.path.join(ROOT, "tests", "golden", "sz_speed.config")) == 0
    rng = np.random.default_rng(5)
    cases = [("S-12x8x32", s_field(12, 8, 32), 1e-4), ("S-10x40x36", s_field(10, 40, 36), 1e-4), ("S-30x70x68", s_field(30, 70, 68), 1e-4),
             ("M24", m_field(24), 1e-4), ("M40", m_field(40), 1e-4)]
    if not quick:
        mean_dom = s_field(20, 36, 40).copy(); mean_dom[rng.random(mean_dom.shape) < 0.7] = 0.25
        cases += [("S-64^3", s_field(64, 64, 64), 1e-4), ("N-60x36x40", (s_field(60, 36, 40) + 0.01 * rng.standard_normal((60, 36, 40))).astype(np.float32), 1e-3),
                  ("mean", mean_dom, 1e-4), ("S64-21x70x36", s_field(21, 70, 36, np.float64), 1e-3), ("M36-f64", m_field(36).astype(np.float64), 1e-4),
                  ("S-100x33x132", s_field(100, 33, 132), 1e-4), ("mean-f64", mean_dom.astype(np.float64), 1e-4)]
    bad = 0
    for name, d, eb in cases:
        d = np.ascontiguousarray(d)
        ref, _ = O.compress(d, O.ABS, eb)
        got = sz_amd.SZ_compress_args(d, sz_amd.ABS, eb)
        st = sz_amd.SZ_hip_last_stats()
        okc = got == ref
        dec = sz_amd.SZ_decompress(ref, d.shape, d.dtype)
        want = O.decompress(ref, d.shape, d.dtype)
        okd = np.array_equal(dec.view(np.uint8), want.view(np.uint8))
        print(f"{name:16s} kernel {int(st.quant_kernel)} compress {'ok' if okc else 'DIFF'} ({len(got)} / {len(ref)} B) decompress {'ok' if okd else 'DIFF'}", flush=True)
        bad += (not okc) + (not okd)
    sz_amd.SZ_Finalize()
    print("bad:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
