"""The reference's OpenMP container (sz/src/sz_omp.c: a 3-D array cut into thread_num independent boxes, one Huffman code book, one payload per
box) -- the ORACLE: oracle/szo_omp_impl.h restates it, this file pins the restatement (thousands of independent boxes instead of one
dependency front, and a stream that a stock OpenMP build of SZ reads: DESIGN section 4h).  The HIP side (sz_amd/csrc/szh_omp.h) is checked
against this oracle in tests/test_zz_omp_hip.py; nothing here touches the product.

Pin: tests/golden/ref_recorded_omp.json -- outputs of the unmodified reference (oracle/_ref/libSZ_omp.so) recorded by
tools/record_reference_omp.py: the bytes behind the parameter block and the decoded array must match, md5 for md5.  Where that library is
present (the build container) it is also run live.  Two of the cases carry fill values (1e30) and NaN / -inf: they pin what the x86-64 build of
the reference does with interval-optimiser quotients beyond the range of `unsigned long`.  float64 (round 4): the stream is pinned by the
same sources built at -O1 (oracle/_ref/libSZ_omp_O1.so; at -O3 the double entry point runs off the end of a function and traps); the reference's
double DECODER cannot read its own streams (it steps over the 4-byte interval count with sizeof(double), sz_omp.c:940-942), so for float64
the decoded array is the restatement's alone, checked against the bound."""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
RECORDED = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_recorded_omp.json")))


def _field(rec):
    import record_reference_omp as R
    return R.make_field(rec["field"], tuple(rec["shape"]), rec["dtype"])


@pytest.mark.parametrize("name", sorted(RECORDED))
def test_oracle_reproduces_recorded_reference_container(oracle, name):
    rec = RECORDED[name]
    d = _field(rec)
    meta = bytes.fromhex(rec["meta_hex"])
    s = oracle.omp_compress(d, rec["eb"], rec["threads"], meta)
    assert len(s) == rec["stream_len"]
    assert s[:len(meta)] == meta
    assert hashlib.md5(s[len(meta):]).hexdigest() == rec["body_md5"]
    dec = oracle.omp_decompress(s, len(meta), d.shape, d.dtype)
    if rec["decoded_md5"] is not None:                       # (float64: the reference has no working decoder to record from)
        assert hashlib.md5(dec.tobytes()).hexdigest() == rec["decoded_md5"]
    ok = np.isfinite(d)                                       # (the fill-value / NaN cases)
    assert float(np.abs(dec.astype(np.float64) - d.astype(np.float64))[ok].max()) <= rec["eb"]


def test_box_grid_and_uneven_shapes_round_trip(oracle):
    """Shapes that do not divide by the box grid, thread counts that are not cubes: the restatement decodes its own streams within the
    bound (the reference itself is not a pin there: it counts Huffman frequencies over uninitialised gaps of its code array)."""
    from sz_amd.fields import s_field
    meta = bytes(32)
    for shape, threads, eb in (((33, 47, 50), 8, 1e-3), ((20, 64, 31), 16, 1e-4), ((16, 16, 16), 2, 1e-4), ((40, 40, 40), 1, 1e-3)):
        d = s_field(*shape)
        s = oracle.omp_compress(d, eb, threads, meta)
        dec = oracle.omp_decompress(s, len(meta), shape, d.dtype)
        assert float(np.abs(dec.astype(np.float64) - d.astype(np.float64)).max()) <= eb, (shape, threads)


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libSZ_omp.so")), reason="the reference build lives in the build container only")
def test_recorded_outputs_are_what_the_reference_library_gives_today():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "record_reference_omp.py"), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-600:] + r.stderr[-600:]
