"""the reference's only caller of its OpenMP container, example/sz_openmp.c, built UNCHANGED against include/ (it includes
"sz_omp.h", round 5) and this library.  The source is read where it lies under /root/reference (this container only: the test skips elsewhere);
nothing of it is copied.  Linked against the CPU shim of the product code it compresses a 3-D array with `-k` (the OpenMP container), and the
stream is the oracle's restatement of sz_omp.c for the same box count, byte for byte, and decompresses within the bound."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CLI = "/root/reference/example/sz_openmp.c"


@pytest.mark.skipif(not os.path.exists(REF_CLI), reason="the reference tree is only in the build container")
def test_reference_openmp_cli_links_unchanged_and_round_trips_on_the_shim(built, tmp_path):
    import sim_lib
    from sz_amd.fields import s_field
    simdir = os.path.dirname(sim_lib.shim_path())
    exe = str(tmp_path / "sz_openmp_ref")
    inc = os.path.join(ROOT, "include")
    # against the product library: must compile against include/sz_omp.h and link ...
    r = subprocess.run(["gcc", "-O1", "-w", "-fopenmp", "-I", inc, "-o", exe + "_hip", REF_CLI, "-L", os.path.join(ROOT, "sz_amd", "csrc"), "-lszhip", "-lm",
                        "-Wl,-rpath," + os.path.join(ROOT, "sz_amd", "csrc")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    # ... and against the same code on the CPU shim, to run it here
    r = subprocess.run(["gcc", "-O1", "-w", "-fopenmp", "-I", inc, "-o", exe, REF_CLI, "-L", simdir, "-lszhip_sim", "-lm", "-Wl,-rpath," + simdir], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    d = s_field(16, 32, 32)                      # two boxes of 8 x 32 x 32 with the box count below
    src = str(tmp_path / "t.dat")
    d.tofile(src)
    cfg = os.path.join(ROOT, "tests", "golden", "sz_speed.config")
    env = dict(os.environ, OMP_NUM_THREADS="2", SZ_HIP_OMP_THREADS="2")
    r = subprocess.run([exe, "-z", "-k", "-f", "-c", cfg, "-i", src, "-M", "ABS", "-A", "1E-4", "-3", "32", "32", "16"], capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert r.returncode == 0 and os.path.exists(src + ".sz"), r.stdout[-600:] + r.stderr[-600:]
    r = subprocess.run([exe, "-x", "-k", "-f", "-s", src + ".sz", "-3", "32", "32", "16", "-i", src, "-a"], capture_output=True, text=True, cwd=str(tmp_path), env=env)
    assert r.returncode == 0 and os.path.exists(src + ".sz.out"), r.stdout[-600:] + r.stderr[-600:]
    back = np.fromfile(src + ".sz.out", dtype=np.float32)
    assert back.size == d.size and float(np.abs(back.astype(np.float64) - d.ravel()).max()) <= 1e-4
