"""SURVEY 8(b) "who calls it": the reference's own command-line tool, example/sz.c, built UNCHANGED against include/ and this library
(it needs eight helpers next to the API: SZ_printMetadata, is_lossless_compressed_data, sz_lossless_decompress65536bytes, detransposeData,
checkFileExistance, writeFloatData, writeDoubleData, writeStrings -- round 4).  The source is read where it lies under /root/reference (this
container only: the test skips elsewhere); nothing of it is copied.  Linked against the CPU shim of the product code it compresses the
reference's sample file, prints its metadata and decompresses it within the bound."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CLI = "/root/reference/example/sz.c"


@pytest.mark.skipif(not os.path.exists(REF_CLI), reason="the reference tree is only in the build container")
def test_reference_cli_links_unchanged_and_round_trips_on_the_shim(built, tmp_path):
    import sim_lib
    simdir = os.path.dirname(sim_lib.shim_path())
    exe = str(tmp_path / "sz_ref_cli")
    # against the product library: must link (every symbol the tool uses is exported) ...
    r = subprocess.run(["gcc", "-O1", "-w", "-I", os.path.join(ROOT, "include"), "-o", exe + "_hip", REF_CLI, "-L", os.path.join(ROOT, "sz_amd", "csrc"), "-lszhip", "-lm",
                        "-Wl,-rpath," + os.path.join(ROOT, "sz_amd", "csrc")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    # ... and against the same code on the CPU shim, to run it here
    r = subprocess.run(["gcc", "-O1", "-w", "-I", os.path.join(ROOT, "include"), "-o", exe, REF_CLI, "-L", simdir, "-lszhip_sim", "-lm", "-Wl,-rpath," + simdir],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-800:]
    data = np.fromfile(os.path.join(ROOT, "tests", "golden", "testfloat_8_8_128.dat"), dtype=np.float32)
    src = str(tmp_path / "t.dat")
    data.tofile(src)
    cfg = os.path.join(ROOT, "tests", "golden", "sz_speed.config")
    r = subprocess.run([exe, "-z", "-f", "-c", cfg, "-i", src, "-M", "ABS", "-A", "1E-4", "-3", "8", "8", "128"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and os.path.exists(src + ".sz"), r.stdout[-600:] + r.stderr[-600:]
    assert os.path.getsize(src + ".sz") == 3546                      # config #1's BEST_SPEED stream
    r = subprocess.run([exe, "-p", "-s", src + ".sz"], capture_output=True, text=True, cwd=str(tmp_path))
    assert "SZ Compression Meta Data" in r.stdout and "FLOAT" in r.stdout and "8192" in r.stdout, r.stdout[-600:]
    r = subprocess.run([exe, "-x", "-f", "-s", src + ".sz", "-3", "8", "8", "128", "-i", src, "-a"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and os.path.exists(src + ".sz.out"), r.stdout[-600:] + r.stderr[-600:]
    back = np.fromfile(src + ".sz.out", dtype=np.float32)
    assert back.size == data.size and float(np.abs(back.astype(np.float64) - data).max()) <= 1e-4
    assert "PSNR" in r.stdout and "99.11" in r.stdout, r.stdout[-600:]      # the tool's own -a report: PSNR 99.114337
