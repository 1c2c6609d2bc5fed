#!/usr/bin/env python3
"""Static evidence for the k_omp_* kernels, written while they had not run on hardware (they have since: round 4): resource usage, the order of memory accesses and
waits, instruction mix per step -- read off hipcc's own output.  Writes profiles/r03_k_omp_static_isa_summary.txt.  No GPU needed."""
import os
import re
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''#include <hip/hip_runtime.h>
#include <cstdint>
#include "%s/sz_amd/csrc/szhip_kernels.h"
#define BOX(T, D, V) template __global__ void k_omp_box<T, D, V>(szh_omp_geom, const T*, T*, T, T, int, uint16_t*, unsigned*, u64*, T*, const T*, const u64*);
BOX(float, false, true) BOX(float, true, true) BOX(double, false, true) BOX(double, true, true) BOX(float, false, false)
template __global__ void k_omp_gather<float>(szh_omp_geom, const float*, const uint16_t*, const unsigned*, const u64*, float*);
''' % ROOT
FLAGS = "-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt".split()


def main():
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.hip"), "w").write(SRC)
        r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I", os.path.join(ROOT, "sz_amd", "csrc"), "--cuda-device-only",
                            "-Rpass-analysis=kernel-resource-usage", "-S", os.path.join(d, "t.hip"), "-o", os.path.join(d, "t.s")], capture_output=True, text=True)
        res, asm = r.stderr, open(os.path.join(d, "t.s")).read()
    out = ["k_omp_* kernels: static evidence (hipcc %s). NOT a measurement: the kernels had not run on hardware when round 3 ended (DESIGN 4h).\n"
           "Regenerate: python tools/omp_isa_summary.py\n\n== resource usage (-Rpass-analysis=kernel-resource-usage)\n" % " ".join(FLAGS)]
    for b in re.split(r"remark: Function Name: ", res)[1:]:
        name = b.split()[0]
        if "k_omp" not in name:
            continue
        g = lambda k: re.search(k + r": (\d+)", b).group(1)
        out.append("%-84s VGPRs %3s  SGPRs %3s  scratch %s B/lane  occupancy %s waves/SIMD  static LDS %5s B\n" % (
            name[:84], g("VGPRs"), g("TotalSGPRs"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
    for kn, label in (("_Z9k_omp_boxIfLb0ELb1E", "k_omp_box<float, compress, VEC>"), ("_Z9k_omp_boxIfLb1ELb1E", "k_omp_box<float, decompress, VEC>")):
        a = asm.index("\n" + kn)
        lines = [l for l in asm[a:asm.index("s_endpgm", a)].split("\n") if l.strip() and not l.strip().startswith(";")]
        out.append("\n== %s: memory accesses, waits for them and barriers in program order (index = instruction line of the kernel)\n" % label)
        for i, l in enumerate(lines):
            if re.search(r"s_waitcnt vmcnt|global_load|global_store|s_barrier|Loop Header", l):
                out.append("%5d  %s\n" % (i, l.strip()[:110]))
        hdrs = [i for i, l in enumerate(lines) if "Loop Header" in l]
        bars = [i for i, l in enumerate(lines) if "s_barrier" in l]
        # the step loop: the loop header followed by the longest run of barriers
        best = max(hdrs, key=lambda h: sum(1 for x in bars if x > h and not any(h < h2 < x for h2 in hdrs)))
        inloop = [x for x in bars if x > best and not any(best < h2 < x for h2 in hdrs)]
        loop = [l.split()[0] for l in lines[best + 1:inloop[-1] + 1] if not l.strip().endswith(":") and not l.startswith(".")]
        nb = len(inloop)
        cnt = lambda f: sum(1 for x in loop if f(x))
        out.append("step loop: %d steps unrolled, %d instructions; per step %.0f VALU, %.0f SALU / branch, %.1f LDS, %.2f global accesses\n" % (
            nb, len(loop), cnt(lambda x: x.startswith("v_")) / nb, cnt(lambda x: x.startswith("s_") and x not in ("s_barrier", "s_waitcnt", "s_nop")) / nb,
            cnt(lambda x: x.startswith("ds_")) / nb, cnt(lambda x: x.startswith("global_")) / nb))
    open(os.path.join(ROOT, "profiles", "r03_k_omp_static_isa_summary.txt"), "w").write("".join(out))
    print("".join(out))


if __name__ == "__main__":
    main()
