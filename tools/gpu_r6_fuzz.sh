#!/bin/bash
# round 6: random differential cases on the final build (SZ 2.1 path: the packing passes from natural-order codes, k_col_unpack, the lean sweep; the SZ 1.4 container; point-wise relative bounds), oracle against HIP, streams and decoded bits
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_differential_fuzz_long.txt
echo "# round 6, final build: tools/gpu_fuzz.py <cases> <seed> [path] -- oracle against the HIP library, streams byte for byte, decoded values bit for bit" > $out
for job in "12000 611" "12000 612" "12000 613" "4000 614 sz14" "2000 615 pwr" "2000 616 msst"; do
  ( timeout 1500 python tools/gpu_fuzz.py $job 2>&1 | grep -E "^fuzz:|^FAIL|EXCEPTION" | tail -5 | sed "s/^/[$job] /" ) >> $out
done
SZ_HIP_SEG_TILE_KB=6 SZ_HIP_UNPACK_TILE_KB=6 timeout 1500 python tools/gpu_fuzz.py 6000 617 2>&1 | grep -E "^fuzz:|^FAIL|EXCEPTION" | tail -5 | sed "s/^/[6000 617, 6 KB segments] /" >> $out
SZ_HIP_SEGHIST=0 SZ_HIP_SEG_SCAN1=0 timeout 1500 python tools/gpu_fuzz.py 6000 618 2>&1 | grep -E "^fuzz:|^FAIL|EXCEPTION" | tail -5 | sed "s/^/[6000 618, k_hist_u16 + k_col_bits + general scans] /" >> $out
cat $out
