#!/bin/bash
# round 6: random differential cases on the final build (SZ 2.1 path: the packing passes from natural-order codes, k_col_unpack, the lean sweep; the SZ 1.4 container), oracle against HIP, streams and decoded bits
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_differential_fuzz.txt
echo "# round 6, final build: tools/gpu_fuzz.py <cases> <seed> [path] -- oracle against the HIP library, streams byte for byte, decoded values bit for bit" > $out
for job in "4000 601" "4000 602" "3000 603" "1500 604 sz14"; do
  ( timeout 1500 python tools/gpu_fuzz.py $job 2>&1 | grep -E "^fuzz:|^FAIL|EXCEPTION" | tail -5 | sed "s/^/[$job] /" ) >> $out
done
SZ_HIP_SEG_TILE_KB=6 SZ_HIP_UNPACK_TILE_KB=6 timeout 1500 python tools/gpu_fuzz.py 2500 605 2>&1 | grep -E "^fuzz:|^FAIL|EXCEPTION" | tail -5 | sed "s/^/[2500 605, 6 KB segments] /" >> $out
SZ_HIP_FIT_TILE=1 timeout 1500 python tools/gpu_fuzz.py 2500 606 2>&1 | grep -E "^fuzz:|^FAIL|EXCEPTION" | tail -5 | sed "s/^/[2500 606, k_fit_tile] /" >> $out
cat $out
