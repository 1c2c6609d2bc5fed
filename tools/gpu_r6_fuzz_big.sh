#!/bin/bash
# round 6: random differential cases on LARGER 3-D arrays (tools/gpu_fuzz.py with FUZZ_MAXDIM: extents 20 .. MAXDIM, rows a multiple of four values two times in three: many tiles of the sweep, many
# segments of the packing passes and of k_col_unpack), the last build of the round
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_differential_fuzz_larger_arrays.txt
echo "# tools/gpu_r6_fuzz_big.sh: FUZZ_MAXDIM=<largest extent> tools/gpu_fuzz.py <cases> <seed> -- oracle against the HIP library, streams byte for byte, decoded values bit for bit" > $out
run() { echo "[$1] $(env $2 timeout 1500 python tools/gpu_fuzz.py $3 $4 2>&1 | grep -E '^fuzz:|^FAIL|EXCEPTION' | tail -4 | tr '\n' ' ')" >> $out; }
run "extents up to 230, 350 cases" FUZZ_MAXDIM=230 350 631
run "extents up to 150, 1200 cases" FUZZ_MAXDIM=150 1200 632
run "extents up to 230, 3000 cases" FUZZ_MAXDIM=230 3000 633
run "extents up to 400, 500 cases" FUZZ_MAXDIM=400 500 634
run "extents up to 150, 6 KB segments both ways, 1500 cases" "FUZZ_MAXDIM=150 SZ_HIP_SEG_TILE_KB=6 SZ_HIP_UNPACK_TILE_KB=6" 1500 635
run "extents up to 150, the older passes both ways (SZ_HIP_SEGENC=0, SZ_HIP_COL_UNPACK=0), 1000 cases" "FUZZ_MAXDIM=150 SZ_HIP_SEGENC=0 SZ_HIP_COL_UNPACK=0" 1000 636
cat $out
