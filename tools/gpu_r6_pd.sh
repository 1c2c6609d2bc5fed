#!/bin/bash
# round 6: the sweep's prefetch distance (beam::PD: lines a wavefront asks ahead; register sets in flight) -- the shipped build against a variant library (SZ_AMD_LIB) on the S-field, the M-field, both decompressions, the C4 slab
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
out=gpurun_out/r6_pd_variants.txt; : > $out
for lib in "" $@; do
  [ -n "$lib" ] && export SZ_AMD_LIB=$GRAFT_REPO_ROOT/sz_amd/csrc/$lib
  echo "== ${lib:-libszhip.so}" >> $out
  FIELD=s NCALLS=16 TAG=s python tools/gpu_r6_calls.py 2>&1 | tail -1 >> $out
  FIELD=m NCALLS=16 TAG=m python tools/gpu_r6_calls.py 2>&1 | tail -1 >> $out
  SZ_HIP_BEAM_FEED=0 FIELD=m NCALLS=16 TAG=m-unfed python tools/gpu_r6_calls.py 2>&1 | tail -1 >> $out
  FIELD=c4 NCALLS=10 TAG=c4 python tools/gpu_r6_calls.py 2>&1 | tail -1 >> $out
  python tools/gpu_r5_mtime.py 512 m,s 2>&1 | grep dec_it >> $out
done
cat $out
